"""Register / scratch budget of every kernel in the BUILT library objects (zsgnet-pytorch_amd/build/*.o).

    python tools/kernel_resources.py            # table: kernel, VGPRs, AGPRs, SGPRs, LDS, scratch, spills
    python tools/kernel_resources.py --check    # exit 1 when any kernel spills vector registers to scratch memory (vgpr_spill_count > 0)

Reads the gfx950 code object embedded in each object file (.hip_fatbin section -> clang-offload-bundler -> the AMDGPU metadata note),
i.e. what actually ships — no recompilation.  `make -C zsgnet-pytorch_amd/csrc check` and __graft_entry__.build() run the --check form:
a spilling instantiation is a tuner candidate that can only lose and fails the build.
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "zsgnet-pytorch_amd", "build")


def kernels_of(obj: str):
    """[(name, {field: int})] of the gfx950 code object inside a host object file"""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat"), os.path.join(td, "co")
        r = subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj, os.path.join(td, "x.o")], capture_output=True)
        if r.returncode or not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return []
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}",
                            f"--output={co}", "--unbundle"], capture_output=True)
        if r.returncode or not os.path.exists(co):
            return []
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    out, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if line.lstrip().startswith("- ."):          # a new kernel record starts with its first key
            cur = {}
            out.append(cur)
        if cur is not None:
            cur[k] = v
    res = []
    for rec in out:
        if "symbol" not in rec and "name" not in rec:
            continue
        name = rec.get("name", rec.get("symbol", "?")).strip("'\"")
        f = {}
        for k in ("vgpr_count", "agpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "vgpr_spill_count",
                  "sgpr_spill_count", "max_flat_workgroup_size"):
            try:
                f[k] = int(rec.get(k, "0"))
            except ValueError:
                f[k] = 0
        res.append((name, f))
    return res


def demangle(names):
    r = subprocess.run([f"{LLVM}/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True)
    if r.returncode:
        return names
    return r.stdout.splitlines()


def main():
    check = "--check" in sys.argv
    rows = []
    for obj in sorted(glob.glob(os.path.join(BUILD, "*.o"))):
        if os.path.basename(obj).startswith("exp_"):      # (tools/experiments: not part of the product library)
            continue
        for name, f in kernels_of(obj):
            rows.append((os.path.basename(obj).split(".")[0], name, f))
    if not rows:
        print("kernel_resources: no code objects under", BUILD, "(build first)")
        return 1 if check else 0
    names = demangle([r[1] for r in rows]) if os.path.exists(f"{LLVM}/llvm-cxxfilt") else [r[1] for r in rows]
    bad = 0
    if not check:
        print(f"{'file':8s} {'vgpr':>4s} {'agpr':>4s} {'sgpr':>4s} {'lds':>6s} {'scratch':>7s} {'vspill':>6s} {'sspill':>6s}  kernel")
    for (fn, _, f), nm in zip(rows, names):
        nm = nm.split("(")[0].replace("void ", "")
        # what fails the build: VGPRs spilled to scratch memory.  SGPRs "spilled" to lanes of a spare VGPR (v_writelane / v_readlane, no
        # memory traffic; the stream-K pass loops keep a few kernel arguments there, outside their K loops) are listed, not failed.
        spill = f["vgpr_spill_count"] + (1 if f["private_segment_fixed_size"] else 0)
        bad += 1 if spill else 0
        if not check or spill:
            print(f"{fn:8s} {f['vgpr_count']:4d} {f['agpr_count']:4d} {f['sgpr_count']:4d} {f['group_segment_fixed_size']:6d} "
                  f"{f['private_segment_fixed_size']:7d} {f['vgpr_spill_count']:6d} {f['sgpr_spill_count']:6d}  {nm}")
    print(f"kernel_resources: {len(rows)} kernels, {bad} spilling vector registers to scratch")
    return 1 if (check and bad) else 0


if __name__ == "__main__":
    sys.exit(main())
