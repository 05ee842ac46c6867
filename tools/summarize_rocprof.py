"""Condense a tools/rocprof_round.sh output directory into the small files committed under profiles/:
<tag>_kernel_stats.csv (rocprofv3 --stats per-kernel summary) and <tag>_hbm_traffic.json (per-launch HBM bytes per
kernel class from the FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for
wide coalesced reads on gfx950)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "gpurun_out", "profiles_" + tag)
os.makedirs(dst, exist_ok=True)
ks = glob.glob(out + "/stats/*/*kernel_stats.csv")
if ks:
    shutil.copy(ks[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))


DEFAULTS = {"igemm_kernel": [None] * 4 + ["1"], "wino_kernel": [None] * 3, "wgrad_kernel": [None] * 5 + ["true"]}


def klass(name):
    """the kernel name as libzsg's event profiler reports it — 'igemm_kernel<64, 64, 4, false, 2>', 'wino_kernel<2, 2, 4>',
    'wgrad_kernel<2, 1, 16, 2, 4>': rocprofv3 prints every template argument, libzsg leaves trailing defaults out and writes the
    variants as suffixes: '+pre' (BatchNorm-applying loader), '+sk' (stream-K), '+k64' (64-deep K tile)"""
    name = name.replace("void ", "").split("(")[0]
    if name.startswith("wgrad_reduce"):
        return "wgrad_reduce_kernel"
    base = name.split("<")[0]
    if base in DEFAULTS and "<" in name:
        args = [a.strip() for a in name[name.index("<") + 1:name.rindex(">")].split(",")]
        suffix = ""
        if base == "igemm_kernel" and len(args) >= 6:
            # <BM, BN, NW, MERGE_X, KS, BK, PRE, SK>
            pre = len(args) > 6 and args[6] == "true"
            sk = len(args) > 7 and args[7] == "true"
            suffix = ("+pre" if pre else "") + ("+sk" if sk else "") + ("+k64" if args[5] == "64" else "")
            args = args[:5]
        if base == "wino_kernel" and len(args) >= 4:
            suffix = "+sk" if args[3] == "true" else ""
            args = args[:3]
        dflt = DEFAULTS[base]
        while len(args) > 1 and len(args) <= len(dflt) and dflt[len(args) - 1] is not None and args[-1] == dflt[len(args) - 1]:
            args.pop()
        return f"{base}<{', '.join(args)}>{suffix}"
    return name[:47]


res = {}
for pass_, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = glob.glob(out + f"/{pass_}/*/*counter_collection.csv")
    agg = collections.defaultdict(lambda: [0, 0.0])
    if f:
        for r in csv.DictReader(open(f[0])):
            a = agg[klass(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        res.setdefault(k, {})[key + "_KB_per_launch"] = v / n
        res[k]["launches_in_pass"] = n
for k, d in res.items():
    d["hbm_bytes_per_launch"] = (2.0 * d.get("FETCH_SIZE_KB_per_launch", 0) + d.get("WRITE_SIZE_KB_per_launch", 0)) * 1024
if ks:                                   # average duration per kernel from the --stats pass, for bench.py's cross-check
    for r in csv.DictReader(open(ks[0])):
        res.setdefault(klass(r["Name"]), {})["rocprof_avg_ms"] = float(r["AverageNs"]) / 1e6
kser = glob.glob(out + "/stats_serial/*/*kernel_stats.csv")        # ZSG_SIDE_STREAM=0: every launch alone on the GPU
if kser:
    shutil.copy(kser[0], os.path.join(dst, f"{tag}_kernel_stats_serial.csv"))
    for r in csv.DictReader(open(kser[0])):
        res.setdefault(klass(r["Name"]), {})["rocprof_avg_ms_serial"] = float(r["AverageNs"]) / 1e6
sys.path.insert(0, root)
import bench  # noqa: E402  (source_stamp: ties this summary to the kernel sources it was measured on)
res["_source_stamp"] = bench.source_stamp()
json.dump(res, open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w"), indent=1, sort_keys=True)
for k in sorted(res):
    if k.startswith(("igemm_kernel", "wgrad_kernel", "wino")):
        print(k, {a: round(b, 4) for a, b in res[k].items()})
if ks:
    print(open(ks[0]).read()[:3000])
