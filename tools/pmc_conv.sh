#!/bin/bash
# MFMA-pipe / LDS / wait counters of single convolution launches (rocprofv3 --pmc, counters only: no sys/hip traces).
# usage (GPU box): bash tools/pmc_conv.sh   -> gpurun_out/pmc_<tag>/*.csv + a per-launch mean JSON
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pmc_r01; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # tag shape mode bm bn w8 sp
  tag=$1; shift
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $OUT/${tag}_a --output-format csv -- python $R/tools/one_conv.py "$@" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS -d $OUT/${tag}_b --output-format csv -- python $R/tools/one_conv.py "$@" > /dev/null 2>&1
}
run head3x3_fwd_128x128w8 head3x3_38 fwd 128 128 1
run head3x3_fwd_64x64 head3x3_38 fwd 64 64 0
run head3x3_dgrad_128x128w8 head3x3_38 dgrad 128 128 1
run head3x3_wgrad_128x128w8 head3x3_38 wgrad 128 128 1
run l3conv2_fwd_64x64_s2 l3_conv2 fwd 64 64 0 2
run l1conv3_fwd_64x64 l1_conv3 fwd 64 64 0
cd $R && python - <<'PY'
import csv, glob, json, os, collections
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "pmc_r01")
res = {}
for d in sorted(glob.glob(out + "/*_[ab]")):
    tag = os.path.basename(d)[:-2]
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if "igemm_kernel" in r["Kernel_Name"] or "wgrad_kernel" in r["Kernel_Name"]:
                a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
        for k, (n, v) in agg.items():
            res.setdefault(tag, {})[k] = v / n
for tag, c in res.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        # MFMA busy cycles are summed over the 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE is the kernel's GPU-active cycles
        c["mfma_pipe_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 256 * 4)
json.dump(res, open(out + "/conv_kernel_counters_mean_per_launch.json", "w"), indent=1, sort_keys=True)
for tag, c in res.items():
    print(tag, {k: round(v, 3) if v < 10 else int(v) for k, v in c.items() if k in ("mfma_pipe_busy_frac", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE")})
PY
