#!/bin/bash
# MFMA-pipe / LDS / wait / cache counters of single Winograd convolution launches (rocprofv3 --pmc, counters only).
# usage (GPU box): bash tools/pmc_wino.sh [tag]  -> gpurun_out/pmc_<tag>/wino_counters.json
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r02}; OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # tag shape TB BN sp
  tag=$1; shift
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE -d $OUT/${tag}_a --output-format csv -- python $R/tools/one_wino.py "$@" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS -d $OUT/${tag}_b --output-format csv -- python $R/tools/one_wino.py "$@" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${tag}_c --output-format csv -- python $R/tools/one_wino.py "$@" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/${tag}_d --output-format csv -- python $R/tools/one_wino.py "$@" > /dev/null 2>&1
}
run head_64x64 head 64 64
run l3conv2_64x64s2 l3_conv2 64 64 2
run l1conv2_64x64 l1_conv2 64 64
cd $R && python - "$OUT" <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(out + "/*_[abcd]")):
    tag = os.path.basename(d)[:-2]
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if "wino_kernel" in r["Kernel_Name"]:
                a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
        for k, (n, v) in agg.items():
            res.setdefault(tag, {})[k] = v / n
for tag, c in res.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        c["kernel_cycles"] = c["GRBM_GUI_ACTIVE"] / 8
        c["mfma_pipe_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["kernel_cycles"] * 1024 * (c["SQ_INSTS_MFMA"] * 64 / c["SQ_VALU_MFMA_BUSY_CYCLES"] if c.get("SQ_INSTS_MFMA") else 1))
    if "FETCH_SIZE" in c:
        c["hbm_read_MB_x2"] = 2 * c["FETCH_SIZE"] / 1024      # FETCH_SIZE is in KB; gfx950 reports half of wide streaming reads
json.dump(res, open(out + "/wino_counters.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
