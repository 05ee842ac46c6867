#!/bin/bash
# fp32-MFMA vs bf16x6 variants of the implicit-GEMM kernel on the same launch: MFMA-pipe busy cycles, VALU / LDS instruction
# counts, wave cycles (rocprofv3 --pmc, counters only).  usage (GPU box): bash tools/pmc_bx.sh -> gpurun_out/pmc_bx/bx_counters.json
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pmc_bx; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # tag shape mode bm bn w8 sp bx
  tag=$1; shift
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE -d $OUT/${tag}_a --output-format csv -- python $R/tools/one_conv.py "$@" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS -d $OUT/${tag}_b --output-format csv -- python $R/tools/one_conv.py "$@" > /dev/null 2>&1
}
run head3x3_fp32_128x64w head3x3_38 fwd 128 64 1 0 0
run head3x3_bx_128x64w head3x3_38 fwd 128 64 1 0 1
run l3conv1_fp32_128x64w l3_conv1 fwd 128 64 1 0 0
run l3conv1_bx_128x64w l3_conv1 fwd 128 64 1 0 1
cd $R && python - "$OUT" <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(out + "/*_[ab]")):
    tag = os.path.basename(d)[:-2]
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if "igemm_kernel" in r["Kernel_Name"]:
                a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
        for k, (n, v) in agg.items():
            res.setdefault(tag, {})[k] = v / n
for tag, c in res.items():
    if "GRBM_GUI_ACTIVE" in c:
        c["kernel_cycles"] = c["GRBM_GUI_ACTIVE"] / 8        # (the counter is summed over the 8 XCDs)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "kernel_cycles" in c:
        c["mfma_pipe_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["kernel_cycles"] * 1024)      # 256 CUs x 4 SIMDs
    if "SQ_INSTS_VALU" in c and "SQ_INSTS_MFMA" in c:
        c["valu_per_mfma"] = c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"]
json.dump(res, open(out + "/bx_counters.json", "w"), indent=1, sort_keys=True)
for tag, c in sorted(res.items()):
    print(tag, {k: (round(v, 3) if v < 100 else int(v)) for k, v in c.items() if k in ("kernel_cycles", "mfma_pipe_busy_frac", "valu_per_mfma", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_VALU_MFMA_BUSY_CYCLES")})
PY
