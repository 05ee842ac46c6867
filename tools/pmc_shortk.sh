#!/bin/bash
# Developer tool (GPU box): counters of ONE implicit-GEMM launch on layer1's 1x1 shapes (M = 90000, K or N = 64): waves in flight,
# MFMA-pipe busy cycles, instruction mix, waits.  usage: bash tools/pmc_shortk.sh -> gpurun_out/pmc_shortk/counters.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pmc_shortk; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  tag=$1; shift
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES SQ_WAVES -d $OUT/${tag}_a --output-format csv -- python $R/tools/one_conv.py "$@" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU -d $OUT/${tag}_b --output-format csv -- python $R/tools/one_conv.py "$@" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC -d $OUT/${tag}_c --output-format csv -- python $R/tools/one_conv.py "$@" > /dev/null 2>&1
}
run l1conv3_128x128w $1 fwd 128 128 1 0 0
[ -z "$PW_ONLY" ] && run l1conv3_64x64 $1 fwd 64 64 0 0 0
run l1conv3_pw64 $1 fwd 32 64 0 0 0        # the filter-resident streaming kernel (csrc/pw.hip): tile_hint BM = 32
run l1conv3_pw128 $1 fwd 32 128 0 0 0
cd $R && python - "$OUT" <<'PY' | tee $OUT/counters.txt
import csv, glob, os, sys, collections
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(out + "/*_[abc]")):
    tag = os.path.basename(d)[:-2]
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if "igemm_kernel" in r["Kernel_Name"] or "pw_kernel" in r["Kernel_Name"]:
                a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
        for k, (n, v) in agg.items():
            res.setdefault(tag, {})[k] = v / n
for tag, c in sorted(res.items()):
    kc = c.get("GRBM_GUI_ACTIVE", 0) / 8
    print(tag, "kernel cycles", int(kc))
    for k, v in sorted(c.items()):
        print(f"   {k:28s} {v:16.0f}   per CU-cycle {v / (kc * 256) if kc else 0:8.3f}")
PY
