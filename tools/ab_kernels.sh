#!/bin/bash
# Developer tool (GPU box): per-kernel isolated ms/step (bench.py's roofline leg, one stream) of two builds of libzsg with a shared tuning cache.
#   tools/ab_kernels.sh <other libzsg.so>
R=${GRAFT_REPO_ROOT:-/root/repo}; OLD=$1; mkdir -p $R/gpurun_out/ab; export ZSG_TUNE_CACHE=$R/gpurun_out/ab/tune_lib.json ZSG_SHIPPED_TUNE=0
K='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["ms_per_step"], d["forward"]["median_ms"]); [print("  %-44s %7.3f ms %5.1f launches %s" % (k["kernel"], k["ms_per_step"], k["launches_per_step"], k.get("frac"))) for k in d["roofline"]["top_kernels"]]'
python $R/bench.py --no-cpu-baseline --no-roofline --steps 5 --warmup 2 > /dev/null 2>&1
ZSG_LIB_PATH=$OLD python $R/bench.py --no-cpu-baseline --steps 30 --warmup 10 --other-configs off 2>/dev/null | grep "^{" | python -c "$K" old
python $R/bench.py --no-cpu-baseline --steps 30 --warmup 10 --other-configs off 2>/dev/null | grep "^{" | python -c "$K" new
