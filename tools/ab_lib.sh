#!/bin/bash
# Developer tool (GPU box): A/B of two builds of libzsg on ONE box with a shared tuning table.
#   tools/ab_lib.sh <other libzsg.so> [rounds]   -> alternating `bench.py --steps 100 --warmup 20` runs: "old" = the other library
#   (ZSG_LIB_PATH), "new" = the in-tree one; also the per-kernel ms/step of the HBM-bound kernels from one roofline run each.
R=${GRAFT_REPO_ROOT:-/root/repo}; OLD=$1; N=${2:-4}; mkdir -p $R/gpurun_out/ab; export ZSG_TUNE_CACHE=$R/gpurun_out/ab/tune_lib.json ZSG_SHIPPED_TUNE=0
B="python $R/bench.py --no-cpu-baseline --no-roofline --steps 100 --warmup 20 ${AB_ARGS:-}"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], (d.get("forward") or {}).get("median_ms"))'
if [ -n "$TUNE_WITH_OLD" ]; then ZSG_LIB_PATH=$OLD $B > /dev/null 2>&1; else $B > /dev/null 2>&1; fi     # (TUNE_WITH_OLD=1: the shared table holds the OTHER library's choices)
for i in $(seq $N); do
  ZSG_LIB_PATH=$OLD $B 2>/dev/null | grep "^{" | python -c "$P" old
  $B 2>/dev/null | grep "^{" | python -c "$P" new
done
K='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], [(k["kernel"], k["ms_per_step"], k.get("gbps")) for k in d["roofline"]["top_kernels"] if k.get("tflops_executed") is None])'
ZSG_LIB_PATH=$OLD python $R/bench.py --no-cpu-baseline --steps 30 --warmup 10 ${AB_ARGS:-} 2>/dev/null | grep "^{" | python -c "$K" old
python $R/bench.py --no-cpu-baseline --steps 30 --warmup 10 ${AB_ARGS:-} 2>/dev/null | grep "^{" | python -c "$K" new
