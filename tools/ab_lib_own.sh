#!/bin/bash
# Developer tool (GPU box): A/B of two builds of libzsg on ONE box, EACH WITH ITS OWN tuning cache (a shared cache hands one library the
# other's tile choices: tools/ab_lib.sh).   tools/ab_lib_own.sh <other libzsg.so> [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}; OLD=$1; N=${2:-3}; mkdir -p $R/gpurun_out/ab; export ZSG_SHIPPED_TUNE=0
B="python $R/bench.py --no-cpu-baseline --no-roofline --steps 100 --warmup 20 --other-configs off ${AB_ARGS:-}"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d["median_ms_per_step"], (d.get("forward") or {}).get("median_ms"))'
rm -f $R/gpurun_out/ab/tune_old.json $R/gpurun_out/ab/tune_new.json
ZSG_TUNE_CACHE=$R/gpurun_out/ab/tune_old.json ZSG_LIB_PATH=$OLD $B > /dev/null 2>&1
ZSG_TUNE_CACHE=$R/gpurun_out/ab/tune_new.json $B > /dev/null 2>&1
for i in $(seq $N); do
  ZSG_TUNE_CACHE=$R/gpurun_out/ab/tune_old.json ZSG_LIB_PATH=$OLD $B 2>/dev/null | grep "^{" | python -c "$P" old
  ZSG_TUNE_CACHE=$R/gpurun_out/ab/tune_new.json $B 2>/dev/null | grep "^{" | python -c "$P" new
done
