#!/bin/bash
# dev run 19 (round 5): UPPER BOUND of what removing the 69 slab-reduce launches could buy (ZSG_WG_NO_REDUCE=1 skips them: timing only)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export ZSG_TUNE_CACHE=$O/tune19.json
cp $O/tune17.json $O/tune19.json 2>/dev/null
Q="--no-cpu-baseline --steps 100 --warmup 10 --other-configs off --no-roofline"
run() { echo -n "$* : "; env "$@" python bench.py $Q 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['median_ms_per_step'])"; }
run ZSG_WG_NO_REDUCE=0 > /dev/null
for i in 1 2 3; do
  run ZSG_WG_NO_REDUCE=0
  run ZSG_WG_NO_REDUCE=1
done > $O/ab19.txt 2>&1
cat $O/ab19.txt
