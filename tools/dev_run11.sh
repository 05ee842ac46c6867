#!/bin/bash
# dev run 11 (round 5): A/B of the streaming first-layer kernel on ONE box (ZSG_MX=0 / 1, shared tuning cache otherwise)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export ZSG_TUNE_CACHE=$O/tune11.json
Q="--no-cpu-baseline --steps 100 --warmup 10 --other-configs off"
ZSG_MX=0 python bench.py $Q > /dev/null 2>&1
ZSG_MX=1 python bench.py $Q > /dev/null 2>&1
for i in 1 2 3; do for mx in 0 1; do echo -n "ZSG_MX=$mx "; ZSG_MX=$mx python bench.py $Q 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['median_ms_per_step'], d['forward']['median_ms'], d['forward']['mfma_frac'])"; done; done > $O/ab11.txt 2>&1
cat $O/ab11.txt
