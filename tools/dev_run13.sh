#!/bin/bash
# dev run 13 (round 5): one-launch input staging + FPN order variants: parity, then A/B on ONE box
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export ZSG_TUNE_CACHE=$O/tune13.json
cp $O/tune12.json $O/tune13.json 2>/dev/null
timeout 1200 python -m pytest tests/test_gpu_net.py tests/test_gpu_determinism.py -x -q > $O/t13_net.log 2>&1; tail -3 $O/t13_net.log
ZSG_FPN_ORDER=p6 timeout 1200 python -m pytest tests/test_gpu_net.py -x -q -k "fpn or e2e or golden" > $O/t13_net_p6.log 2>&1; tail -3 $O/t13_net_p6.log
Q="--no-cpu-baseline --steps 100 --warmup 10 --other-configs off"
run() { echo -n "$* : "; env "$@" python bench.py $Q 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['median_ms_per_step'], d['forward']['median_ms'], d['forward']['mfma_frac'])"; }
for i in 1 2 3; do
  run ZSG_STAGE_INPUTS=0 ZSG_FPN_ORDER=0
  run ZSG_STAGE_INPUTS=1 ZSG_FPN_ORDER=0
  run ZSG_STAGE_INPUTS=1 ZSG_FPN_ORDER=p6
  run ZSG_STAGE_INPUTS=1 ZSG_FPN_ORDER=p6m
done > $O/ab13.txt 2>&1
cat $O/ab13.txt
