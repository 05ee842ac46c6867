#!/bin/bash
# dev run 16 (round 5): bn3 + residual + ReLU applied by the next block's conv1 (zsg_conv_igemm_bnpre): parity, then A/B on ONE box
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export ZSG_TUNE_CACHE=$O/tune16.json
cp $O/tune15.json $O/tune16.json 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "bnpre" > $O/t16_ops.log 2>&1; tail -5 $O/t16_ops.log
timeout 1500 python -m pytest tests/test_gpu_net.py -x -q > $O/t16_net.log 2>&1; tail -5 $O/t16_net.log
Q="--no-cpu-baseline --steps 100 --warmup 10 --other-configs off"
run() { echo -n "$* : "; env "$@" python bench.py $Q 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['median_ms_per_step'], d['forward']['median_ms'], d['forward']['mfma_frac'])"; }
for i in 1 2 3; do
  run ZSG_BN_PRE_MIN_MB=0
  run ZSG_BN_PRE_MIN_MB=90
  run ZSG_BN_PRE_MIN_MB=40
  run ZSG_BN_PRE_MIN_MB=10
done > $O/ab16.txt 2>&1
cat $O/ab16.txt
ZSG_BN_PRE_MIN_MB=40 python tools/fwd_listing.py fwd 2>&1 | head -75 > $O/listing16.txt
