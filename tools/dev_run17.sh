#!/bin/bash
# dev run 17 (round 5): bnpre with 128x128 tiles + packed mask stores: parity, listing, A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export ZSG_TUNE_CACHE=$O/tune17.json
cp $O/tune15.json $O/tune17.json 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "bnpre" > $O/t17_ops.log 2>&1; tail -3 $O/t17_ops.log
python tools/fwd_listing.py fwd 2>&1 | grep -i "bnpre" > $O/listing17.txt; cat $O/listing17.txt
Q="--no-cpu-baseline --steps 100 --warmup 10 --other-configs off"
run() { echo -n "$* : "; env "$@" python bench.py $Q 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['median_ms_per_step'], d['forward']['median_ms'], d['forward']['mfma_frac'])"; }
for i in 1 2 3; do
  run ZSG_BN_PRE_MIN_MB=0
  run ZSG_BN_PRE_MIN_MB=90
  run ZSG_BN_PRE_MIN_MB=40
done > $O/ab17.txt 2>&1
cat $O/ab17.txt
