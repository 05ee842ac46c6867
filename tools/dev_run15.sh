#!/bin/bash
# dev run 15 (round 5): input staging with float qlens (the bench's dtype): parity + A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export ZSG_TUNE_CACHE=$O/tune15.json
cp $O/tune14.json $O/tune15.json 2>/dev/null
timeout 1200 python -m pytest tests/test_gpu_net.py -x -q -k "staging or buckets or golden" > $O/t15_net.log 2>&1; tail -3 $O/t15_net.log
Q="--no-cpu-baseline --steps 100 --warmup 10 --other-configs off"
run() { echo -n "$* : "; env "$@" python bench.py $Q 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['median_ms_per_step'], d['forward']['median_ms'], d['forward']['mfma_frac'])"; }
for i in 1 2 3; do
  run ZSG_STAGE_INPUTS=0
  run ZSG_STAGE_INPUTS=1
done > $O/ab15.txt 2>&1
cat $O/ab15.txt
