#!/bin/bash
# dev run 10 (round 5): the streaming first-layer kernel (csrc/mx.hip): parity, then listing + bench with it in the tuner
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export ZSG_TUNE_CACHE=$O/tune10.json
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "mx_streaming or pw_streaming" > $O/t10_mx.log 2>&1
tail -5 $O/t10_mx.log
python tools/fwd_listing.py fwd 2>&1 | head -12 > $O/listing10.txt
for i in 1 2; do python bench.py --no-cpu-baseline --steps 50 --warmup 10 --other-configs off 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['median_ms_per_step'], d['forward']['median_ms'], d['forward']['mfma_frac'], d['settle_steps'])"; done > $O/bench10.txt 2>&1
cat $O/listing10.txt $O/bench10.txt
