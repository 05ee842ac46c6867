#!/bin/bash
# dev run 4 (round 5): transposer + 16-wave finalize regression, listing
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export ZSG_TUNE_CACHE=$R/tools/dev_tune.json
timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_ops.py -x -q > $O/t4_net.log 2>&1
python tools/fwd_listing.py fwd prep > $O/listing5.txt 2>&1
for i in 1 2; do python bench.py --no-cpu-baseline --steps 50 --warmup 10 --other-configs off 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['median_ms_per_step'], d['forward']['median_ms'], d['forward']['mfma_frac'], d['settle_steps'])"; done > $O/bench4.txt 2>&1
