#!/bin/bash
# dev run 9 (round 5, second session): listings of both programs + a quick bench on the dev tune cache
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export ZSG_TUNE_CACHE=$R/tools/dev_tune.json
python tools/fwd_listing.py fwd bwd prep > $O/listing9.txt 2>&1
for i in 1 2; do python bench.py --no-cpu-baseline --steps 50 --warmup 10 --other-configs off 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['median_ms_per_step'], d['forward']['median_ms'], d['forward']['mfma_frac'], d['settle_steps'])"; done > $O/bench9.txt 2>&1
python bench.py --no-cpu-baseline --other-configs off > $O/bench9_full.log 2>$O/bench9_full.err
cp $R/tools/dev_tune.json $O/dev_tune_after9.json
