#!/bin/bash
# dev run 2 (round 5): in-kernel BatchNorm finalize — op tests, net tests, A/B against ZSG_BN_TAIL=0
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export ZSG_TUNE_CACHE=$R/tools/dev_tune.json
timeout 900 python -m pytest tests/test_gpu_bntail.py tests/test_gpu_bnb.py -x -q > $O/t2_tail.log 2>&1
timeout 900 python -m pytest tests/test_gpu_bnb_net.py tests/test_gpu_determinism.py tests/test_gpu_net.py -x -q > $O/t2_net.log 2>&1
timeout 900 python -m pytest tests/test_gpu_fullshape.py -x -q -rP -k "configs1 or ssd_vgg_b32" > $O/t2_full.log 2>&1
python tools/fwd_listing.py fwd bwd > $O/listing3.txt 2>&1
AB_STEPS=100 bash tools/ab_env.sh "ZSG_BN_TAIL=0" "ZSG_BN_TAIL=1" "ZSG_BN_TAIL=fwd" > $O/ab_tail.txt 2>&1
for v in 0 1; do ZSG_BN_TAIL=$v python bench.py --no-cpu-baseline --steps 50 --warmup 10 --other-configs off 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BN_TAIL=$v', d['ms_per_step'], d['forward'])"; done > $O/ab_tail_fwd.txt 2>&1
cp $R/tools/dev_tune.json $O/dev_tune_after2.json
