#!/bin/bash
# dev run 1 (round 5): new tests + bench legs + forward listing on the dev tune cache
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export ZSG_TUNE_CACHE=$R/tools/dev_tune.json
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "wgrad or batchnorm or test_conv_fwd" > $O/t_ops.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_fullshape.py -x -q -rP -k "configs0 or ssd_vgg_b32 or learnable" > $O/t_full.log 2>&1
timeout 600 python -m pytest tests/test_gpu_bnb.py tests/test_gpu_bnb_net.py -x -q > $O/t_bnb.log 2>&1
python tools/fwd_listing.py fwd > $O/listing2.txt 2>&1
python bench.py --other-configs on > $O/bench_full.log 2> $O/bench_full.err
cp $R/tools/dev_tune.json $O/dev_tune_after.json
