#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export ZSG_TUNE_CACHE=$R/tools/dev_tune.json
timeout 900 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_wino4.py -x -q > $O/t8_trainer.log 2>&1
timeout 900 python tools/loader_rate.py 512 1 4 16 > $O/loader_rate.txt 2>&1
timeout 300 python tools/one_launch.py fwd "backbone.encoder.layer3.1.conv1" > $O/one_launch.txt 2>&1
timeout 300 python tools/one_launch.py bwd "wgrad:backbone.encoder.layer3.1.conv1" >> $O/one_launch.txt 2>&1
