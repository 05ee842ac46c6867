#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_wino4.py -x -q -rP > $O/t5_wino4.log 2>&1
timeout 600 python tools/bench_wino4.py > $O/bench_wino4.txt 2>&1
