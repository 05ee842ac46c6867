#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_wino4.py -x -q > $O/t6_wino4.log 2>&1
for v in "ZSG_W4_ORDER=1" "ZSG_W4_ORDER=0" "ZSG_W4_ORDER=1 ZSG_W4_ABL=3"; do echo "== $v"; env $v timeout 300 python tools/bench_wino4.py 2>/dev/null | head -2; done > $O/bench_wino4_abl.txt 2>&1
