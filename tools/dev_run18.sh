#!/bin/bash
# dev run 18 (round 5): the whole GPU suite on the current tree
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/t18_all.log 2>&1; tail -8 $O/t18_all.log
