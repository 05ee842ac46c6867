#!/bin/bash
# Developer tool (GPU box): A/B of environment switches on ONE box with a shared tuning table.
#   tools/ab_env.sh "ZSG_X=0" "ZSG_X=1" ...   -> 4 alternating rounds of `bench.py --steps 100 --warmup 20`, ms/step each
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out/ab; export ZSG_TUNE_CACHE=${ZSG_TUNE_CACHE:-$R/gpurun_out/ab/tune.json}
B="python $R/bench.py --no-cpu-baseline --no-roofline --steps ${AB_STEPS:-100} --warmup 20 ${AB_ARGS:-}"     # AB_ARGS: e.g. "--arch resnet101 --img 600 --bs 32"
for v in "$@"; do env $v $B > /dev/null 2>&1; done      # (tune every variant's shapes first)
for i in 1 2 3 4; do for v in "$@"; do echo "$v $(env $v $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; done; done
