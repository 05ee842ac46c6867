#!/bin/bash
# Developer tool (GPU box): tools/ab_env.sh with the forward leg printed beside the step (images/s, ms/step, median ms/step, forward ms).
#   tools/ab_env2.sh "ZSG_X=0" "ZSG_X=1" ...     (AB_ROUNDS alternating rounds, default 3; AB_ARGS extra bench.py flags)
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out/ab; export ZSG_TUNE_CACHE=${ZSG_TUNE_CACHE:-$R/gpurun_out/ab/tune2.json} ZSG_SHIPPED_TUNE=${ZSG_SHIPPED_TUNE:-0}
B="python $R/bench.py --no-cpu-baseline --no-roofline --forward-leg --other-configs off --steps ${AB_STEPS:-100} --warmup 20 ${AB_ARGS:-}"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d["median_ms_per_step"], "fwd", (d.get("forward") or {}).get("median_ms"), (d.get("forward") or {}).get("mfma_frac"))'
for v in "$@"; do env $v $B > /dev/null 2>&1; done      # (tune every variant's shapes first)
for i in $(seq ${AB_ROUNDS:-3}); do for v in "$@"; do env $v $B 2>/dev/null | grep "^{" | python -c "$P" "$v"; done; done
