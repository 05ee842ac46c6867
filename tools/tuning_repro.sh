#!/bin/bash
# Developer tool (GPU box): is a from-scratch tuning reproducible, and does the in-step refinement pay?
#   tools/tuning_repro.sh [N=3]   -> N fresh processes that tune from scratch WITHOUT the in-step refinement, N with it (bench.py --refine
#   auto: ZSGNet.refine_tuning re-ranks the tuner's near-ties inside the real two-stream step), then — when the shipped table matches the
#   sources — N with the shipped table.  One line per run: images/s, ms/step, median, forward ms, refinement summary.
R=${GRAFT_REPO_ROOT:-/root/repo}; N=${1:-3}; cd $R
B="python bench.py --no-cpu-baseline --no-roofline --forward-leg --other-configs off --steps ${AB_STEPS:-100} --warmup 20 ${AB_ARGS:-}"
P='import sys,json; d=json.loads(sys.stdin.read()); t=d["tuning"]; print(sys.argv[1], d["value"], d["ms_per_step"], d["median_ms_per_step"], "fwd", (d.get("forward") or {}).get("median_ms"), "tuned_now", t.get("tuned_now"), "loaded", t.get("loaded"), "refined", t.get("refined"))'
for i in $(seq $N); do ZSG_SHIPPED_TUNE=0 $B --refine off 2>/dev/null | grep "^{" | python -c "$P" "scratch,no-refine"; done
for i in $(seq $N); do ZSG_SHIPPED_TUNE=0 $B --refine auto 2>/dev/null | grep "^{" | python -c "$P" "scratch,refined"; done
for i in $(seq $N); do $B 2>/dev/null | grep "^{" | python -c "$P" "shipped-table"; done
