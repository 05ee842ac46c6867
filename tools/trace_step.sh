#!/bin/bash
# Developer tool (GPU box): rocprofv3 kernel trace of a short tuned bench run -> per-launch listing of one steady-state step
# (tools/trace_overlap.py --list) under gpurun_out/trace_<tag>/.  Environment switches (ZSG_*) pass through.
#   tools/trace_step.sh <tag> [step indices from the end, default "3 5"]
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-t}; shift; KS=${@:-3 5}; OUT=$R/gpurun_out/trace_$TAG; mkdir -p $OUT
export ZSG_TUNE_CACHE=${ZSG_TUNE_CACHE:-$R/gpurun_out/r3m/tune.json}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_$TAG
rocprofv3 --kernel-trace -d /tmp/kt_$TAG --output-format csv -- python $R/bench.py --steps 10 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
F=$(find /tmp/kt_$TAG -name "*kernel_trace.csv" | head -1)
cd $R
for k in $KS; do
  python tools/trace_overlap.py $F $k --list > $OUT/step_$k.txt
  echo "[$TAG] step -$k: main-stream gaps > 40 us"; grep "^  q0" $OUT/step_$k.txt | awk '{g=$8+0; if (g>40) print}'; grep "^busy\|^step" $OUT/step_$k.txt
done
