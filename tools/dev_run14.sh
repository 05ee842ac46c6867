#!/bin/bash
# dev run 14 (round 5): rocprofv3 kernel trace of whole steps -> per-launch listing with queue gaps (backward waits), forward trace again
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export ZSG_TUNE_CACHE=$O/tune14.json
cp $O/tune13.json $O/tune14.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 20 --warmup 5 --other-configs off --no-roofline > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt14 /tmp/kt14f
rocprofv3 --kernel-trace -d /tmp/kt14 --output-format csv -- python $R/bench.py --steps 10 --warmup 6 --no-cpu-baseline --no-roofline --other-configs off 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
F=$(find /tmp/kt14 -name "*kernel_trace.csv" | head -1)
cd $R
python tools/trace_overlap.py $F 3 --list > $O/trace_step14.txt 2>&1
cd /tmp; rocprofv3 --kernel-trace -d /tmp/kt14f --output-format csv -- python $R/tools/trace_fwd.py > /dev/null 2>&1
F=$(find /tmp/kt14f -name "*kernel_trace.csv" | head -1)
cd $R; python tools/trace_fwd.py $F 3 > $O/trace_fwd14.txt 2>&1; tail -1 $O/trace_fwd14.txt
grep "^step\|^busy" $O/trace_step14.txt
