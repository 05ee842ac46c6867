#!/bin/bash
# dev run 12 (round 5): packed language map + mx kernel: parity subset, forward trace with per-boundary gaps, quick bench
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export ZSG_TUNE_CACHE=$O/tune12.json
cp $O/tune11.json $O/tune12.json 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "lang_map or head_conv0 or mx_streaming" > $O/t12_ops.log 2>&1; tail -3 $O/t12_ops.log
timeout 1200 python -m pytest tests/test_gpu_net.py -x -q > $O/t12_net.log 2>&1; tail -3 $O/t12_net.log
for i in 1 2; do python bench.py --no-cpu-baseline --steps 100 --warmup 10 --other-configs off 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['median_ms_per_step'], d['forward']['median_ms'], d['forward']['mfma_frac'])"; done > $O/bench12.txt 2>&1
cat $O/bench12.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt12
rocprofv3 --kernel-trace -d /tmp/kt12 --output-format csv -- python $R/tools/trace_fwd.py > /dev/null 2>&1
F=$(find /tmp/kt12 -name "*kernel_trace.csv" | head -1)
cd $R; python tools/trace_fwd.py $F 3 > $O/trace_fwd12.txt 2>&1; tail -2 $O/trace_fwd12.txt
