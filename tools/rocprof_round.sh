#!/bin/bash
# Round profile on the GPU box: (1) tune once, (2) rocprofv3 --kernel-trace --stats of the bench command,
# (2b) the same with every launch on one stream (ZSG_SIDE_STREAM=0: per-kernel durations undisturbed by co-running kernels),
# (3) separate --pmc passes for FETCH_SIZE / WRITE_SIZE (HBM traffic), all on the tuned steady state.
# Every pass runs the headline configuration ONLY (--other-configs off): the per-kernel averages and the traffic per launch must be those of
# the launches bench.py's roofline leg times, not a mix with the SSD-VGG / ResNet-101 legs.
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03}; OUT=$R/gpurun_out/rp_$TAG; mkdir -p $OUT
[ -n "$USE_SHIPPED" ] || export ZSG_TUNE_CACHE=$OUT/tune_cache.json     # USE_SHIPPED=1: the shipped, source-stamped table instead of a private cache
cd $R && python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --other-configs off > $OUT/tune_run.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --other-configs off > $OUT/bench_under_rocprof.log 2>&1
ZSG_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats -d $OUT/stats_serial --output-format csv -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --other-configs off > $OUT/bench_under_rocprof_serial.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --other-configs off > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --other-configs off > /dev/null 2>&1
cd $R && python tools/summarize_rocprof.py $OUT $TAG
cp $R/gpurun_out/profiles_$TAG/${TAG}_hbm_traffic.json $R/profiles/ 2>/dev/null || true   # a bench.py run that follows in the same call picks the fresh, stamped traffic up
