#!/bin/bash
# Round-end measurement cycle on ONE GPU box, kernel sources frozen (any csrc edit invalidates the table's stamp):
#   tools/final_round.sh A   tuning table from scratch (single-launch medians, then the near-ties re-ranked inside the step:
#                            tools/make_tuning_table.py) -> reproducibility runs (tools/tuning_repro.sh: from scratch without / with the
#                            in-step refinement, then the shipped table) -> full bench.py -> the other configurations -> 1-rank DDP legs
#   tools/final_round.sh B   rocprofv3 stats + HBM traffic (tools/rocprof_round.sh) -> bench.py with the fresh traffic file -> PMC
#                            counters (tools/pmc_round5.sh, PMC_TAG) -> loader rate
#   tools/final_round.sh T   the GPU suite with the measured parity numbers printed (-rP) and the slowest tests (--durations)
#   tools/final_round.sh C   after a HOST-side change (kernel sources and table untouched): five fresh processes with the shipped table,
#                            the bench lines, rocprofv3 stats + traffic, bench.py with the fresh traffic file, the GPU suite
# Everything lands under gpurun_out/<tag>final/; the builder copies what is judged into profiles/<tag>_*.
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${ROUND_TAG:-r06}; O=$R/gpurun_out/${TAG}final; mkdir -p $O; cd $R
Q="--no-cpu-baseline --no-roofline --other-configs off"
if [ "$1" = "A" ]; then
  python tools/make_tuning_table.py r50 r18 ssd r101 > $O/make_table.log 2>&1
  cp zsgnet-pytorch_amd/tuning/gfx950.json $O/gfx950.json
  bash tools/tuning_repro.sh 3 > $O/tuning_repro.txt 2>&1
  python bench.py > $O/bench_default.log 2>$O/bench_default.err
  python bench.py --steps 20 --warmup 3 > $O/bench_steps20_warmup3.log 2>/dev/null
  python bench.py --backbone ssd_vgg --bs 32 --no-cpu-baseline --other-configs off > $O/bench_ssd_vgg_b32.log 2>&1
  python bench.py --arch resnet101 --img 600 --bs 32 --steps 30 --warmup 5 --no-cpu-baseline --other-configs off > $O/bench_r101_600_b32.log 2>&1
  python bench.py --arch resnet18 --no-cpu-baseline --other-configs off > $O/bench_r18.log 2>&1
  python bench.py --force-ddp --no-cpu-baseline --other-configs off > $O/bench_force_ddp.log 2>&1
  ZSG_COMM=native python bench.py --force-ddp --no-cpu-baseline --other-configs off > $O/bench_force_ddp_native.log 2>&1
elif [ "$1" = "B" ]; then
  USE_SHIPPED=1 bash tools/rocprof_round.sh $TAG > $O/rocprof_round.log 2>&1
  python bench.py > $O/bench_final.log 2>$O/bench_final.err
  PMC_TAG=$TAG bash tools/pmc_round5.sh > $O/pmc.log 2>&1
  timeout 900 python tools/loader_rate.py > $O/loader_rate.txt 2>&1
  # gpurun merges at most 64 MiB back: keep the summaries, drop the raw traces
  rm -rf $R/gpurun_out/rp_$TAG/stats $R/gpurun_out/rp_$TAG/stats_serial $R/gpurun_out/rp_$TAG/fetch $R/gpurun_out/rp_$TAG/write
  find $R/gpurun_out/pmc_$TAG -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
  du -sh $R/gpurun_out
elif [ "$1" = "C" ]; then
  P='import sys,json; d=json.loads(sys.stdin.read()); t=d["tuning"]; print(sys.argv[1], d["value"], d["ms_per_step"], d["median_ms_per_step"], "fwd", (d.get("forward") or {}).get("median_ms"), "tuned_now", t.get("tuned_now"), "loaded", t.get("loaded"))'
  for i in 1 2 3 4 5; do python bench.py --forward-leg --steps 100 --warmup 20 $Q 2>/dev/null | grep "^{" | python -c "$P" "shipped-table"; done > $O/shipped_repro.txt
  python bench.py > $O/bench_default.log 2>$O/bench_default.err
  python bench.py --steps 20 --warmup 3 > $O/bench_steps20_warmup3.log 2>/dev/null
  python bench.py --force-ddp --no-cpu-baseline --other-configs off > $O/bench_force_ddp.log 2>&1
  ZSG_COMM=native python bench.py --force-ddp --no-cpu-baseline --other-configs off > $O/bench_force_ddp_native.log 2>&1
  python bench.py --arch resnet101 --img 600 --bs 32 --steps 30 --warmup 5 --no-cpu-baseline --other-configs off > $O/bench_r101_600_b32.log 2>&1
  USE_SHIPPED=1 bash tools/rocprof_round.sh $TAG > $O/rocprof_round.log 2>&1
  python bench.py > $O/bench_final.log 2>$O/bench_final.err
  rm -rf $R/gpurun_out/rp_$TAG/stats $R/gpurun_out/rp_$TAG/stats_serial $R/gpurun_out/rp_$TAG/fetch $R/gpurun_out/rp_$TAG/write
  python -m pytest tests -m gpu -q -rP --durations=15 > $O/gpu_tests_rP.log 2>&1; echo "rc=$?" >> $O/gpu_tests_rP.log
  cat $O/shipped_repro.txt; grep -E "passed|failed" $O/gpu_tests_rP.log | tail -1
else
  python -m pytest tests -m gpu -q -rP --durations=15 > $O/gpu_tests_rP.log 2>&1; echo "rc=$?" >> $O/gpu_tests_rP.log
  tail -25 $O/gpu_tests_rP.log
fi
ls -la $O
