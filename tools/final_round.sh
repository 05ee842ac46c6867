#!/bin/bash
# Round-end measurement cycle on ONE GPU box, kernel sources frozen (any csrc edit invalidates the table's stamp):
#   (before A, once per source stamp: tools/best_of_tunings.sh 6 -> tools/refine_tuning.py best.json c*.json -> tools/_tunings/seed.json)
#   tools/final_round.sh A   tuning table (seeded) -> reproducibility runs -> full bench.py -> the other configurations
#   tools/final_round.sh B   rocprofv3 stats + HBM traffic (tools/rocprof_round.sh) -> bench.py with the fresh traffic file -> PMC
#                            counters (tools/pmc_round5.sh) -> co-issue micro-benchmark -> loader rate
# Everything lands under gpurun_out/r05final/; the builder copies what is judged into profiles/r05_*.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05final; mkdir -p $O; cd $R
Q="--no-cpu-baseline --no-roofline --other-configs off"
if [ "$1" = "A" ]; then
  # (tools/_tunings/seed.json: the fastest of six fresh tunings of the headline configuration, refined in-step — tools/best_of_tunings.sh, refine_tuning.py)
  # (used only for the sources it was made on: seed.stamp = the csrc stamp; after a kernel edit the tunings have to be made again)
  CUR=$(python3 zsgnet-pytorch_amd/csrc/stamp.py | grep -o '"[0-9a-f]*"' | tr -d '"')
  SEED=""; [ -f tools/_tunings/seed.json ] && [ "$(cat tools/_tunings/seed.stamp 2>/dev/null)" = "$CUR" ] && SEED="--seed tools/_tunings/seed.json"
  echo "seed: ${SEED:-none (stamp $CUR)}" > $O/seed_used.txt
  python tools/make_tuning_table.py $SEED r50 r18 ssd r101 > $O/make_table.log 2>&1
  cp zsgnet-pytorch_amd/tuning/gfx950.json $O/gfx950.json
  { echo "# five fresh processes, shipped table (python bench.py --steps 100 --warmup 20 $Q)"
    for i in 1 2 3 4 5; do python bench.py --steps 100 --warmup 20 $Q 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"tuning": {[^}]*}' | tr '\n' ' '; echo; done
    echo "# three fresh processes, each tuning from scratch (ZSG_SHIPPED_TUNE=0)"
    for i in 1 2 3; do ZSG_SHIPPED_TUNE=0 python bench.py --steps 100 --warmup 20 $Q 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"tuning": {[^}]*}' | tr '\n' ' '; echo; done
  } > $O/tuning_repro.txt 2>&1
  python bench.py > $O/bench_default.log 2>$O/bench_default.err
  python bench.py --backbone ssd_vgg --bs 32 --no-cpu-baseline --other-configs off > $O/bench_ssd_vgg_b32.log 2>&1
  python bench.py --arch resnet101 --img 600 --bs 32 --steps 30 --warmup 5 --no-cpu-baseline --other-configs off > $O/bench_r101_600_b32.log 2>&1
  python bench.py --arch resnet18 --no-cpu-baseline --other-configs off > $O/bench_r18.log 2>&1
  python bench.py --force-ddp --no-cpu-baseline --other-configs off > $O/bench_force_ddp.log 2>&1
  ZSG_COMM=native python bench.py --force-ddp --no-cpu-baseline --other-configs off > $O/bench_force_ddp_native.log 2>&1
else
  USE_SHIPPED=1 bash tools/rocprof_round.sh r05 > $O/rocprof_round.log 2>&1
  python bench.py > $O/bench_final.log 2>$O/bench_final.err
  bash tools/pmc_round5.sh > $O/pmc.log 2>&1
  [ -x tools/ubench/build/mfma_coissue ] && timeout 600 tools/ubench/build/mfma_coissue > $O/mfma_coissue.txt 2>&1
  timeout 900 python tools/loader_rate.py > $O/loader_rate.txt 2>&1
  # gpurun merges at most 64 MiB back: keep the summaries (gpurun_out/profiles_r05, pmc_r05/{summary.txt,counters.json,*.info}), drop the raw traces
  rm -rf $R/gpurun_out/rp_r05/stats $R/gpurun_out/rp_r05/stats_serial $R/gpurun_out/rp_r05/fetch $R/gpurun_out/rp_r05/write
  find $R/gpurun_out/pmc_r05 -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
  du -sh $R/gpurun_out
fi
ls -la $O
