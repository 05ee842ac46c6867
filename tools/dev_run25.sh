cd $GRAFT_REPO_ROOT
bash tools/best_of_tunings.sh 6
python tools/make_tuning_table.py --seed gpurun_out/tunings/best.json r50 r18 ssd r101 > gpurun_out/tunings/make_table.log 2>&1
cp zsgnet-pytorch_amd/tuning/gfx950.json gpurun_out/tunings/gfx950.json
for i in 1 2 3; do python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --other-configs off 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo; done | tee gpurun_out/tunings/table_check.txt
