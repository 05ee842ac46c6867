cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tunings
T=tools/_tunings
python tools/refine_tuning.py $T/best.json $T/c1.json $T/c2.json $T/c3.json $T/c4.json $T/c5.json 2>&1 | tee gpurun_out/tunings/refine.log
cp $T/best.refined.json gpurun_out/tunings/best.refined.json
python tools/make_tuning_table.py --seed gpurun_out/tunings/best.refined.json r50 r18 ssd r101 > gpurun_out/tunings/make_table2.log 2>&1
cp zsgnet-pytorch_amd/tuning/gfx950.json gpurun_out/tunings/gfx950_refined.json
for i in 1 2 3; do python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --other-configs off 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo; done | tee gpurun_out/tunings/table_check2.txt
