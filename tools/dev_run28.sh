cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tunings
cp tools/_tunings/seed.json tools/_tunings/w8.json
python tools/refine_tuning.py tools/_tunings/w8.json --toggle-w8 2>&1 | tee gpurun_out/tunings/refine_toggle_w8.log
cp tools/_tunings/w8.refined.json gpurun_out/tunings/w8.refined.json
