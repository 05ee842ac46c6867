cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tunings
python tools/refine_tuning.py tools/_tunings/best2.json --toggle24 2>&1 | tee gpurun_out/tunings/refine_toggle24.log
cp tools/_tunings/best2.refined.json gpurun_out/tunings/best2.refined.json
