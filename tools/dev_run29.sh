cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tunings
T=tools/_tunings
cp $T/seed.json $T/fwd.json
REFINE_NOISE_MS=0.012 python tools/refine_tuning.py $T/fwd.json $T/c1.json $T/c2.json $T/c3.json $T/c4.json $T/c5.json --toggle24 --toggle-w8 --objective-forward 2>&1 | tee gpurun_out/tunings/refine_forward.log
cp $T/fwd.refined.json gpurun_out/tunings/fwd.refined.json
