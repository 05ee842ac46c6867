cd $GRAFT_REPO_ROOT
BASE=$GRAFT_REPO_ROOT/zsgnet-pytorch_amd/build/base/libzsg_base.so
python -m pytest tests/test_gpu_wino.py -x -q -k wgrad 2>&1 | tail -2
bash tools/ab_lib_own.sh $BASE 3 2>&1 | tee gpurun_out/ab_winowg_epi.txt
cd /tmp && export TMPDIR=/tmp
ZSG_TUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/ab/tune_new.json ZSG_SHIPPED_TUNE=0 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/trace1 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --other-configs off > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_overlap.py $(ls gpurun_out/trace1/*/*kernel_trace.csv | head -1) 3 --list > gpurun_out/trace1_list.txt 2>&1
tail -5 gpurun_out/trace1_list.txt
