cd $GRAFT_REPO_ROOT
PREV=$GRAFT_REPO_ROOT/zsgnet-pytorch_amd/build/base/libzsg_prev.so
python -m pytest tests/test_gpu_ops.py tests/test_gpu_wino.py -x -q -m gpu 2>&1 | tail -3
export ZSG_DETERMINISTIC=1 ZSG_SHIPPED_TUNE=0 ZSG_TUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/bits_tune.json
ZSG_LIB_PATH=$PREV python tools/dev_step_bits.py gpurun_out/bits_prev.pt 2>&1 | grep -v amdgpu.ids | tail -3
python tools/dev_step_bits.py gpurun_out/bits_new.pt gpurun_out/bits_prev.pt 2>&1 | grep -v amdgpu.ids | tail -6
unset ZSG_DETERMINISTIC ZSG_SHIPPED_TUNE ZSG_TUNE_CACHE
rm -f gpurun_out/bits_*.pt
bash tools/ab_lib_own.sh $PREV 3 2>&1 | tee gpurun_out/ab_epilogue_stores.txt
