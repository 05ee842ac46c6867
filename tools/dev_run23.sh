cd $GRAFT_REPO_ROOT
BASE=$GRAFT_REPO_ROOT/zsgnet-pytorch_amd/build/base/libzsg_base.so
python -m pytest tests/test_gpu_ops.py -x -q -k "stem_bn_relu_maxpool" 2>&1 | tail -2
ZSG_LIB_PATH=$BASE python tools/dev_stem_bits.py gpurun_out/stem_base.pt 2>&1 | grep -v amdgpu.ids
python tools/dev_stem_bits.py gpurun_out/stem_new.pt gpurun_out/stem_base.pt 2>&1 | grep -v amdgpu.ids
rm -f gpurun_out/stem_*.pt
