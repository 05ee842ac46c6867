cd $GRAFT_REPO_ROOT
BASE=$GRAFT_REPO_ROOT/zsgnet-pytorch_amd/build/base/libzsg_base.so
python -m pytest tests -x -q -m gpu -k "maxpool or stem_bn or ssd or upsample" 2>&1 | tail -2
ZSG_LIB_PATH=$BASE python tools/dev_stem_bits.py gpurun_out/stem_base.pt 2>&1 | grep -v amdgpu.ids
python tools/dev_stem_bits.py gpurun_out/stem_new.pt gpurun_out/stem_base.pt 2>&1 | grep -v amdgpu.ids
rm -f gpurun_out/stem_*.pt
bash tools/ab_lib_own.sh $BASE 3 2>&1 | tee gpurun_out/ab_round5b.txt
echo "--- SSD-VGG B=32"
AB_ARGS="--backbone ssd_vgg --bs 32" bash tools/ab_lib_own.sh $BASE 2 2>&1 | tee gpurun_out/ab_round5b_ssd.txt
