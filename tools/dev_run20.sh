cd $GRAFT_REPO_ROOT
BASE=zsgnet-pytorch_amd/build/base/libzsg_base.so
python -m pytest tests/test_gpu_wino.py -x -q -k wgrad 2>&1 | tail -3
ZSG_LIB_PATH=$BASE python tools/dev_ww_bits.py gpurun_out/ww_base.pt 2>&1 | tail -2
python tools/dev_ww_bits.py gpurun_out/ww_new.pt gpurun_out/ww_base.pt 2>&1 | tail -5
ZSG_WW_XMAP=1 python tools/dev_ww_bits.py gpurun_out/ww_new2.pt gpurun_out/ww_base.pt 2>&1 | tail -5
for i in 1 2; do
echo "base: $(ZSG_LIB_PATH=$BASE python tools/bench_winowg.py 2>/dev/null | tail -1)"
echo "new : $(python tools/bench_winowg.py 2>/dev/null | tail -1)"
echo "xmap: $(ZSG_WW_XMAP=1 python tools/bench_winowg.py 2>/dev/null | tail -1)"
done
rm -f gpurun_out/ww_*.pt
