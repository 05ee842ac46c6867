#!/bin/bash
# Developer tool: which part of the Winograd K loop is the time?
#   tools/wino_ablation.sh build   (anywhere: cross-compiles)  -> zsgnet-pytorch_amd/build/abl/libzsg_wabl<N>.so, wino.hip at -DWN_ABL=<N>
#   tools/wino_ablation.sh run [shapes]   (GPU box)             -> tools/bench_wino.py with each library (times only; results are wrong)
# WN_ABL bits: 1 patch loads hit one cache line, 2 filter chunk always chunk 0, 4 no MFMAs, 8 no fragment reads, 16 no patch loads /
# transform, 32 no filter DMA.
R=${GRAFT_REPO_ROOT:-/root/repo}; P=$R/zsgnet-pytorch_amd; OUT=$P/build/abl; mkdir -p $OUT
LIST=${ABLS:-0 1 2 3 4 8 16 32 48 52 60}
if [ "$1" = "build" ]; then
  FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -I$R/include -I$P/csrc -I$P/build -I/opt/rocm/include -Wno-unused-result -Wno-unused-value -Wno-array-bounds"
  OBJS=$(ls $P/build/*.o | grep -v wino.hip.o)
  # the product source carries no experiment switches: they live in tools/ablation/wino_abl.patch and are applied to a COPY here
  cp $P/csrc/wino.hip $OUT/wino_ablsrc.hip && patch -s $OUT/wino_ablsrc.hip $R/tools/ablation/wino_abl.patch || { echo "tools/ablation/wino_abl.patch no longer applies to csrc/wino.hip"; exit 1; }
  for n in $LIST; do
    ( /opt/rocm/bin/hipcc $FLAGS -DWN_ABL=$n -c $OUT/wino_ablsrc.hip -o $OUT/wino_abl$n.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libzsg_wabl$n.so $OUT/wino_abl$n.o $OBJS -ldl && rm $OUT/wino_abl$n.o ) &
  done
  wait
  ls $OUT
else
  shift
  for n in $LIST; do
    echo "WN_ABL=$n"
    ZSG_LIB_PATH=$OUT/libzsg_wabl$n.so python $R/tools/bench_wino.py ${@:-l3_conv2 head} 2>/dev/null | grep -o "^[a-zA-Z0-9_]* \|wn\[[^]]*\] *[0-9.]*us" | tr '\n' ' '; echo
  done
fi
