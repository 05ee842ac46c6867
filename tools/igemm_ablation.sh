#!/bin/bash
# Developer tool: which part of the implicit-GEMM K loop is the time?
#   tools/igemm_ablation.sh build   (anywhere: cross-compiles)  -> zsgnet-pytorch_amd/build/abl/libzsg_abl<N>.so, igemm.hip at -DIG_ABL=<N>
#   tools/igemm_ablation.sh run     (GPU box)                    -> single-launch times with each library
# Compile-time ablation (IG_ABL bits: 1 no MFMAs, 2 no global loads in the K loop, 4 no LDS stores, 8 no fragment reads, 16 no
# barrier): no runtime branches in the loop; the results of an ablated kernel are wrong, only its time is meaningful.
R=${GRAFT_REPO_ROOT:-/root/repo}; P=$R/zsgnet-pytorch_amd; OUT=$P/build/abl; mkdir -p $OUT
LIST=${ABLS:-0 1 2 4 8 16 3 24 28 31}
if [ "$1" = "build" ]; then
  FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -I$R/include -I$P/csrc -I$P/build -I/opt/rocm/include -Wno-unused-result -Wno-unused-value"
  OBJS=$(ls $P/build/*.o | grep -v igemm.hip.o)
  # the product source carries no experiment switches: they live in tools/ablation/igemm_abl.patch and are applied to a COPY here
  cp $P/csrc/igemm.hip $OUT/igemm_ablsrc.hip && patch -s $OUT/igemm_ablsrc.hip $R/tools/ablation/igemm_abl.patch || { echo "tools/ablation/igemm_abl.patch no longer applies to csrc/igemm.hip"; exit 1; }
  for n in $LIST; do
    ( /opt/rocm/bin/hipcc $FLAGS -DIG_ABL=$n -c $OUT/igemm_ablsrc.hip -o $OUT/igemm_abl$n.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libzsg_abl$n.so $OUT/igemm_abl$n.o $OBJS -ldl && rm $OUT/igemm_abl$n.o ) &
  done
  wait
  ls -la $OUT
else
  for n in $LIST; do
    echo "IG_ABL=$n"
    ZSG_LIB_PATH=$OUT/libzsg_abl$n.so SHAPES="${SHAPES:-1600x512,5776x256}" KS="${KS:-1024,2048}" python $R/tools/igemm_model.py 2>/dev/null | grep "^M="
  done
fi
