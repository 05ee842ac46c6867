#!/bin/bash
# Developer tool: which part of the streaming 1x1 kernel (csrc/pw.hip) is the time?
#   tools/pw_ablation.sh build   (anywhere: cross-compiles)  -> zsgnet-pytorch_amd/build/abl/libzsg_pwabl<N>.so, pw.hip at -DPW_ABL=<N>
#   tools/pw_ablation.sh run     (GPU box)                    -> single-launch times with each library (tools/pw_bench.py)
# PW_ABL bits: 1 no MFMAs, 2 no output stores, 4 no fragment reads, 8 no source loads in the streaming loop (results are wrong,
# only the times mean something).
R=${GRAFT_REPO_ROOT:-/root/repo}; P=$R/zsgnet-pytorch_amd; OUT=$P/build/abl; mkdir -p $OUT
LIST=${ABLS:-0 1 2 3 5 8 10 15}
if [ "$1" = "build" ]; then
  FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -I$R/include -I$P/csrc -I$P/build -I/opt/rocm/include -Wno-unused-result -Wno-unused-value"
  OBJS=$(ls $P/build/*.o | grep -v pw.hip.o)
  # the product source carries no experiment switches: they live in tools/ablation/pw_abl.patch and are applied to a COPY here
  cp $P/csrc/pw.hip $OUT/pw_ablsrc.hip && patch -s $OUT/pw_ablsrc.hip $R/tools/ablation/pw_abl.patch || { echo "tools/ablation/pw_abl.patch no longer applies to csrc/pw.hip"; exit 1; }
  for n in $LIST; do
    ( /opt/rocm/bin/hipcc $FLAGS -DPW_ABL=$n -c $OUT/pw_ablsrc.hip -o $OUT/pw_abl$n.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libzsg_pwabl$n.so $OUT/pw_abl$n.o $OBJS -ldl && rm $OUT/pw_abl$n.o ) &
  done
  wait
  ls -la $OUT | grep pwabl
else
  for n in $LIST; do
    echo "PW_ABL=$n"
    ZSG_LIB_PATH=$OUT/libzsg_pwabl$n.so SHAPES="${SHAPES:-90000x64x256,90000x256x64}" python $R/tools/pw_bench.py 2>/dev/null | grep "^M=" | sed 's/.*| //; s/64x64[^p]*//; s/128x[^p]*//g'
  done
fi
