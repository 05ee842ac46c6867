#!/bin/bash
# Developer tool: which part of wino_wgrad_kernel's stage loop is the time?  The variants are made from a COPY of csrc/winowg.hip by text
# substitution (tools/winowg_ablate.py) — the product source carries no experiment switches.  Results of the variants are wrong: timing only.
#   tools/winowg_ablation.sh build   (anywhere)   -> zsgnet-pytorch_amd/build/abl/libzsg_wwabl_<name>.so
#   tools/winowg_ablation.sh run [shape:splits..] (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}; P=$R/zsgnet-pytorch_amd; OUT=$P/build/abl; mkdir -p $OUT
LIST="full noload notransform nomfma nofrag noepilogue nobarrier"
if [ "$1" = "build" ]; then
  FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -I$R/include -I$P/csrc -I$P/build -I/opt/rocm/include -Wno-unused-result -Wno-unused-value"
  OBJS=$(ls $P/build/*.o | grep -v winowg.hip.o)
  for n in $LIST; do
    ( python $R/tools/winowg_ablate.py $n > $OUT/winowg_$n.hip && /opt/rocm/bin/hipcc $FLAGS -c $OUT/winowg_$n.hip -o $OUT/winowg_$n.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libzsg_wwabl_$n.so $OUT/winowg_$n.o $OBJS -ldl && rm $OUT/winowg_$n.o $OUT/winowg_$n.hip ) &
  done
  wait
  ls $OUT
else
  shift
  for n in $LIST; do
    echo "$n: $(ZSG_LIB_PATH=$OUT/libzsg_wwabl_$n.so python $R/tools/bench_winowg.py "$@" 2>/dev/null | tail -1)"
  done
fi
