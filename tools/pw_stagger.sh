#!/bin/bash
# Developer tool (GPU box): start offset of one wave of every SIMD pair in the streaming 1x1 kernel (csrc/pw.hip), single-launch times.
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in ${STAGGERS:-0 4,4 8,4 12,4 16,4 8,1 8,2}; do
  echo "ZSG_PW_STAGGER=$v"
  ZSG_PW_STAGGER=$v SHAPES="${SHAPES:-90000x64x256,90000x256x64,23104x128x512}" python $R/tools/pw_bench.py 2>/dev/null | grep "^M=" | sed 's/.*| //; s/64x64[^p]*//; s/128x[^p]*//g'
done
