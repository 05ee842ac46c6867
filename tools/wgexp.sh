for dbg in 0 1 2 3; do echo "== ZSG_WG_DEBUG=$dbg"; ZSG_WG_DEBUG=$dbg python tools/bench_conv.py head3x3_38 l2_conv2 l3_conv1 l3_conv2 2>&1 | grep -v amdgpu | sed 's/.*| wgrad/wgrad/'; done
