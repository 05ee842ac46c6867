/* C ABI of the F(4x4,3x3) experiment (tools/experiments/wino4): declared here, NOT in include/zsg.h — the product library does not
 * export these symbols (round 6).  Build: make -C zsgnet-pytorch_amd/csrc EXPERIMENTS=1 OUT=/tmp/libzsg_exp.so */
#ifndef ZSG_WINO4_H
#define ZSG_WINO4_H
#include "zsg.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Winograd F(4x4,3x3) (round 5; csrc/wino4.hip): 2.25 instead of 4 (F(2x2,3x3)) or 9 multiply-adds per (pixel, cin, cout) for the
 * 3x3 / stride 1 / pad 1 convolutions BEHIND the network's last BatchNorm — the pyramid's output convolutions and the shared head
 * (fpn_resnet.py:157-172, mdl.py:211-244), forward and data gradient.  Same descriptor and epilogue terms as zsg_conv_wino (bias,
 * add_src / accumulate, ReLU, float ReLU mask; no split-K, no BatchNorm partials: it is never offered in front of a BatchNorm — its
 * rounding error, 6.5e-6 of the output range per layer against 4.8e-7, is only admissible where nothing amplifies it;
 * profiles/r05_wino_f4_gate.txt).  `U` is the transformed filter image made by zsg_wino4_weights ([C/8][36][Npad][8] floats =
 * zsg_wino4_u_elems); jobs as zsg_wino_weights (one record per convolution; blocks = ceil(chunks * Npad * 8 / 256) each). */
int64_t zsg_wino4_u_elems(int32_t C, int32_t N);
int zsg_wino4_weights(const void* jobs_dev, int32_t njobs, int32_t total_blocks, void* stream);
int zsg_conv_wino4(const zsg_conv_desc* d, const float* src, const float* U, float* out, const float* bias, const float* add_src,
                   const float* mask_src, void* stream);
#ifdef __cplusplus
}
#endif
#endif
