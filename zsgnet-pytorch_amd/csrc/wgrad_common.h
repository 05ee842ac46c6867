// wgrad_common.h — pieces shared by the direct (wgrad.hip) and the Winograd (winowg.hip) weight-gradient kernels:
// the parameter block and the deterministic slab reduction.
#pragma once
#include "common.h"

#define WG_PAD 4
#define ZSG_WG_MAX_JOBS 8      // jobs of one batched weight-gradient launch (zsg_conv_wgrad_wino_batched)

struct WgSegDev {
    int rows_y, rows_x, rows, kt0;
    int src_H, src_W, sy, sx;
    int out_W, osy, osx, opy, opx;
    int src_off, src_bstride, out_off, out_bstride;
    float inv_per, inv_rx;
};

struct WgParams {
    const float* src;
    const float* dy;
    float* dw;
    float* ws;        // split-K slabs [splits][N][ncols] (NULL when splits == 1 and the tile is written straight to dw)
    int accumulate;   // 1: dw += result (splits == 1 path only; the reduce kernel handles it otherwise)
    int C, N, src_ld, out_ld, wS, wC, wc0, wt_ld, nseg;
    int ncols, txn;
    int m_tiles, n_tiles, splits, kt_total, kt_chunk, bk;
    zsg_taps ty, tx;
    WgSegDev seg[ZSG_MAX_SEG];
};

// 24-bit integer multiply (v_mul_i32_i24 / v_mad_i32_i24: full rate; a 32-bit v_mul_lo_u32 is quarter rate, and next to fp32 MFMAs
// every VALU cycle of the K loop is paid in full, tools/ubench/mfma_coissue.hip).  Both operands must fit 24 signed bits — row and
// pixel counts, image strides and row pitches do (checked by the host); the 32-bit result is exact.
// (inline asm: the compiler lowers __mul24 with a wave-uniform operand to s_bfe_i32 + the quarter-rate v_mul_lo_u32.)
__device__ __forceinline__ int mul24(int a, int b_uniform) {     // a: per-lane, b_uniform: wave-uniform (kernel argument / segment field)
    int r;
    asm("v_mul_i32_i24 %0, %2, %1" : "=v"(r) : "v"(a), "s"(b_uniform));
    return r;
}

__device__ __forceinline__ int fdiv(int a, int d, float rcp) {   // a < 2^24, exact after one correction step
    int q = (int)((float)a * rcp);
    int r = a - mul24(q, d);
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

// One slab reduction: dw[n][col(q)] (+)= sum_s ws[s][n][q].
struct WgReduceJob {
    const float* ws;
    float* dw;
    int N, ncols, splits, C, txn, wS, wC, wc0, wt_ld, accumulate;
    int ty_w0, ty_wstep, tx_w0, tx_wstep;
    int kl;           // split lanes per element (16 | 4)
};

static inline int wg_reduce_kl(int N, int ncols, int splits) {
    return (splits >= 32 || (int64_t)N * (ncols / 4) < 65536) ? 16 : 4;
}
static inline int wg_reduce_blocks(int N, int ncols, int kl) { return (int)(((int64_t)N * (ncols / 4) + 256 / kl - 1) / (256 / kl)); }

// fills the job from the convolution descriptor (is_wino: the 3x3 Winograd kernel's slab layout = the direct kernel's with ncols = 9 C)
void wg_reduce_job_fill(WgReduceJob& j, const zsg_conv_desc* d, const float* ws, float* dw, int accumulate, int splits);
// one reduction as its own launch (zsg_conv_wgrad / zsg_conv_wgrad_wino); defined in wgrad.hip
int wg_reduce_launch(const WgReduceJob& j, hipStream_t st);
