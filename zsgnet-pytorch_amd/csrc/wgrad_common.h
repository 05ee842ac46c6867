// wgrad_common.h — pieces shared by the direct (wgrad.hip) and the Winograd (winowg.hip) weight-gradient kernels:
// the parameter block and the deterministic slab reduction.
#pragma once
#include "common.h"

#define WG_PAD 4

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

__device__ __forceinline__ int fdiv(int a, int d, float rcp) {   // a < 2^24, exact after one correction step
    int q = (int)((float)a * rcp);
    int r = a - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

// dw[n][col(q)] (+)= sum_s ws[s][n][q]   — fixed summation order (deterministic).  A block of 256 threads covers
// 256/KL float4 elements with KL "split lanes" each (lane l sums slabs l, l+KL, ...), then an LDS tree over the lanes, so
// many-split launches (small weights, huge pixel counts) are not one serial latency chain per element.
template <int KL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgParams p) {
    constexpr int EL = 256 / KL;
    __shared__ f32x4 sm[KL][EL];
    const int q4 = p.ncols / 4;
    const int64_t total = (int64_t)p.N * q4;
    const size_t slab = (size_t)p.N * p.ncols;
    const int el = threadIdx.x % EL, kl = threadIdx.x / EL;
    const int64_t i = (int64_t)blockIdx.x * EL + el;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int n = 0, q = 0;
    if (i < total) {
        n = (int)(i / q4);
        q = (int)(i % q4) * 4;
        const float* src = p.ws + (size_t)n * p.ncols + q;
#pragma unroll 4
        for (int k = kl; k < p.splits; k += KL) s += *(const f32x4*)(src + k * slab);
    }
    sm[kl][el] = s;
    __syncthreads();
    for (int o = KL / 2; o > 0; o >>= 1) {
        if (kl < o) sm[kl][el] += sm[kl + o][el];
        __syncthreads();
    }
    if (kl == 0 && i < total) {
        s = sm[0][el];
        const int tapi = q / p.C;
        const int c = q - tapi * p.C;
        const int jy = tapi / p.txn, jx = tapi - jy * p.txn;
        const int wr = p.ty.w0 + jy * p.ty.wstep, ws_ = p.tx.w0 + jx * p.tx.wstep;
        float* o = p.dw + (size_t)n * p.wt_ld + (wr * p.wS + ws_) * p.wC + p.wc0 + c;
        if (p.accumulate) s += *(const f32x4*)o;
        *(f32x4*)o = s;
    }
}

