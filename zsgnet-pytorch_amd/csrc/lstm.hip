// lstm.hip — recurrent part of the BiLSTM query encoder (one workgroup per query, 4H threads).
// The input projections x_t W_ih^T + b_ih come from the MFMA GEMM (zsg_conv_igemm with a 1x1 "conv"); this file
// runs the 20 sequential [4H x H] mat-vecs with the recurrent weights held in registers (one gate row per thread),
// and the BPTT sweep that produces d(gate pre-activations); weight gradients are again MFMA GEMMs (zsg_conv_wgrad).
#include "common.h"

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// rank of sample b in a stable descending sort of the lengths (reference mdl.py:309 qlens.sort(0, descending=True))
__device__ __forceinline__ int sorted_rank(const float* __restrict__ qlens, int B, int b) {
    const float me = qlens[b];
    int r = 0;
    for (int j = 0; j < B; ++j) {
        const float o = qlens[j];
        r += (o > me) || (o == me && j < b);
    }
    return r;
}

template <int H, bool REG>
__global__ __launch_bounds__(4 * H) void lstm_fwd_kernel(const float* __restrict__ gin, const float* __restrict__ w_hh,
                                                         const float* __restrict__ b_hh, const float* __restrict__ h0,
                                                         const float* __restrict__ c0, const float* __restrict__ qlens,
                                                         const float* __restrict__ lens, int B, int T, float* __restrict__ gates,
                                                         float* __restrict__ cst, float* __restrict__ hprev, float* __restrict__ we,
                                                         int we_ld, int we_off) {
    __shared__ __attribute__((aligned(16))) float hs[H];
    __shared__ float gs[4 * H];
    const int b = blockIdx.x;
    const int j = threadIdx.x;          // gate row
    const int rank = sorted_rank(qlens, B, b);
    int len = lens ? (int)lens[b] : 1;
    len = len < 0 ? 0 : (len > T ? T : len);
    float w[REG ? H : 1];
    if (REG) {
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            const f32x4 v = *(const f32x4*)(w_hh + (size_t)j * H + k);
            w[k] = v[0];
            w[k + 1] = v[1];
            w[k + 2] = v[2];
            w[k + 3] = v[3];
        }
    }
    const float bias = b_hh[j];
    float c = 0.f;
    if (j < H) {
        hs[j] = h0[rank * H + j];
        c = c0[rank * H + j];
    }
    __syncthreads();
    for (int t = 0; t < len; ++t) {
        float pre = gin[((size_t)b * T + t) * 4 * H + j] + bias;
        if (REG) {
#pragma unroll
            for (int k = 0; k < H; k += 4) {
                const f32x4 hv = *(const f32x4*)(hs + k);
                pre += w[k] * hv[0] + w[k + 1] * hv[1] + w[k + 2] * hv[2] + w[k + 3] * hv[3];
            }
        } else {
            // (not fully unrolled: hipcc hoists all H / 4 row loads in front of the sum — 256 registers at H = 256, 162 of them spilled
            // at this block's 128-register budget; 8 loads in flight are plenty for an L2-resident row)
#pragma unroll 8
            for (int k = 0; k < H; k += 4) {
                const f32x4 hv = *(const f32x4*)(hs + k);
                const f32x4 wv = *(const f32x4*)(w_hh + (size_t)j * H + k);
                pre += wv[0] * hv[0] + wv[1] * hv[1] + wv[2] * hv[2] + wv[3] * hv[3];
            }
        }
        const float act = (j >= 2 * H && j < 3 * H) ? tanhf(pre) : sigmoidf_(pre);
        gs[j] = act;
        gates[((size_t)b * T + t) * 4 * H + j] = act;
        if (j < H) hprev[((size_t)b * T + t) * H + j] = hs[j];
        __syncthreads();
        if (j < H) {
            c = gs[H + j] * c + gs[j] * gs[2 * H + j];
            const float h = gs[3 * H + j] * tanhf(c);
            cst[((size_t)b * T + t) * H + j] = c;
            hs[j] = h;
        }
        __syncthreads();
    }
    if (j < H) we[(size_t)b * we_ld + we_off + j] = hs[j];
}

template <int H>
__global__ __launch_bounds__(4 * H) void lstm_bwd_kernel(const float* __restrict__ dwe, int we_ld, int we_off,
                                                         const float* __restrict__ w_hh, const float* __restrict__ gates,
                                                         const float* __restrict__ cst, const float* __restrict__ c0,
                                                         const float* __restrict__ qlens, const float* __restrict__ lens, int B, int T,
                                                         float* __restrict__ dgates) {
    __shared__ float dpre[4 * H];
    __shared__ float part[4][H];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int u = tid % H, q = tid / H;
    const int rank = sorted_rank(qlens, B, b);
    int len = lens ? (int)lens[b] : 1;
    len = len < 0 ? 0 : (len > T ? T : len);
    float dh = 0.f, dc = 0.f;
    if (q == 0) dh = dwe[(size_t)b * we_ld + we_off + u];
    for (int t = T - 1; t >= len; --t) dgates[((size_t)b * T + t) * 4 * H + tid] = 0.f;
    for (int t = len - 1; t >= 0; --t) {
        if (q == 0) {
            const float* g = gates + ((size_t)b * T + t) * 4 * H;
            const float gi = g[u], gf = g[H + u], gg = g[2 * H + u], go = g[3 * H + u];
            const float c = cst[((size_t)b * T + t) * H + u];
            const float cp = t > 0 ? cst[((size_t)b * T + t - 1) * H + u] : c0[rank * H + u];
            const float tc = tanhf(c);
            const float d_o = dh * tc;
            dc += dh * go * (1.f - tc * tc);
            const float d_i = dc * gg, d_g = dc * gi, d_f = dc * cp;
            dpre[u] = d_i * gi * (1.f - gi);
            dpre[H + u] = d_f * gf * (1.f - gf);
            dpre[2 * H + u] = d_g * (1.f - gg * gg);
            dpre[3 * H + u] = d_o * go * (1.f - go);
            dc = dc * gf;
        }
        __syncthreads();
        dgates[((size_t)b * T + t) * 4 * H + tid] = dpre[tid];
        float s = 0.f;
        for (int k = 0; k < H; ++k) s += w_hh[(size_t)(q * H + k) * H + u] * dpre[q * H + k];
        part[q][u] = s;
        __syncthreads();
        if (q == 0) dh = part[0][u] + part[1][u] + part[2][u] + part[3][u];
        __syncthreads();
    }
}

__global__ void lstm_gather_last_kernel(const float* __restrict__ qvec, const float* __restrict__ qlens, int B, int T, int E,
                                        float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * E) return;
    const int b = i / E, e = i % E;
    int t = (int)qlens[b] - 1;
    t = t < 0 ? 0 : (t >= T ? T - 1 : t);
    out[i] = qvec[((size_t)b * T + t) * E + e];
}

extern "C" int zsg_lstm_gather_last(const float* qvec, const float* qlens, int32_t B, int32_t T, int32_t E, float* out, void* stream) {
    ZSG_REQUIRE(qvec && qlens && out && B > 0 && T > 0 && E > 0, "lstm_gather_last: bad argument");
    ZSG_LAUNCH(lstm_gather_last_kernel, dim3(cdiv((int64_t)B * E, 256)), dim3(256), 0, (hipStream_t)stream, qvec, qlens, B, T, E, out);
    ZSG_CHECK_LAUNCH("lstm_gather_last");
    return 0;
}

extern "C" int zsg_lstm_fwd(const float* gin, const float* w_hh, const float* b_hh, const float* h0, const float* c0,
                            const float* qlens_rank, const float* lens, int32_t B, int32_t T, int32_t H, float* gates, float* cst,
                            float* hprev, float* we, int32_t we_ld, int32_t we_off, void* stream) {
    ZSG_REQUIRE(gin && w_hh && b_hh && h0 && c0 && qlens_rank && gates && cst && hprev && we && B > 0 && T > 0, "lstm_fwd: null argument");
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("lstm_fwd", st, 2.0 * B * T * 4 * H * H, 0);
#define ZSG_LSTM_FWD(HH, REG)                                                                                                   \
    ZSG_LAUNCH((lstm_fwd_kernel<HH, REG>), dim3(B), dim3(4 * HH), 0, st, gin, w_hh, b_hh, h0, c0, qlens_rank, lens, B, T, \
                       gates, cst, hprev, we, we_ld, we_off)
    if (H == 128) ZSG_LSTM_FWD(128, true);
    else if (H == 64) ZSG_LSTM_FWD(64, true);
    else if (H == 256) ZSG_LSTM_FWD(256, false);
    else if (H == 32) ZSG_LSTM_FWD(32, true);
    else ZSG_FAIL(-1, "lstm_fwd: lstm_dim %d not supported (32/64/128/256)", H);
#undef ZSG_LSTM_FWD
    ZSG_CHECK_LAUNCH("lstm_fwd");
    return 0;
}

extern "C" int zsg_lstm_bwd(const float* dwe, int32_t we_ld, int32_t we_off, const float* w_hh, const float* gates, const float* cst,
                            const float* c0, const float* qlens_rank, const float* lens, int32_t B, int32_t T, int32_t H, float* dgates,
                            void* stream) {
    ZSG_REQUIRE(dwe && w_hh && gates && cst && c0 && qlens_rank && dgates && B > 0 && T > 0, "lstm_bwd: null argument");
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("lstm_bwd", st, 2.0 * B * T * 4 * H * H, 0);
#define ZSG_LSTM_BWD(HH) \
    ZSG_LAUNCH((lstm_bwd_kernel<HH>), dim3(B), dim3(4 * HH), 0, st, dwe, we_ld, we_off, w_hh, gates, cst, c0, qlens_rank, lens, B, T, dgates)
    if (H == 128) ZSG_LSTM_BWD(128);
    else if (H == 64) ZSG_LSTM_BWD(64);
    else if (H == 256) ZSG_LSTM_BWD(256);
    else if (H == 32) ZSG_LSTM_BWD(32);
    else ZSG_FAIL(-1, "lstm_bwd: lstm_dim %d not supported (32/64/128/256)", H);
#undef ZSG_LSTM_BWD
    ZSG_CHECK_LAUNCH("lstm_bwd");
    return 0;
}
