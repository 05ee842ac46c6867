// bn.hip — train-mode BatchNorm2d over NHWC [rows][C] (+ ReLU, + residual), forward and backward.  HBM-bound.
// Each thread owns 4 consecutive channels (16-byte loads); a block spans up to 256 channels x a chunk of rows.
// Statistics: per-chunk fp32 partial (sum, sum of squares) -> a finalize kernel merges them in fp64.
#include <stdlib.h>

#include "common.h"

ZSG_DEFINE_PRIO_FLAG()

struct BnGeom {
    int lanes;      // float4 lanes across channels inside a block (<= 64)
    int rowlanes;   // 256 / lanes
    int slabs;      // channel slabs of lanes*4
    int rpb;        // rows per block
    int chunks;     // row chunks
};

static BnGeom bn_geom(int64_t rows, int C) {
    BnGeom g;
    const int c4 = C / 4;
    g.lanes = c4 >= 64 ? 64 : (c4 >= 32 ? 32 : (c4 >= 16 ? 16 : (c4 >= 8 ? 8 : (c4 >= 4 ? 4 : (c4 >= 2 ? 2 : 1)))));
    g.rowlanes = 256 / g.lanes;
    g.slabs = cdiv(c4, g.lanes);
    int64_t rpb = (rows * g.slabs + 1023) / 1024;
    if (rpb < 4 * g.rowlanes) rpb = 4 * g.rowlanes;
    g.rpb = (int)rpb;
    g.chunks = cdiv(rows, g.rpb);
    return g;
}

extern "C" size_t zsg_bn_workspace_bytes(int64_t rows, int32_t C) {
    BnGeom g = bn_geom(rows, C);
    return ((size_t)g.chunks * 2 * C + 2 * (size_t)C) * sizeof(float);
}

// MODE 0: (sum x, sum x^2).  MODE 1: backward (sum g, sum g*xhat) with g = dout * (relu_out > 0).
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, const float* __restrict__ dout,
                                                         const float* __restrict__ relu_out, const uint8_t* __restrict__ relu_mask,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         int64_t rows, int C, int lanes, int rpb, float* __restrict__ part) {
    ZSG_SET_MAIN_PRIO();
    __shared__ f32x4 red[2][256];
    const int rowlanes = 256 / lanes;
    const int l = threadIdx.x % lanes, rl = threadIdx.x / lanes;
    const int c = (blockIdx.y * lanes + l) * 4;
    const bool cok = c < C;
    const int64_t r_begin = (int64_t)blockIdx.x * rpb;
    const int64_t r_end = min(rows, r_begin + (int64_t)rpb);
    f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
    f32x4 mu = {0, 0, 0, 0}, is = {0, 0, 0, 0};
    if (MODE == 1 && cok) {
        mu = *(const f32x4*)(mean + c);
        is = *(const f32x4*)(invstd + c);
    }
    if (cok) {
        for (int64_t r = r_begin + rl; r < r_end; r += rowlanes) {
            const f32x4 v = *(const f32x4*)(x + r * C + c);
            if (MODE == 0) {
                s0 += v;
                s1 += v * v;
            } else {
                f32x4 g = *(const f32x4*)(dout + r * C + c);
                if (relu_mask) {
                    const unsigned m = relu_mask[(r * C + c) >> 2];
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] = ((m >> e) & 1u) ? g[e] : 0.f;
                } else if (relu_out) {
                    const f32x4 o = *(const f32x4*)(relu_out + r * C + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
                }
                s0 += g;
                s1 += g * ((v - mu) * is);
            }
        }
    }
    red[0][threadIdx.x] = s0;
    red[1][threadIdx.x] = s1;
    __syncthreads();
    if (rl == 0 && cok) {
        for (int k = 1; k < rowlanes; ++k) {
            s0 += red[0][k * lanes + l];
            s1 += red[1][k * lanes + l];
        }
        float* o = part + (size_t)blockIdx.x * 2 * C;
        *(f32x4*)(o + c) = s0;
        *(f32x4*)(o + C + c) = s1;
    }
}

// Finalize: ONE BLOCK (BN_FW waves) per 4 channels.  Every lane issues all its row loads (BN_FU rows x 2 x 16 bytes) before it adds
// anything — one memory round trip for up to 64 * BN_FW * BN_FU = 768 partial rows, i.e. every BatchNorm of the 300^2 network —
// accumulates in fp64, then a transpose-reduce over the wave (xor 1 / 2 / 4 halve the value set: 8 + 4 + 2 + 1 + 3 exchanges
// instead of 8 x 6) and ONE barrier for the BN_FW partial sums.  History (tools/ubench/bn_finalize.hip, launch after a producer
// that rewrites the rows, 704 rows x 64 channels): 8-channel x 32-lane block with an LDS tree 5.7 us (five barriers); one wave per
// 4 channels with `unroll 4` loads and full butterflies 6.2 us; this 3.2 us, against 2.4 us for an empty launch.  88 such
// launches sit on the step's dependent chain.
#define BN_FW 4                       // waves per block
#define BN_FU 3                       // rows in flight per lane
// FW: waves per block.  BN_FW (4) everywhere but for thousands of partial rows (the stem's 5625 at the bench shape: 19.9 us with 4
// waves — eight dependent memory round trips per lane — against ~7 us with 16 waves and two).
#define BN_FW_MANY 16
#define BN_MANY_ROWS 1536
template <int FW>
__device__ __forceinline__ void bn_reduce_partials(const float* __restrict__ part, int chunks, int C, int c, double& se, double& sse) {
    __shared__ double red[FW][8];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k0 = t; k0 < chunks; k0 += 64 * FW * BN_FU) {
        f32x4 a[BN_FU], b[BN_FU];
#pragma unroll
        for (int u = 0; u < BN_FU; ++u) {
            const int k = k0 + u * 64 * FW;
            const int kk = k < chunks ? k : chunks - 1;
            a[u] = *(const f32x4*)(part + (size_t)kk * 2 * C + c);
            b[u] = *(const f32x4*)(part + (size_t)kk * 2 * C + C + c);
        }
#pragma unroll
        for (int u = 0; u < BN_FU; ++u) {
            const bool ok = k0 + u * 64 * FW < chunks;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] += ok ? (double)a[u][e] : 0.0;
                v[4 + e] += ok ? (double)b[u][e] : 0.0;
            }
        }
    }
    // after the exchange with lane ^ 1 / ^ 2 / ^ 4 a lane keeps half of its values, each summed with the partner's copy
    double w4[4], w2[2], w1;
    {
        const bool hi = lane & 1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double keep = hi ? v[4 + e] : v[e], give = hi ? v[e] : v[4 + e];
            w4[e] = keep + __shfl_xor(give, 1, 64);
        }
    }
    {
        const bool hi = lane & 2;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const double keep = hi ? w4[2 + e] : w4[e], give = hi ? w4[e] : w4[2 + e];
            w2[e] = keep + __shfl_xor(give, 2, 64);
        }
    }
    {
        const bool hi = lane & 4;
        const double keep = hi ? w2[1] : w2[0], give = hi ? w2[0] : w2[1];
        w1 = keep + __shfl_xor(give, 4, 64);
    }
    // lane l holds value (l&1)*4 + ((l>>1)&1)*2 + ((l>>2)&1) of its 8-lane group: add the 8 groups, then the waves
    w1 += __shfl_xor(w1, 8, 64);
    w1 += __shfl_xor(w1, 16, 64);
    w1 += __shfl_xor(w1, 32, 64);
    if (lane < 8) red[wave][(lane & 1) * 4 + ((lane >> 1) & 1) * 2 + ((lane >> 2) & 1)] = w1;
    __syncthreads();
    se = sse = 0;
    if (t < 4) {
#pragma unroll
        for (int w = 0; w < FW; ++w) {
            se += red[w][t];
            sse += red[w][4 + t];
        }
    }
}

template <int FW>
__global__ __launch_bounds__(64 * FW) void bn_stats_finalize_kernel(const float* __restrict__ part, int chunks, int C, int64_t rows,
                                                                    float* mean, float* invstd, float* rmean, float* rvar,
                                                                    float momentum, float eps) {
    ZSG_SET_MAIN_PRIO();
    const int c = blockIdx.x * 4;
    double se, sse;
    bn_reduce_partials<FW>(part, chunks, C, c, se, sse);
    const int e = threadIdx.x;
    if (e >= 4) return;                            // threads 0..3 write one channel each
    const double n = (double)rows;
    const double m = se / n;
    double var = sse / n - m * m;
    if (var < 0) var = 0;
    mean[c + e] = (float)m;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    invstd[c + e] = is;
    if (rmean) rmean[c + e] = (1.f - momentum) * rmean[c + e] + momentum * (float)m;
    if (rvar) rvar[c + e] = (1.f - momentum) * rvar[c + e] + momentum * (float)(n > 1 ? var * n / (n - 1) : var);
}

__global__ void bn_eval_stats_kernel(const float* rmean, const float* rvar, int C, float eps, float* mean, float* invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = rmean[c];
    invstd[c] = 1.0f / sqrtf(rvar[c] + eps);
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, int64_t rows, int C, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ residual, int relu,
                                                       float* __restrict__ out, uint8_t* __restrict__ relu_mask, int lanes, int rpb) {
    ZSG_SET_MAIN_PRIO();
    const int rowlanes = 256 / lanes;
    const int l = threadIdx.x % lanes, rl = threadIdx.x / lanes;
    const int c = (blockIdx.y * lanes + l) * 4;
    if (c >= C) return;
    const f32x4 mu = *(const f32x4*)(mean + c);
    const f32x4 sc = *(const f32x4*)(invstd + c) * *(const f32x4*)(gamma + c);
    const f32x4 be = *(const f32x4*)(beta + c);
    const int64_t r_begin = (int64_t)blockIdx.x * rpb;
    const int64_t r_end = min(rows, r_begin + (int64_t)rpb);
    for (int64_t r = r_begin + rl; r < r_end; r += rowlanes) {
        f32x4 v = (*(const f32x4*)(x + r * C + c) - mu) * sc + be;
        if (residual) v += *(const f32x4*)(residual + r * C + c);
        if (relu) {
            if (relu_mask)          // 4 mask bits per 16-byte group: the backward reads this byte instead of the output
                relu_mask[(r * C + c) >> 2] = (uint8_t)((v[0] > 0.f) | ((v[1] > 0.f) << 1) | ((v[2] > 0.f) << 2) | ((v[3] > 0.f) << 3));
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        *(f32x4*)(out + r * C + c) = v;
    }
}

// coef[0][c] = sum g / n ; coef[1][c] = sum g*xhat / n ; dgamma/dbeta written or accumulated.
__global__ __launch_bounds__(64 * BN_FW) void bn_bwd_finalize_kernel(const float* __restrict__ part, int chunks, int C, int64_t rows,
                                                                     float* coef, float* dgamma, float* dbeta, int accumulate) {
    ZSG_SET_MAIN_PRIO();
    const int c = blockIdx.x * 4;
    double se, sse;
    bn_reduce_partials<BN_FW>(part, chunks, C, c, se, sse);
    const int e = threadIdx.x;
    if (e >= 4) return;
    coef[c + e] = (float)(se / (double)rows);
    coef[C + c + e] = (float)(sse / (double)rows);
    if (dbeta) dbeta[c + e] = (accumulate ? dbeta[c + e] : 0.f) + (float)se;
    if (dgamma) dgamma[c + e] = (accumulate ? dgamma[c + e] : 0.f) + (float)sse;
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ relu_out,
                                                           const uint8_t* __restrict__ relu_mask,
                                                           const float* __restrict__ x, int64_t rows, int C,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ coef,
                                                           float* __restrict__ dx, float* __restrict__ g_out, int lanes, int rpb) {
    ZSG_SET_MAIN_PRIO();
    const int rowlanes = 256 / lanes;
    const int l = threadIdx.x % lanes, rl = threadIdx.x / lanes;
    const int c = (blockIdx.y * lanes + l) * 4;
    if (c >= C) return;
    const f32x4 mu = *(const f32x4*)(mean + c);
    const f32x4 is = *(const f32x4*)(invstd + c);
    const f32x4 sc = is * *(const f32x4*)(gamma + c);
    const f32x4 c1 = *(const f32x4*)(coef + c);
    const f32x4 c2 = *(const f32x4*)(coef + C + c);
    const int64_t r_begin = (int64_t)blockIdx.x * rpb;
    const int64_t r_end = min(rows, r_begin + (int64_t)rpb);
    for (int64_t r = r_begin + rl; r < r_end; r += rowlanes) {
        f32x4 g = *(const f32x4*)(dout + r * C + c);
        if (relu_mask) {
            const unsigned m = relu_mask[(r * C + c) >> 2];
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = ((m >> e) & 1u) ? g[e] : 0.f;
        } else if (relu_out) {
            const f32x4 o = *(const f32x4*)(relu_out + r * C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
        }
        const f32x4 xh = (*(const f32x4*)(x + r * C + c) - mu) * is;
        if (g_out) *(f32x4*)(g_out + r * C + c) = g;
        *(f32x4*)(dx + r * C + c) = sc * (g - c1 - xh * c2);
    }
}

// ---- finalize-free variants for FEW partial rows --------------------------------------------------------------------------
// When the producer left at most BN_INL_MAX partial rows (small tensors: layer3 / layer4 / pyramid-sized maps), every block
// of the streaming pass reduces them for its own channels itself (fp64, fixed order: deterministic) instead of waiting for a
// separate finalize launch — one dependent ~6 us launch and one launch gap less per such BatchNorm.  Per block that
// is chunks x 2 coalesced 16-byte loads per thread from L2.
#define BN_INL_MAX 64

__device__ __forceinline__ void bn_reduce_rows(const float* __restrict__ part, int chunks, int C, int c, double (&s)[4], double (&ss)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = ss[e] = 0;
#pragma unroll 8
    for (int k = 0; k < chunks; ++k) {
        const f32x4 a = *(const f32x4*)(part + (size_t)k * 2 * C + c);
        const f32x4 b = *(const f32x4*)(part + (size_t)k * 2 * C + C + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s[e] += (double)a[e];
            ss[e] += (double)b[e];
        }
    }
}

__global__ __launch_bounds__(256) void bn_apply_inl_kernel(const float* __restrict__ x, int64_t rows, int C, const float* __restrict__ part,
                                                           int chunks, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ residual, int relu, float* __restrict__ out,
                                                           uint8_t* __restrict__ relu_mask, float* __restrict__ mean_out,
                                                           float* __restrict__ invstd_out, float* rmean, float* rvar, float momentum,
                                                           float eps, int lanes, int rpb) {
    ZSG_SET_MAIN_PRIO();
    const int rowlanes = 256 / lanes;
    const int l = threadIdx.x % lanes, rl = threadIdx.x / lanes;
    const int c = (blockIdx.y * lanes + l) * 4;
    // the block's row lanes share the reduction of the partial rows (row lane rl takes rows rl, rl + rowlanes, ...) and combine
    // through LDS in row-lane order (fixed: deterministic): a rowlanes-times shorter chain of dependent L2 loads than every
    // thread walking all rows (the prologue was most of these small launches)
    __shared__ double sh[8][256];
    double s[4], ss[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = ss[e] = 0;
    if (c < C) {
        // every lane issues ALL its row loads of a batch (8 rows x 2 x 16 bytes) before it adds anything: one L2 round trip per batch
        // (<= 2 batches for BN_INL_MAX rows at the >= 4 row lanes of a block) instead of one per unrolled group of four — this
        // prologue, not the streaming body, was most of the small apply launches (11.5 us against 6.9 us for the plain apply of the
        // same 5776 x 256 tensor, profiles/r05_fwd_listing.txt); same summation order as before
        constexpr int U = 8;
        for (int k0 = rl; k0 < chunks; k0 += U * rowlanes) {
            f32x4 a[U], b[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + u * rowlanes;
                const int kk = k < chunks ? k : chunks - 1;
                a[u] = *(const f32x4*)(part + (size_t)kk * 2 * C + c);
                b[u] = *(const f32x4*)(part + (size_t)kk * 2 * C + C + c);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = k0 + u * rowlanes < chunks;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s[e] += ok ? (double)a[u][e] : 0.0;
                    ss[e] += ok ? (double)b[u][e] : 0.0;
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sh[e][threadIdx.x] = s[e];
        sh[4 + e][threadIdx.x] = ss[e];
    }
    __syncthreads();
    if (c >= C) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        double a = 0, b = 0;
        for (int r = 0; r < rowlanes; ++r) {
            a += sh[e][r * lanes + l];
            b += sh[4 + e][r * lanes + l];
        }
        s[e] = a;
        ss[e] = b;
    }
    const double n = (double)rows;
    f32x4 mu, is;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double m = s[e] / n;
        double var = ss[e] / n - m * m;
        if (var < 0) var = 0;
        mu[e] = (float)m;
        is[e] = (float)(1.0 / sqrt(var + (double)eps));
        if (blockIdx.x == 0 && rl == 0) {              // one thread per channel publishes the statistics (backward needs them)
            mean_out[c + e] = mu[e];
            invstd_out[c + e] = is[e];
            if (rmean) rmean[c + e] = (1.f - momentum) * rmean[c + e] + momentum * (float)m;
            if (rvar) rvar[c + e] = (1.f - momentum) * rvar[c + e] + momentum * (float)(n > 1 ? var * n / (n - 1) : var);
        }
    }
    const f32x4 sc = is * *(const f32x4*)(gamma + c);
    const f32x4 be = *(const f32x4*)(beta + c);
    const int64_t r_begin = (int64_t)blockIdx.x * rpb;
    const int64_t r_end = min(rows, r_begin + (int64_t)rpb);
    for (int64_t r = r_begin + rl; r < r_end; r += rowlanes) {
        f32x4 v = (*(const f32x4*)(x + r * C + c) - mu) * sc + be;
        if (residual) v += *(const f32x4*)(residual + r * C + c);
        if (relu) {
            if (relu_mask)
                relu_mask[(r * C + c) >> 2] = (uint8_t)((v[0] > 0.f) | ((v[1] > 0.f) << 1) | ((v[2] > 0.f) << 2) | ((v[3] > 0.f) << 3));
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        *(f32x4*)(out + r * C + c) = v;
    }
}

// (ZSG_BN_INL_MAX overrides the threshold for A/B measurements; the kernel itself handles any row count)
static int bn_inl_max() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ZSG_BN_INL_MAX");
        v = (e && atoi(e) > 0) ? atoi(e) : BN_INL_MAX;
        if (v > 1024) v = 1024;
    }
    return v;
}
extern "C" int32_t zsg_bn_inline_max_chunks(void) { return bn_inl_max(); }

extern "C" int zsg_bn_apply_from_partials(const float* x, int64_t rows, int32_t C, const float* partials, int32_t chunks, const float* gamma,
                                          const float* beta, const float* residual, int32_t relu, float* out, uint8_t* relu_mask,
                                          float* mean, float* invstd, float* running_mean, float* running_var, float momentum,
                                          float eps, void* stream) {
    ZSG_REQUIRE(x && partials && gamma && beta && out && mean && invstd && rows > 0 && C > 0 && (C % 4) == 0 && chunks > 0 && chunks <= bn_inl_max(),
                "bn_apply_from_partials: bad argument (chunks=%d, at most %d)", chunks, bn_inl_max());
    BnGeom g = bn_geom(rows, C);
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("bn_apply", st, 0, (double)rows * C * (4 * (residual ? 3 : 2) + (relu_mask && relu ? 0.25 : 0)));
    ZSG_LAUNCH(bn_apply_inl_kernel, dim3(g.chunks, g.slabs), dim3(256), 0, st, x, rows, C, partials, chunks, gamma, beta, residual, relu,
                       out, relu_mask, mean, invstd, running_mean, running_var, momentum, eps, g.lanes, g.rpb);
    ZSG_CHECK_LAUNCH("bn_apply_from_partials");
    return 0;
}

extern "C" int zsg_bn_stats(const float* x, int64_t rows, int32_t C, float* mean, float* invstd, float* running_mean,
                            float* running_var, float momentum, float eps, void* ws, size_t ws_bytes, void* stream) {
    ZSG_REQUIRE(x && mean && invstd && ws && rows > 0 && C > 0 && (C % 4) == 0, "bn_stats: bad argument (C=%d rows=%lld)", C, (long long)rows);
    if (ws_bytes < zsg_bn_workspace_bytes(rows, C)) ZSG_FAIL(-2, "bn_stats: workspace too small");
    BnGeom g = bn_geom(rows, C);
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("bn_stats", st, 0, (double)rows * C * 4);
    float* part = (float*)ws;
    ZSG_LAUNCH((bn_partial_kernel<0>), dim3(g.chunks, g.slabs), dim3(256), 0, st, x, nullptr, nullptr, nullptr, nullptr, nullptr,
                       rows, C, g.lanes, g.rpb, part);
    if (g.chunks > BN_MANY_ROWS)
        ZSG_LAUNCH(bn_stats_finalize_kernel<BN_FW_MANY>, dim3(C / 4), dim3(64 * BN_FW_MANY), 0, st, part, g.chunks, C, rows, mean, invstd,
                           running_mean, running_var, momentum, eps);
    else
        ZSG_LAUNCH(bn_stats_finalize_kernel<BN_FW>, dim3(C / 4), dim3(64 * BN_FW), 0, st, part, g.chunks, C, rows, mean, invstd,
                           running_mean, running_var, momentum, eps);
    ZSG_CHECK_LAUNCH("bn_stats");
    return 0;
}

// Finalize only: the (sum, sum^2) partials were produced by the convolution's epilogue (zsg_conv_igemm bn_partials).
extern "C" int zsg_bn_stats_from_partials(const float* partials, int32_t chunks, int64_t rows, int32_t C, float* mean, float* invstd,
                                          float* running_mean, float* running_var, float momentum, float eps, void* stream) {
    ZSG_REQUIRE(partials && mean && invstd && chunks > 0 && rows > 0 && C > 0 && (C % 4) == 0, "bn_stats_from_partials: bad argument");
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("bn_stats", st, 0, (double)chunks * C * 8);
    if (chunks > BN_MANY_ROWS)
        ZSG_LAUNCH(bn_stats_finalize_kernel<BN_FW_MANY>, dim3(C / 4), dim3(64 * BN_FW_MANY), 0, st, partials, chunks, C, rows, mean, invstd,
                           running_mean, running_var, momentum, eps);
    else
        ZSG_LAUNCH(bn_stats_finalize_kernel<BN_FW>, dim3(C / 4), dim3(64 * BN_FW), 0, st, partials, chunks, C, rows, mean, invstd,
                           running_mean, running_var, momentum, eps);
    ZSG_CHECK_LAUNCH("bn_stats_from_partials");
    return 0;
}

// ---- stem: BatchNorm + ReLU + max-pool in one pass (fpn_resnet.py / mdl.py:149-152: conv1 -> bn1 -> relu -> maxpool(3, 2, 1)) ------------
// The 150x150x64 stem activation is the largest tensor of the network (92 MB at B=16).  Separate launches read and write it three
// times in the forward (BatchNorm apply: x -> a; pool: a -> out) and six times in the backward (pool backward writes a dense d(a),
// BatchNorm backward reads it twice beside x).  Fused, the normalised activation never exists: the forward reads x once and writes
// the pooled map + window indices; the backward's statistics pass walks the POOLED gradient (the only non-zero entries of d(a) sit at
// the arg-max positions) and its apply pass gathers them per input pixel.  bn_pool_val is the one expression both directions use
// for relu(bn(x)), so the forward's arg-max / ReLU decisions and the backward's are the same bits.
__device__ __forceinline__ f32x4 bn_pool_val(const f32x4 x, const f32x4 mu, const f32x4 sc, const f32x4 be) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(x[e] - mu[e], sc[e], be[e]), 0.f);
    return v;
}

__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(const float* __restrict__ x, int B, int H, int W, int C4,
                                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta, int k, int s,
                                                                  int p, int Ho, int Wo, float* __restrict__ out, uint8_t* __restrict__ idx) {
    ZSG_SET_MAIN_PRIO();
    const int64_t total = (int64_t)B * Ho * Wo * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int c4, wo, ho, b;
        if (total < (1ll << 32)) {            // (32-bit divisions: a 64-bit one is ~4x the instructions, and there are three per element)
            const unsigned iu = (unsigned)i, t1 = iu / (unsigned)C4, t2 = t1 / (unsigned)Wo;
            c4 = (int)(iu - t1 * (unsigned)C4);
            wo = (int)(t1 - t2 * (unsigned)Wo);
            b = (int)(t2 / (unsigned)Ho);
            ho = (int)(t2 - (unsigned)b * (unsigned)Ho);
        } else {
            c4 = (int)(i % C4);
            int64_t t = i / C4;
            wo = (int)(t % Wo);
            t /= Wo;
            ho = (int)(t % Ho);
            b = (int)(t / Ho);
        }
        const f32x4 mu = *(const f32x4*)(mean + 4 * c4);
        const f32x4 sc = *(const f32x4*)(invstd + 4 * c4) * *(const f32x4*)(gamma + 4 * c4);
        const f32x4 be = *(const f32x4*)(beta + 4 * c4);
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
        for (int r = 0; r < k; ++r) {
            const int hi = ho * s - p + r;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int q = 0; q < k; ++q) {
                const int wi = wo * s - p + q;
                if ((unsigned)wi >= (unsigned)W) continue;
                const f32x4 v = bn_pool_val(*(const f32x4*)(x + (((int64_t)b * H + hi) * W + wi) * C4 * 4 + c4 * 4), mu, sc, be);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (v[e] > best[e] || v[e] != v[e]) {      // first maximum wins; NaN propagates (torch rule)
                        best[e] = v[e];
                        bi[e] = r * k + q;
                    }
            }
        }
        *(f32x4*)(out + i * 4) = best;
        *(uchar4*)(idx + i * 4) = make_uchar4((unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2], (unsigned char)bi[3]);
    }
}

// partial (sum g, sum g * xhat) over a chunk of POOLED pixels: g = dout * (relu(bn(x)) > 0) at the window's arg-max position
__global__ __launch_bounds__(256) void bn_pool_bwd_partial_kernel(const float* __restrict__ dout, const uint8_t* __restrict__ idx,
                                                                  const float* __restrict__ x, int H, int W, int C, int k, int s, int p, int Ho,
                                                                  int Wo, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta, int64_t rows,
                                                                  int lanes, int rpb, float* __restrict__ part) {
    ZSG_SET_MAIN_PRIO();
    __shared__ f32x4 red[2][256];
    const int rowlanes = 256 / lanes;
    const int l = threadIdx.x % lanes, rl = threadIdx.x / lanes;
    const int c = (blockIdx.y * lanes + l) * 4;
    const bool cok = c < C;
    const int64_t r_begin = (int64_t)blockIdx.x * rpb;
    const int64_t r_end = min(rows, r_begin + (int64_t)rpb);
    f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
    if (cok) {
        const f32x4 mu = *(const f32x4*)(mean + c), is = *(const f32x4*)(invstd + c);
        const f32x4 sc = is * *(const f32x4*)(gamma + c), be = *(const f32x4*)(beta + c);
        // pooled pixel cursor (b, ho, wo) advanced by rowlanes per trip (one 64-bit division per thread instead of two per trip), window
        // code / k by a 16-bit reciprocal (exact for code < 256, k <= 15)
        int64_t r = r_begin + rl;
        int wo = (int)(r % Wo);
        int64_t t0 = r / Wo;
        int ho = (int)(t0 % Ho);
        int64_t b = t0 / Ho;
        const unsigned kmag = 65536u / (unsigned)k + 1u;
        for (; r < r_end; r += rowlanes) {
            const uchar4 u = *(const uchar4*)(idx + r * C + c);
            const f32x4 g = *(const f32x4*)(dout + r * C + c);
            const unsigned code[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int cr = (int)((code[e] * kmag) >> 16), cq = (int)code[e] - cr * k;
                const int hi = ho * s - p + cr, wi = wo * s - p + cq;
                const float xv = x[((b * H + hi) * W + wi) * C + c + e];
                const float v = fmaxf(fmaf(xv - mu[e], sc[e], be[e]), 0.f);
                const float ge = v > 0.f ? g[e] : 0.f;
                s0[e] += ge;
                s1[e] += ge * ((xv - mu[e]) * is[e]);
            }
            wo += rowlanes;
            while (wo >= Wo) {
                wo -= Wo;
                if (++ho == Ho) {
                    ho = 0;
                    ++b;
                }
            }
        }
    }
    red[0][threadIdx.x] = s0;
    red[1][threadIdx.x] = s1;
    __syncthreads();
    if (rl == 0 && cok) {
        for (int kk = 1; kk < rowlanes; ++kk) {
            s0 += red[0][kk * lanes + l];
            s1 += red[1][kk * lanes + l];
        }
        float* o = part + (size_t)blockIdx.x * 2 * C;
        *(f32x4*)(o + c) = s0;
        *(f32x4*)(o + C + c) = s1;
    }
}

// dx = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)) per INPUT pixel; g gathered from the (<= ceil(k/s)^2) windows that contain it
__global__ __launch_bounds__(256) void bn_pool_bwd_apply_kernel(const float* __restrict__ dout, const uint8_t* __restrict__ idx,
                                                                const float* __restrict__ x, int H, int W, int C, int k, int s, int p, int Ho,
                                                                int Wo, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ coef, int64_t rows, float* __restrict__ dx, int lanes,
                                                                int rpb) {
    ZSG_SET_MAIN_PRIO();
    const int rowlanes = 256 / lanes;
    const int l = threadIdx.x % lanes, rl = threadIdx.x / lanes;
    const int c = (blockIdx.y * lanes + l) * 4;
    if (c >= C) return;
    const f32x4 mu = *(const f32x4*)(mean + c), is = *(const f32x4*)(invstd + c);
    const f32x4 sc = is * *(const f32x4*)(gamma + c), be = *(const f32x4*)(beta + c);
    const f32x4 c1 = *(const f32x4*)(coef + c), c2 = *(const f32x4*)(coef + C + c);
    const int64_t r_begin = (int64_t)blockIdx.x * rpb;
    const int64_t r_end = min(rows, r_begin + (int64_t)rpb);
    // input pixel cursor (b, hi, wi) advanced by rowlanes per trip; the windows that contain a pixel are ho in [ceil((hi + p - k + 1) / s),
    // floor((hi + p) / s)] (two divisions per pixel — shifts for the stem's stride 2 — instead of two per window tap), visited in the
    // order of the tap loop this replaces (tap row ascending = ho descending): the same sum, bit for bit
    int64_t r = r_begin + rl;
    int wi = (int)(r % W);
    int64_t t0 = r / W;
    int hi = (int)(t0 % H);
    int64_t b = t0 / H;
    const bool s2 = s == 2;
    for (; r < r_end; r += rowlanes) {
        const f32x4 xv = *(const f32x4*)(x + r * C + c);
        const f32x4 v = bn_pool_val(xv, mu, sc, be);
        f32x4 g = {0, 0, 0, 0};
        const int hn0 = hi + p, wn0 = wi + p;
        const int ho_hi = min(s2 ? (hn0 >> 1) : hn0 / s, Ho - 1), wo_hi = min(s2 ? (wn0 >> 1) : wn0 / s, Wo - 1);
        const int hlo = hn0 - k + s, wlo = wn0 - k + s;          // ceil((n - k + 1) / s) = floor((n - k + s) / s) for n - k + 1 > 0
        const int ho_lo = (hn0 - k + 1 <= 0) ? 0 : (s2 ? (hlo >> 1) : hlo / s), wo_lo = (wn0 - k + 1 <= 0) ? 0 : (s2 ? (wlo >> 1) : wlo / s);
        for (int ho = ho_hi; ho >= ho_lo; --ho) {
            const int rr = hn0 - ho * s;
            for (int wo = wo_hi; wo >= wo_lo; --wo) {
                const int64_t o = ((b * Ho + ho) * Wo + wo) * C + c;
                const uchar4 u = *(const uchar4*)(idx + o);
                const f32x4 d = *(const f32x4*)(dout + o);
                const unsigned code = rr * k + (wn0 - wo * s);
                g[0] += (u.x == code) ? d[0] : 0.f;
                g[1] += (u.y == code) ? d[1] : 0.f;
                g[2] += (u.z == code) ? d[2] : 0.f;
                g[3] += (u.w == code) ? d[3] : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = v[e] > 0.f ? g[e] : 0.f;
        const f32x4 xh = (xv - mu) * is;
        *(f32x4*)(dx + r * C + c) = sc * (g - c1 - xh * c2);
        wi += rowlanes;
        while (wi >= W) {
            wi -= W;
            if (++hi == H) {
                hi = 0;
                ++b;
            }
        }
    }
}

extern "C" int zsg_bn_relu_maxpool_fwd(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, const float* mean, const float* invstd,
                                       const float* gamma, const float* beta, int32_t k, int32_t s, int32_t p, int32_t Ho, int32_t Wo, float* out,
                                       uint8_t* idx, void* stream) {
    ZSG_REQUIRE(x && mean && invstd && gamma && beta && out && idx && B > 0 && C > 0 && (C % 4) == 0 && k > 0 && k <= 15 && s > 0,
                "bn_relu_maxpool_fwd: bad argument");
    const int64_t n = (int64_t)B * Ho * Wo * (C / 4);
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("bn_relu_maxpool_fwd", st, 0, ((double)B * H * W + (double)B * Ho * Wo * 1.25) * C * 4);
    int64_t blocks = (n + 255) / 256;
    if (blocks > ZSG_NUM_CU * 16) blocks = ZSG_NUM_CU * 16;
    ZSG_LAUNCH(bn_relu_maxpool_fwd_kernel, dim3((int)blocks), dim3(256), 0, st, x, B, H, W, C / 4, mean, invstd, gamma, beta, k, s, p, Ho, Wo,
                       out, idx);
    ZSG_CHECK_LAUNCH("bn_relu_maxpool_fwd");
    return 0;
}

// ws: >= zsg_bn_workspace_bytes(B * Ho * Wo, C)
extern "C" int zsg_bn_relu_maxpool_bwd(const float* dout, const uint8_t* idx, const float* x, int32_t B, int32_t H, int32_t W, int32_t C,
                                       const float* mean, const float* invstd, const float* gamma, const float* beta, int32_t k, int32_t s,
                                       int32_t p, int32_t Ho, int32_t Wo, float* dx, float* dgamma, float* dbeta, int32_t accumulate, void* ws,
                                       size_t ws_bytes, void* stream) {
    ZSG_REQUIRE(dout && idx && x && mean && invstd && gamma && beta && dx && ws && B > 0 && C > 0 && (C % 4) == 0 && k > 0 && k <= 15 && s > 0,
                "bn_relu_maxpool_bwd: bad argument");
    const int64_t prow = (int64_t)B * Ho * Wo, rows = (int64_t)B * H * W;
    if (ws_bytes < zsg_bn_workspace_bytes(prow, C)) ZSG_FAIL(-2, "bn_relu_maxpool_bwd: workspace too small");
    const BnGeom gp = bn_geom(prow, C), g = bn_geom(rows, C);
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("bn_backward", st, 0, ((double)prow * 2.25 + (double)rows * 2) * C * 4);
    float* part = (float*)ws;
    float* coef = part + (size_t)gp.chunks * 2 * C;
    ZSG_LAUNCH(bn_pool_bwd_partial_kernel, dim3(gp.chunks, gp.slabs), dim3(256), 0, st, dout, idx, x, H, W, C, k, s, p, Ho, Wo, mean, invstd,
                       gamma, beta, prow, gp.lanes, gp.rpb, part);
    ZSG_LAUNCH(bn_bwd_finalize_kernel, dim3(C / 4), dim3(64 * BN_FW), 0, st, part, gp.chunks, C, rows, coef, dgamma, dbeta,
                       accumulate);
    ZSG_LAUNCH(bn_pool_bwd_apply_kernel, dim3(g.chunks, g.slabs), dim3(256), 0, st, dout, idx, x, H, W, C, k, s, p, Ho, Wo, mean, invstd, gamma,
                       beta, coef, rows, dx, g.lanes, g.rpb);
    ZSG_CHECK_LAUNCH("bn_relu_maxpool_bwd");
    return 0;
}

// ---- eval mode: fold BatchNorm into the convolution that feeds it ------------------------------------------------------
// y = gamma*(conv(x,W) - mean)/sqrt(var+eps) + beta  ==  conv(x, W*s) + (beta - mean*s),  s = gamma/sqrt(var+eps) per
// output channel.  One launch rescales every folded weight row (OHWI: a row = one output channel) into an arena and
// writes the folded biases; the eval plan then runs conv(+bias, +residual, ReLU) with no BatchNorm launch at all.
struct ZsgFoldJob {
    int64_t w_off, dst_off, gamma_off, beta_off, bias_off;   // element offsets: flat parameters / arena
    int32_t row0, N, row_len, bn_index;                      // first global row, rows (= cout), floats per row, slot in rm / rv
};
__global__ __launch_bounds__(256) void bn_fold_kernel(const float* __restrict__ flat, const float* __restrict__ rmean,
                                                      const float* __restrict__ rvar, float eps, const ZsgFoldJob* __restrict__ jobs,
                                                      int njobs, float* __restrict__ arena) {
    int lo = 0, hi = njobs - 1;                      // last job whose row0 <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].row0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const ZsgFoldJob jb = jobs[lo];
    const int n = blockIdx.x - jb.row0;
    const float sc = flat[jb.gamma_off + n] / sqrtf(rvar[jb.bn_index + n] + eps);
    const float* src = flat + jb.w_off + (int64_t)n * jb.row_len;
    float* dst = arena + jb.dst_off + (int64_t)n * jb.row_len;
    for (int i = threadIdx.x * 4; i < jb.row_len; i += 1024) *(f32x4*)(dst + i) = *(const f32x4*)(src + i) * sc;
    if (threadIdx.x == 0) arena[jb.bias_off + n] = flat[jb.beta_off + n] - rmean[jb.bn_index + n] * sc;
}
extern "C" int zsg_bn_fold(const float* flat, const float* running_mean, const float* running_var, float eps, const void* jobs,
                           int32_t njobs, int32_t total_rows, float* arena, void* stream) {
    ZSG_REQUIRE(flat && running_mean && running_var && jobs && arena && njobs > 0 && total_rows > 0, "bn_fold: bad argument");
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("bn_fold", st, 0, 0);
    ZSG_LAUNCH(bn_fold_kernel, dim3(total_rows), dim3(256), 0, st, flat, running_mean, running_var, eps, (const ZsgFoldJob*)jobs, njobs,
                       arena);
    ZSG_CHECK_LAUNCH("bn_fold");
    return 0;
}

extern "C" int zsg_bn_eval_stats(const float* running_mean, const float* running_var, int32_t C, float eps, float* mean,
                                 float* invstd, void* stream) {
    ZSG_REQUIRE(running_mean && running_var && mean && invstd && C > 0, "bn_eval_stats: bad argument");
    ZSG_LAUNCH(bn_eval_stats_kernel, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, running_mean, running_var, C, eps,
                       mean, invstd);
    ZSG_CHECK_LAUNCH("bn_eval_stats");
    return 0;
}

extern "C" int zsg_bn_apply(const float* x, int64_t rows, int32_t C, const float* mean, const float* invstd, const float* gamma,
                            const float* beta, const float* residual, int32_t relu, float* out, uint8_t* relu_mask, void* stream) {
    ZSG_REQUIRE(x && mean && invstd && gamma && beta && out && rows > 0 && C > 0 && (C % 4) == 0, "bn_apply: bad argument");
    BnGeom g = bn_geom(rows, C);
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("bn_apply", st, 0, (double)rows * C * (4 * (residual ? 3 : 2) + (relu_mask && relu ? 0.25 : 0)));
    ZSG_LAUNCH(bn_apply_kernel, dim3(g.chunks, g.slabs), dim3(256), 0, st, x, rows, C, mean, invstd, gamma, beta, residual,
                       relu, out, relu_mask, g.lanes, g.rpb);
    ZSG_CHECK_LAUNCH("bn_apply");
    return 0;
}

// zsg_bn_backward without its first pass: the (sum g, sum g * xhat) partial rows [chunks][2][C] were written by the epilogue of
// the data-gradient convolution that completed dout (zsg_conv_igemm_bnb / zsg_conv_wino_bnb).  ws: >= 2 * C floats (coefficients).
extern "C" int zsg_bn_backward_from_partials(const float* dout, const uint8_t* relu_mask, const float* x, int64_t rows, int32_t C,
                                             const float* mean, const float* invstd, const float* gamma, float* dx, float* g_out,
                                             float* dgamma, float* dbeta, int32_t accumulate, const float* partials, int32_t chunks,
                                             void* ws, size_t ws_bytes, void* stream) {
    ZSG_REQUIRE(dout && x && mean && invstd && gamma && dx && ws && partials && chunks > 0 && rows > 0 && C > 0 && (C % 4) == 0,
                "bn_backward_from_partials: bad argument");
    if (ws_bytes < 2 * (size_t)C * sizeof(float)) ZSG_FAIL(-2, "bn_backward_from_partials: workspace too small");
    BnGeom g = bn_geom(rows, C);
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("bn_backward", st, 0, (double)rows * C * (4 * (2 + 1 + (g_out ? 1 : 0)) + (relu_mask ? 0.25 : 0)));
    float* coef = (float*)ws;
    ZSG_LAUNCH(bn_bwd_finalize_kernel, dim3(C / 4), dim3(64 * BN_FW), 0, st, partials, chunks, C, rows, coef, dgamma, dbeta,
                       accumulate);
    ZSG_LAUNCH(bn_bwd_apply_kernel, dim3(g.chunks, g.slabs), dim3(256), 0, st, dout, (const float*)nullptr, relu_mask, x, rows, C, mean,
                       invstd, gamma, coef, dx, g_out, g.lanes, g.rpb);
    ZSG_CHECK_LAUNCH("bn_backward_from_partials");
    return 0;
}

// The apply pass of the BatchNorm backward on its own: the coefficients (coef[0][c] = sum g / n, coef[1][c] = sum g xhat / n) and
// d(gamma) / d(beta) were finalised inside the data-gradient convolution that completed dout (zsg_conv_igemm_bnb_tail /
// zsg_conv_wino_bnb_tail).  dx = gamma * invstd * (g - coef0 - xhat * coef1), g = dout * relu-bit; g_out (optional) receives g (the
// residual branch's gradient).  Reference: native_batch_norm_backward's input-gradient formula.
extern "C" int zsg_bn_bwd_apply(const float* dout, const uint8_t* relu_mask, const float* x, int64_t rows, int32_t C, const float* mean,
                                const float* invstd, const float* gamma, const float* coef, float* dx, float* g_out, void* stream) {
    ZSG_REQUIRE(dout && x && mean && invstd && gamma && coef && dx && rows > 0 && C > 0 && (C % 4) == 0, "bn_bwd_apply: bad argument");
    BnGeom g = bn_geom(rows, C);
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("bn_backward", st, 0, (double)rows * C * (4 * (2 + 1 + (g_out ? 1 : 0)) + (relu_mask ? 0.25 : 0)));
    ZSG_LAUNCH(bn_bwd_apply_kernel, dim3(g.chunks, g.slabs), dim3(256), 0, st, dout, (const float*)nullptr, relu_mask, x, rows, C, mean,
                       invstd, gamma, coef, dx, g_out, g.lanes, g.rpb);
    ZSG_CHECK_LAUNCH("bn_bwd_apply");
    return 0;
}

extern "C" int zsg_bn_backward(const float* dout, const float* relu_out, const uint8_t* relu_mask, const float* x, int64_t rows, int32_t C, const float* mean,
                               const float* invstd, const float* gamma, float* dx, float* g_out, float* dgamma, float* dbeta,
                               int32_t accumulate, void* ws, size_t ws_bytes, void* stream) {
    ZSG_REQUIRE(dout && x && mean && invstd && gamma && dx && ws && rows > 0 && C > 0 && (C % 4) == 0, "bn_backward: bad argument");
    if (ws_bytes < zsg_bn_workspace_bytes(rows, C)) ZSG_FAIL(-2, "bn_backward: workspace too small");
    BnGeom g = bn_geom(rows, C);
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("bn_backward", st, 0, (double)rows * C * (4 * ((relu_out && !relu_mask ? 3 : 2) * 2 + 1 + (g_out ? 1 : 0)) + (relu_mask ? 0.5 : 0)));
    float* part = (float*)ws;
    float* coef = part + (size_t)g.chunks * 2 * C;
    ZSG_LAUNCH((bn_partial_kernel<1>), dim3(g.chunks, g.slabs), dim3(256), 0, st, x, dout, relu_out, relu_mask, mean, invstd,
                       rows, C, g.lanes, g.rpb, part);
    ZSG_LAUNCH(bn_bwd_finalize_kernel, dim3(C / 4), dim3(64 * BN_FW), 0, st, part, g.chunks, C, rows, coef, dgamma, dbeta,
                       accumulate);
    ZSG_LAUNCH(bn_bwd_apply_kernel, dim3(g.chunks, g.slabs), dim3(256), 0, st, dout, relu_out, relu_mask, x, rows, C, mean, invstd,
                       gamma, coef, dx, g_out, g.lanes, g.rpb);
    ZSG_CHECK_LAUNCH("bn_backward");
    return 0;
}
