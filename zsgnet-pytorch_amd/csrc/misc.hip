// misc.hip — HBM-bound NHWC kernels around the convolutions: pooling, nearest-upsample+add, ReLU, average pool,
// channel L2-norm, image layout conversion, language/grid fusion, weight transposes, column sums.
// All of them move 16 bytes per lane with channel-contiguous (coalesced) accesses.
#include "common.h"

ZSG_DEFINE_PRIO_FLAG()

static inline int grid_for(int64_t n_items, int block = 256, int cap = ZSG_NUM_CU * 16) {
    int64_t b = (n_items + block - 1) / block;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}

// i -> (c4, x, y, b) of a [B][Y][X][C4] index space: 32-bit divisions when the space allows (a 64-bit division is ~4x the instructions,
// and these element-wise kernels are otherwise a load and a store)
__device__ __forceinline__ void decomp4(int64_t i, int64_t total, int C4, int X, int Y, int& c4, int& x, int& y, int& b) {
    if (total < (1ll << 32)) {
        const unsigned iu = (unsigned)i, t1 = iu / (unsigned)C4, t2 = t1 / (unsigned)X;
        c4 = (int)(iu - t1 * (unsigned)C4);
        x = (int)(t1 - t2 * (unsigned)X);
        b = (int)(t2 / (unsigned)Y);
        y = (int)(t2 - (unsigned)b * (unsigned)Y);
    } else {
        c4 = (int)(i % C4);
        int64_t t = i / C4;
        x = (int)(t % X);
        t /= X;
        y = (int)(t % Y);
        b = (int)(t / Y);
    }
}
// windows of a k / stride s / pad p pooling that contain input coordinate n: o in [lo, hi] (empty when lo > hi); tap index = n + p - o*s
__device__ __forceinline__ void pool_windows(int n, int k, int s, int p, int On, int& lo, int& hi) {
    const int np = n + p;
    hi = min(s == 2 ? (np >> 1) : (s == 1 ? np : np / s), On - 1);
    const int num = np - k + 1;
    lo = num <= 0 ? 0 : (s == 2 ? ((num + 1) >> 1) : (s == 1 ? num : (num + s - 1) / s));
}

// ---- max pool -------------------------------------------------------------------------------------------------
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, int B, int H, int W, int C4, int k, int s, int p, int Ho, int Wo,
                                   float* __restrict__ out, uint8_t* __restrict__ idx) {
    ZSG_SET_MAIN_PRIO();
    const int64_t total = (int64_t)B * Ho * Wo * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int c4, wo, ho, b;
        decomp4(i, total, C4, Wo, Ho, c4, wo, ho, b);
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
        for (int r = 0; r < k; ++r) {
            const int hi = ho * s - p + r;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int q = 0; q < k; ++q) {
                const int wi = wo * s - p + q;
                if ((unsigned)wi >= (unsigned)W) continue;
                const f32x4 v = *(const f32x4*)(x + (((int64_t)b * H + hi) * W + wi) * C4 * 4 + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (v[e] > best[e] || v[e] != v[e]) {      // first maximum wins; NaN propagates (torch rule)
                        best[e] = v[e];
                        bi[e] = r * k + q;
                    }
            }
        }
        *(f32x4*)(out + i * 4) = best;
        if (idx) {
            uchar4 u = make_uchar4((unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2], (unsigned char)bi[3]);
            *(uchar4*)(idx + i * 4) = u;
        }
    }
}

__global__ void maxpool_bwd_kernel(const float* __restrict__ dout, const uint8_t* __restrict__ idx, int B, int H, int W, int C4, int k,
                                   int s, int p, int Ho, int Wo, float* __restrict__ dx) {
    ZSG_SET_MAIN_PRIO();
    const int64_t total = (int64_t)B * H * W * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int c4, wi, hi, b;
        decomp4(i, total, C4, W, H, c4, wi, hi, b);
        f32x4 acc = {0, 0, 0, 0};
        // windows (ho, r) with ho*s - p + r == hi, in the order of the tap loop r = 0 .. k-1 (ho descending): the same sum, bit for bit
        int ho_lo, ho_hi, wo_lo, wo_hi;
        pool_windows(hi, k, s, p, Ho, ho_lo, ho_hi);
        pool_windows(wi, k, s, p, Wo, wo_lo, wo_hi);
        for (int ho = ho_hi; ho >= ho_lo; --ho) {
            const int r = hi + p - ho * s;
            for (int wo = wo_hi; wo >= wo_lo; --wo) {
                const int64_t o = ((((int64_t)b * Ho + ho) * Wo + wo) * C4 + c4) * 4;
                const uchar4 u = *(const uchar4*)(idx + o);
                const f32x4 g = *(const f32x4*)(dout + o);
                const int code = r * k + (wi + p - wo * s);
                acc[0] += (u.x == code) ? g[0] : 0.f;
                acc[1] += (u.y == code) ? g[1] : 0.f;
                acc[2] += (u.z == code) ? g[2] : 0.f;
                acc[3] += (u.w == code) ? g[3] : 0.f;
            }
        }
        *(f32x4*)(dx + i * 4) = acc;
    }
}

extern "C" int zsg_maxpool_fwd(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, int32_t s, int32_t p, int32_t Ho,
                               int32_t Wo, float* out, uint8_t* idx, void* stream) {
    ZSG_REQUIRE(x && out && (C % 4) == 0 && k > 0 && k <= 15 && s > 0, "maxpool_fwd: bad argument");
    const int64_t n = (int64_t)B * Ho * Wo * (C / 4);
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("maxpool_fwd", st, 0, ((double)B * H * W + (double)B * Ho * Wo * 1.25) * C * 4);
    ZSG_LAUNCH(maxpool_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, B, H, W, C / 4, k, s, p, Ho, Wo, out, idx);
    ZSG_CHECK_LAUNCH("maxpool_fwd");
    return 0;
}

extern "C" int zsg_maxpool_bwd(const float* dout, const uint8_t* idx, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, int32_t s,
                               int32_t p, int32_t Ho, int32_t Wo, float* dx, void* stream) {
    ZSG_REQUIRE(dout && idx && dx && (C % 4) == 0 && k > 0 && s > 0, "maxpool_bwd: bad argument");
    const int64_t n = (int64_t)B * H * W * (C / 4);
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("maxpool_bwd", st, 0, ((double)B * H * W + (double)B * Ho * Wo * 1.25) * C * 4);
    ZSG_LAUNCH(maxpool_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, st, dout, idx, B, H, W, C / 4, k, s, p, Ho, Wo, dx);
    ZSG_CHECK_LAUNCH("maxpool_bwd");
    return 0;
}

// ---- nearest upsample + add (F.interpolate(size=...), legacy 'nearest': src = floor(dst * in/out)) -------------
__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
    const int s = (int)floorf((float)dst * scale);
    return s < in_size - 1 ? s : in_size - 1;
}

__global__ void upsample_add_fwd_kernel(const float* __restrict__ a, const float* __restrict__ p, int B, int Hs, int Ws, int Hd, int Wd,
                                        int C4, float sh, float sw, float* __restrict__ out) {
    ZSG_SET_MAIN_PRIO();
    const int64_t total = (int64_t)B * Hd * Wd * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        int64_t t = i / C4;
        const int x = (int)(t % Wd);
        t /= Wd;
        const int y = (int)(t % Hd);
        const int b = (int)(t / Hd);
        const int ys = nearest_src(y, sh, Hs), xs = nearest_src(x, sw, Ws);
        const f32x4 u = *(const f32x4*)(p + ((((int64_t)b * Hs + ys) * Ws + xs) * C4 + c4) * 4);
        *(f32x4*)(out + i * 4) = *(const f32x4*)(a + i * 4) + u;
    }
}

__global__ void upsample_add_bwd_kernel(const float* __restrict__ dout, int B, int Hs, int Ws, int Hd, int Wd, int C4, float sh, float sw,
                                        float* __restrict__ dp, int accumulate) {
    ZSG_SET_MAIN_PRIO();
    const int64_t total = (int64_t)B * Hs * Ws * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        int64_t t = i / C4;
        const int xs = (int)(t % Ws);
        t /= Ws;
        const int ys = (int)(t % Hs);
        const int b = (int)(t / Hs);
        f32x4 acc = {0, 0, 0, 0};
        // destination pixels whose nearest source is (ys, xs): scan the small candidate range
        const int y_lo = max(0, (int)floorf((float)ys / sh) - 1), y_hi = min(Hd - 1, (int)ceilf((float)(ys + 1) / sh) + 1);
        const int x_lo = max(0, (int)floorf((float)xs / sw) - 1), x_hi = min(Wd - 1, (int)ceilf((float)(xs + 1) / sw) + 1);
        for (int y = y_lo; y <= y_hi; ++y) {
            if (nearest_src(y, sh, Hs) != ys) continue;
            for (int x = x_lo; x <= x_hi; ++x) {
                if (nearest_src(x, sw, Ws) != xs) continue;
                acc += *(const f32x4*)(dout + ((((int64_t)b * Hd + y) * Wd + x) * C4 + c4) * 4);
            }
        }
        if (accumulate) acc += *(const f32x4*)(dp + i * 4);
        *(f32x4*)(dp + i * 4) = acc;
    }
}

extern "C" int zsg_upsample_add_fwd(const float* a, const float* p, int32_t B, int32_t Hs, int32_t Ws, int32_t Hd, int32_t Wd, int32_t C,
                                    float* out, void* stream) {
    ZSG_REQUIRE(a && p && out && (C % 4) == 0, "upsample_add_fwd: bad argument");
    const int64_t n = (int64_t)B * Hd * Wd * (C / 4);
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("upsample_add_fwd", st, 0, (double)n * 16 * 2.25);
    ZSG_LAUNCH(upsample_add_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, st, a, p, B, Hs, Ws, Hd, Wd, C / 4,
                       (float)Hs / (float)Hd, (float)Ws / (float)Wd, out);
    ZSG_CHECK_LAUNCH("upsample_add_fwd");
    return 0;
}

extern "C" int zsg_upsample_add_bwd(const float* dout, int32_t B, int32_t Hs, int32_t Ws, int32_t Hd, int32_t Wd, int32_t C, float* dp,
                                    int32_t accumulate, void* stream) {
    ZSG_REQUIRE(dout && dp && (C % 4) == 0, "upsample_add_bwd: bad argument");
    const int64_t n = (int64_t)B * Hs * Ws * (C / 4);
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("upsample_add_bwd", st, 0, (double)B * Hd * Wd * C * 4 * 1.25);
    ZSG_LAUNCH(upsample_add_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, st, dout, B, Hs, Ws, Hd, Wd, C / 4,
                       (float)Hs / (float)Hd, (float)Ws / (float)Wd, dp, accumulate);
    ZSG_CHECK_LAUNCH("upsample_add_bwd");
    return 0;
}

// ---- relu ------------------------------------------------------------------------------------------------------
__global__ void relu_fwd_kernel(const float* __restrict__ x, int64_t n4, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 v = *(const f32x4*)(x + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        *(f32x4*)(out + i * 4) = v;
    }
}
__global__ void relu_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ x, int64_t n4, float* __restrict__ dx, int accumulate) {
    ZSG_SET_MAIN_PRIO();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = *(const f32x4*)(x + i * 4);
        f32x4 g = *(const f32x4*)(dout + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = v[e] > 0.f ? g[e] : 0.f;
        if (accumulate) g += *(const f32x4*)(dx + i * 4);
        *(f32x4*)(dx + i * 4) = g;
    }
}
extern "C" int zsg_relu_fwd(const float* x, int64_t n, float* out, void* stream) {
    ZSG_REQUIRE(x && out && (n % 4) == 0, "relu_fwd: bad argument");
    ZSG_LAUNCH(relu_fwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, x, n / 4, out);
    ZSG_CHECK_LAUNCH("relu_fwd");
    return 0;
}
extern "C" int zsg_relu_bwd(const float* dout, const float* x, int64_t n, float* dx, int32_t accumulate, void* stream) {
    ZSG_REQUIRE(dout && x && dx && (n % 4) == 0, "relu_bwd: bad argument");
    ZSG_LAUNCH(relu_bwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, dout, x, n / 4, dx, accumulate);
    ZSG_CHECK_LAUNCH("relu_bwd");
    return 0;
}

// ---- adaptive average pool to 1x1 --------------------------------------------------------------------------------
__global__ void avgpool_fwd_kernel(const float* __restrict__ x, int B, int HW, int C, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i % C;
    float s = 0.f;
    for (int k = 0; k < HW; ++k) s += x[((int64_t)b * HW + k) * C + c];
    out[i] = s / (float)HW;
}
__global__ void avgpool_bwd_kernel(const float* __restrict__ dout, int B, int HW, int C, float* __restrict__ dx, int accumulate) {
    ZSG_SET_MAIN_PRIO();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * HW * C) return;
    const int c = i % C, b = i / (HW * C);
    const float g = dout[b * C + c] / (float)HW;
    dx[i] = (accumulate ? dx[i] : 0.f) + g;
}
extern "C" int zsg_avgpool_fwd(const float* x, int32_t B, int32_t HW, int32_t C, float* out, void* stream) {
    ZSG_REQUIRE(x && out && B > 0 && HW > 0 && C > 0, "avgpool_fwd: bad argument");
    ZSG_LAUNCH(avgpool_fwd_kernel, dim3(cdiv((int64_t)B * C, 256)), dim3(256), 0, (hipStream_t)stream, x, B, HW, C, out);
    ZSG_CHECK_LAUNCH("avgpool_fwd");
    return 0;
}
extern "C" int zsg_avgpool_bwd(const float* dout, int32_t B, int32_t HW, int32_t C, float* dx, int32_t accumulate, void* stream) {
    ZSG_REQUIRE(dout && dx && B > 0 && HW > 0 && C > 0, "avgpool_bwd: bad argument");
    ZSG_LAUNCH(avgpool_bwd_kernel, dim3(cdiv((int64_t)B * HW * C, 256)), dim3(256), 0, (hipStream_t)stream, dout, B, HW, C, dx,
                       accumulate);
    ZSG_CHECK_LAUNCH("avgpool_bwd");
    return 0;
}

// ---- channel L2 norm (one wave per pixel row) ----------------------------------------------------------------------
__global__ void l2norm_fwd_kernel(const float* __restrict__ x, int64_t rows, int C, float* __restrict__ out, float* __restrict__ norm) {
    ZSG_SET_MAIN_PRIO();
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float v = x[row * C + c];
        s += v * v;
    }
    s = wave_sum(s);
    const float nrm = sqrtf(s);
    if (lane == 0) norm[row] = nrm;
    for (int c = lane; c < C; c += 64) out[row * C + c] = x[row * C + c] / nrm;
}
__global__ void l2norm_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out, const float* __restrict__ norm,
                                  int64_t rows, int C, float* __restrict__ dx) {
    ZSG_SET_MAIN_PRIO();
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += dout[row * C + c] * out[row * C + c];
    s = wave_sum(s);
    const float inv = 1.0f / norm[row];
    for (int c = lane; c < C; c += 64) dx[row * C + c] = (dout[row * C + c] - out[row * C + c] * s) * inv;
}
extern "C" int zsg_l2norm_fwd(const float* x, int64_t rows, int32_t C, float* out, float* norm, void* stream) {
    ZSG_REQUIRE(x && out && norm && rows > 0 && C > 0, "l2norm_fwd: bad argument");
    ZSG_LAUNCH(l2norm_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, C, out, norm);
    ZSG_CHECK_LAUNCH("l2norm_fwd");
    return 0;
}
extern "C" int zsg_l2norm_bwd(const float* dout, const float* out, const float* norm, int64_t rows, int32_t C, float* dx, void* stream) {
    ZSG_REQUIRE(dout && out && norm && dx && rows > 0 && C > 0, "l2norm_bwd: bad argument");
    ZSG_LAUNCH(l2norm_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, dout, out, norm, rows, C, dx);
    ZSG_CHECK_LAUNCH("l2norm_bwd");
    return 0;
}

// ---- image NCHW -> NHWC4 ---------------------------------------------------------------------------------------------
__global__ void nchw_to_nhwc4_kernel(const float* __restrict__ img, int B, int C, int HW, float* __restrict__ out) {
    ZSG_SET_MAIN_PRIO();
    const int64_t total = (int64_t)B * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t b, px;
        if (total < (1ll << 32)) {            // (one 32-bit division instead of two 64-bit ones per pixel)
            const unsigned bu = (unsigned)i / (unsigned)HW;
            b = bu;
            px = (unsigned)i - bu * (unsigned)HW;
        } else {
            b = i / HW;
            px = i % HW;
        }
        f32x4 v = {0, 0, 0, 0};
        for (int c = 0; c < C && c < 4; ++c) v[c] = img[(b * C + c) * HW + px];
        *(f32x4*)(out + i * 4) = v;
    }
}
extern "C" int zsg_nchw_to_nhwc4(const float* img, int32_t B, int32_t C, int32_t H, int32_t W, float* out, void* stream) {
    ZSG_REQUIRE(img && out && C >= 1 && C <= 4, "nchw_to_nhwc4: bad argument (C=%d)", C);
    const int64_t n = (int64_t)B * H * W;
    ZSG_LAUNCH(nchw_to_nhwc4_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, img, B, C, H * W, out);
    ZSG_CHECK_LAUNCH("nchw_to_nhwc4");
    return 0;
}

// uint8 HWC image (as PIL decodes it) -> float NHWC4 in [0,1]: `pil2tensor(img).float().div_(255)` (dat_loader.py:26-33,
// 134) plus the stem's layout in one pass; IEEE division, so the values equal the host conversion bit for bit.
__global__ void u8hwc_to_nhwc4_kernel(const uint8_t* __restrict__ img, int64_t pixels, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pixels; i += (int64_t)gridDim.x * blockDim.x) {
        const uint8_t* p = img + i * 3;
        f32x4 v = {(float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, 0.f};
        *(f32x4*)(out + i * 4) = v;
    }
}
extern "C" int zsg_u8hwc_to_nhwc4(const uint8_t* img, int64_t pixels, float* out, void* stream) {
    ZSG_REQUIRE(img && out && pixels > 0, "u8hwc_to_nhwc4: bad argument");
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("u8hwc_to_nhwc4", st, 0, (double)pixels * 19);
    ZSG_LAUNCH(u8hwc_to_nhwc4_kernel, dim3(grid_for(pixels)), dim3(256), 0, st, img, pixels, out);
    ZSG_CHECK_LAUNCH("u8hwc_to_nhwc4");
    return 0;
}

// ---- PIL.Image.resize on the GPU (uint8, bit-identical) ------------------------------------------------------------------
// `img.resize((W, H))` of the reference loader (dat_loader.py:121; Pillow's default filter: bicubic) as Pillow's ImagingResample runs
// it for 8-bit channels: a horizontal pass, then a vertical pass, each a windowed sum of uint8 samples with 22-bit fixed-point
// weights, started at 1 << 21, shifted right by 22 and clipped to [0, 255].  The weights (bounds[o] = first tap | tap count,
// coef[o][0..ksize)) are Pillow's double-precision weights rounded on the HOST (dat_loader.resize_tables: they depend only on the two
// axis lengths); the integer arithmetic here is exact, so the output equals Pillow's byte for byte.  One thread per output
// (pixel, channel); byte work bound by launch latency, not by bandwidth (a 500 x 375 -> 300 x 300 image is 0.7 MB of traffic).
// axis 0: out[y][o][c] = f(in[y][x0 + t][c]);  axis 1: out[o][x][c] = f(in[y0 + t][x][c]).
template <int AXIS>
__global__ __launch_bounds__(256) void resize_pass_kernel(const uint8_t* __restrict__ in, int in_h, int in_w, int C, const int32_t* __restrict__ bounds,
                                                          const int32_t* __restrict__ coef, int ksize, int out_h, int out_w, uint8_t* __restrict__ out) {
    const int64_t total = (int64_t)out_h * out_w * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int x = (int)((i / C) % out_w);
        const int y = (int)(i / ((int64_t)C * out_w));
        const int o = AXIS == 0 ? x : y;
        const int first = bounds[2 * o], n = bounds[2 * o + 1];
        const int32_t* k = coef + (int64_t)o * ksize;
        int acc = 1 << 21;
        if (AXIS == 0) {
            const uint8_t* p = in + ((int64_t)y * in_w + first) * C + c;
            for (int t = 0; t < n; ++t) acc += (int)p[(int64_t)t * C] * k[t];
        } else {
            const uint8_t* p = in + ((int64_t)first * in_w + x) * C + c;
            for (int t = 0; t < n; ++t) acc += (int)p[(int64_t)t * in_w * C] * k[t];
        }
        acc >>= 22;                                  // (arithmetic shift, as Pillow's clip8 lookup index)
        out[i] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
    }
}

// x_* / y_* = the tap tables of the horizontal / vertical pass (device memory), NULL when that axis keeps its length (Pillow skips
// the pass); tmp: in_h x out_w x C bytes of scratch for the intermediate image.
extern "C" int zsg_resize_u8(const uint8_t* src, int32_t H, int32_t W, int32_t C, const int32_t* x_bounds, const int32_t* x_coef, int32_t x_ksize,
                             const int32_t* y_bounds, const int32_t* y_coef, int32_t y_ksize, int32_t Ho, int32_t Wo, uint8_t* tmp, uint8_t* out,
                             void* stream) {
    ZSG_REQUIRE(src && out && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0, "resize_u8: bad argument");
    ZSG_REQUIRE((x_bounds != nullptr) == (x_coef != nullptr) && (y_bounds != nullptr) == (y_coef != nullptr), "resize_u8: a pass needs bounds AND coefficients");
    ZSG_REQUIRE(x_bounds || W == Wo, "resize_u8: the width changes (%d -> %d) but no horizontal tap table was given", W, Wo);
    ZSG_REQUIRE(y_bounds || H == Ho, "resize_u8: the height changes (%d -> %d) but no vertical tap table was given", H, Ho);
    ZSG_REQUIRE(!(x_bounds && y_bounds) || tmp, "resize_u8: two passes need the scratch image");
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("resize_u8", st, 0, (double)H * W * C + 2.0 * H * Wo * C + (double)Ho * Wo * C);
    const uint8_t* cur = src;
    int cur_w = W;
    if (x_bounds) {
        uint8_t* dst = y_bounds ? tmp : out;
        ZSG_LAUNCH(resize_pass_kernel<0>, dim3(grid_for((int64_t)H * Wo * C)), dim3(256), 0, st, cur, H, W, C, x_bounds, x_coef, x_ksize, H, Wo, dst);
        cur = dst;
        cur_w = Wo;
    }
    if (y_bounds) {
        ZSG_LAUNCH(resize_pass_kernel<1>, dim3(grid_for((int64_t)Ho * Wo * C)), dim3(256), 0, st, cur, H, cur_w, C, y_bounds, y_coef, y_ksize, Ho, Wo, out);
    } else if (!x_bounds) {
        hipError_t e = hipMemcpyAsync(out, src, (size_t)H * W * C, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) ZSG_FAIL(-3, "resize_u8: copy: %s", hipGetErrorString(e));
    }
    ZSG_CHECK_LAUNCH("resize_u8");
    return 0;
}

// ---- batched form: every image of a batch in TWO launches (round 5) ------------------------------------------------------------
// The per-image form above costs the consumer thread one call (two launches) and one upload per image: 1 323 img/s from one worker, but
// only 3 680 img/s from sixteen (the host-side resize reached 6 367: profiles/r04_loader_rate.txt).  Here the whole batch is ONE flat
// uint8 upload and one job table; launch 1 runs every image's horizontal pass, launch 2 every vertical pass (a job with a pass that
// keeps its length carries the identity tap table: n = 1, coefficient 2^22 — the value passes through the fixed point exactly).
// Same arithmetic as resize_pass_kernel (Pillow's two-pass fixed-point bicubic, dat_loader.py:121): byte-identical.
struct ZsgResizeJob {
    int64_t src, tmp, out;                 // absolute device addresses: raw image [h][w][C], scratch [h][Wo][C], result [Ho][Wo][C]
    int64_t xb, xc, yb, yc;                // tap tables (int32): bounds [n_out][2], coefficients [n_out][ksize]
    int32_t h, w, xk, yk, blk0_x, blk0_y, pad0, pad1;      // blk0_*: first block of the job in launch 1 / 2 (1024 outputs per block)
};
template <int AXIS>
__global__ __launch_bounds__(256) void resize_batched_kernel(const ZsgResizeJob* __restrict__ jobs, int njobs, int C, int Ho, int Wo) {
    int lo = 0, hi = njobs - 1;                      // last job whose first block <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((AXIS == 0 ? jobs[mid].blk0_x : jobs[mid].blk0_y) <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const ZsgResizeJob jb = jobs[lo];
    const int lb = (int)blockIdx.x - (AXIS == 0 ? jb.blk0_x : jb.blk0_y);
    const uint8_t* in = (const uint8_t*)(AXIS == 0 ? jb.src : jb.tmp);
    uint8_t* out = (uint8_t*)(AXIS == 0 ? jb.tmp : jb.out);
    const int32_t* bounds = (const int32_t*)(AXIS == 0 ? jb.xb : jb.yb);
    const int32_t* coef = (const int32_t*)(AXIS == 0 ? jb.xc : jb.yc);
    const int ksize = AXIS == 0 ? jb.xk : jb.yk;
    const int in_w = AXIS == 0 ? jb.w : Wo, out_h = AXIS == 0 ? jb.h : Ho;
    const int total = out_h * Wo * C;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = lb * 1024 + u * 256 + (int)threadIdx.x;
        if (i >= total) break;
        const int c = i % C;
        const int x = (i / C) % Wo;
        const int y = i / (C * Wo);
        const int o = AXIS == 0 ? x : y;
        const int first = bounds[2 * o], n = bounds[2 * o + 1];
        const int32_t* k = coef + (int64_t)o * ksize;
        int acc = 1 << 21;
        if (AXIS == 0) {
            const uint8_t* p = in + ((int64_t)y * in_w + first) * C + c;
            for (int t = 0; t < n; ++t) acc += (int)p[(int64_t)t * C] * k[t];
        } else {
            const uint8_t* p = in + ((int64_t)first * in_w + x) * C + c;
            for (int t = 0; t < n; ++t) acc += (int)p[(int64_t)t * in_w * C] * k[t];
        }
        acc >>= 22;
        out[i] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
    }
}
extern "C" int zsg_resize_u8_batched(const void* jobs_dev, int32_t njobs, int32_t C, int32_t Ho, int32_t Wo, int32_t blocks_x, int32_t blocks_y,
                                     void* stream) {
    ZSG_REQUIRE(jobs_dev && njobs > 0 && C > 0 && Ho > 0 && Wo > 0 && blocks_x > 0 && blocks_y > 0, "resize_u8_batched: bad argument");
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("resize_u8", st, 0, 0);
    ZSG_LAUNCH(resize_batched_kernel<0>, dim3(blocks_x), dim3(256), 0, st, (const ZsgResizeJob*)jobs_dev, njobs, C, Ho, Wo);
    ZSG_LAUNCH(resize_batched_kernel<1>, dim3(blocks_y), dim3(256), 0, st, (const ZsgResizeJob*)jobs_dev, njobs, C, Ho, Wo);
    ZSG_CHECK_LAUNCH("resize_u8_batched");
    return 0;
}

// ---- weight transpose [N][T][C] -> [C][T][dst_ld >= N] (pad columns zeroed) ------------------------------------------
__global__ void transpose_w_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int T, int C, int dst_ld) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int n0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int n = n0 + j, c = c0 + tx;
        tile[j][tx] = (n < N && c < C) ? src[((int64_t)n * T + t) * C + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, n = n0 + tx;
        if (c < C && n < dst_ld) dst[((int64_t)c * T + t) * dst_ld + n] = tile[tx][j];
    }
}
extern "C" int zsg_transpose_w(const float* src, float* dst, int32_t N, int32_t T, int32_t C, int32_t dst_ld, void* stream) {
    ZSG_REQUIRE(src && dst && N > 0 && T > 0 && C > 0 && dst_ld >= N, "transpose_w: bad argument");
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("transpose_w", st, 0, (double)N * T * C * 8);
    ZSG_LAUNCH(transpose_w_kernel, dim3(cdiv(C, 32), cdiv(dst_ld, 32), T), dim3(256), 0, st, src, dst, N, T, C, dst_ld);
    ZSG_CHECK_LAUNCH("transpose_w");
    return 0;
}

// Batched form: ONE launch refreshes every dgrad weight image of the step.  jobs[j] = {src_off, dst_off, N, T, C,
// dst_ld, tile0, tiles_c} (element offsets from src_base / dst_base; tile0 = first flat tile index of job j).
struct ZsgTransposeJob {
    int64_t src_off, dst_off;
    int32_t N, T, C, dst_ld, tile0, tiles_c, tiles_n, pad;
};
// 64 x 64 tiles moved with 16-byte accesses on both sides (round 5; the 32 x 32 scalar version ran at 1.3-1.8 TB/s: 159 us of
// side-stream HBM time under every training forward): a thread loads float4 runs of a source row [n][c .. c+3], the tile is transposed
// through LDS (row pitch 65 floats: the strided reads of the write phase are conflict-free), and a thread stores float4 runs of a
// destination row [c][n .. n+3].  C, dst_ld: multiples of 4 (padded channel counts); rows n >= N read as zeros.
#define TWB 64
__global__ __launch_bounds__(256) void transpose_w_batched_kernel(const float* __restrict__ src_base, float* __restrict__ dst_base,
                                                                  const ZsgTransposeJob* __restrict__ jobs, int njobs) {
    __shared__ float tile[TWB][TWB + 1];
    int lo = 0, hi = njobs - 1;                      // last job whose tile0 <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const ZsgTransposeJob jb = jobs[lo];
    int t = blockIdx.x - jb.tile0;
    const int tc = t % jb.tiles_c;
    t /= jb.tiles_c;
    const int tn = t % jb.tiles_n;
    const int tap = t / jb.tiles_n;
    const float* src = src_base + jb.src_off;
    float* dst = dst_base + jb.dst_off;
    const int n0 = tn * TWB, c0 = tc * TWB;
    const int g = threadIdx.x & 15, r = threadIdx.x >> 4;        // 16 column groups of 16 bytes x 16 rows per pass
    f32x4 v[4];
#pragma unroll
    for (int pss = 0; pss < 4; ++pss) {               // all four loads first (one memory round trip)
        const int n = n0 + r + 16 * pss, c = c0 + 4 * g;
        v[pss] = (n < jb.N && c < jb.C) ? *(const f32x4*)(src + ((int64_t)n * jb.T + tap) * jb.C + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int pss = 0; pss < 4; ++pss)
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[r + 16 * pss][4 * g + e] = v[pss][e];
    __syncthreads();
#pragma unroll
    for (int pss = 0; pss < 4; ++pss) {
        const int c = c0 + r + 16 * pss, n = n0 + 4 * g;
        if (c < jb.C && n < jb.dst_ld) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = tile[4 * g + e][r + 16 * pss];
            *(f32x4*)(dst + ((int64_t)c * jb.T + tap) * jb.dst_ld + n) = o;
        }
    }
}
extern "C" int zsg_transpose_w_batched(const float* src_base, float* dst_base, const void* jobs, int32_t njobs, int32_t total_tiles,
                                       void* stream) {
    ZSG_REQUIRE(src_base && dst_base && jobs && njobs > 0 && total_tiles > 0, "transpose_w_batched: bad argument");
    ZSG_REQUIRE((((uintptr_t)src_base | (uintptr_t)dst_base) & 15) == 0, "transpose_w_batched: bases not 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("transpose_w", st, 0, 0);
    ZSG_LAUNCH(transpose_w_batched_kernel, dim3(total_tiles), dim3(256), 0, st, src_base, dst_base, (const ZsgTransposeJob*)jobs, njobs);
    ZSG_CHECK_LAUNCH("transpose_w_batched");
    return 0;
}

// ---- column sums ---------------------------------------------------------------------------------------------------------
// grid (col blocks of 64, row splits, groups); block 256 = 64 columns x 4 row lanes; atomics merge the row splits.
__global__ void colsum_kernel(const float* __restrict__ x, int64_t gstride, int rows, int ld, int c0, int C, float* __restrict__ out,
                              int rows_per_block) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int g = blockIdx.z;
    const int r_begin = blockIdx.y * rows_per_block;
    const int r_end = min(rows, r_begin + rows_per_block);
    float s = 0.f;
    if (col < C)
        for (int r = r_begin + rl; r < r_end; r += 4) s += x[g * gstride + (int64_t)r * ld + c0 + col];
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && col < C) {
        s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        unsafeAtomicAdd(out + (int64_t)g * C + col, s);
    }
}
// 16-byte variant: a thread owns 4 consecutive columns; CL column lanes x (256/CL) row lanes; four independent loads
// in flight per thread (the scalar kernel above keeps one: measured 0.3 TB/s).
__global__ void colsum4_kernel(const float* __restrict__ x, int64_t gstride, int rows, int ld, int c0, int C, float* __restrict__ out,
                               int rows_per_block, int CL) {
    __shared__ f32x4 red[256];
    const int RL = 256 / CL;
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    const int col = (blockIdx.x * CL + cl) * 4;
    const int g = blockIdx.z;
    const int r_begin = blockIdx.y * rows_per_block;
    const int r_end = min(rows, r_begin + rows_per_block);
    f32x4 s0 = {0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;
    if (col < C) {
        const float* base = x + g * gstride + c0 + col;
        int r = r_begin + rl;
        for (; r + 3 * RL < r_end; r += 4 * RL) {
            s0 += *(const f32x4*)(base + (int64_t)r * ld);
            s1 += *(const f32x4*)(base + (int64_t)(r + RL) * ld);
            s2 += *(const f32x4*)(base + (int64_t)(r + 2 * RL) * ld);
            s3 += *(const f32x4*)(base + (int64_t)(r + 3 * RL) * ld);
        }
        for (; r < r_end; r += RL) s0 += *(const f32x4*)(base + (int64_t)r * ld);
    }
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rl == 0 && col < C) {
        f32x4 s = red[cl];
        for (int k = 1; k < RL; ++k) s += red[k * CL + cl];
        float* o = out + (int64_t)g * C + col;
#pragma unroll
        for (int e = 0; e < 4; ++e) unsafeAtomicAdd(o + e, s[e]);
    }
}
__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ p, int64_t n, float v) {
    // 16-byte stores over the aligned body, scalar head / tail
    const int64_t head = min(n, (int64_t)((16 - ((uintptr_t)p & 15)) & 15) / 4);
    const int64_t n4 = (n - head) / 4;
    f32x4* q = (f32x4*)(p + head);
    const f32x4 v4 = {v, v, v, v};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) q[i] = v4;
    if (blockIdx.x == 0) {
        for (int64_t i = threadIdx.x; i < head; i += blockDim.x) p[i] = v;
        for (int64_t i = head + n4 * 4 + threadIdx.x; i < n; i += blockDim.x) p[i] = v;
    }
}
extern "C" int zsg_memset_f32(float* p, int64_t n, float value, void* stream) {
    ZSG_REQUIRE(p || n == 0, "memset_f32: null");
    if (n <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    // (a kernel, not hipMemsetAsync: the runtime's fill brings a ~6.5 us marker gap into the stream and cannot carry a completion event)
    ZSG_LAUNCH(fill_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, st, p, n, value);
    ZSG_CHECK_LAUNCH("memset_f32");
    return 0;
}
extern "C" int zsg_colsum(const float* x, int32_t groups, int64_t gstride, int32_t rows, int32_t ld, int32_t c0, int32_t C, float* out,
                          int32_t accumulate, void* stream) {
    ZSG_REQUIRE(x && out && groups > 0 && rows > 0 && C > 0, "colsum: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(out, 0, (size_t)groups * C * 4, st);
        if (e != hipSuccess) ZSG_FAIL(-3, "colsum: %s", hipGetErrorString(e));
    }
    ZSG_PROF("colsum", st, 0, (double)groups * rows * C * 4);
    if ((C % 4) == 0 && (ld % 4) == 0 && (c0 % 4) == 0 && (gstride % 4) == 0 && (((uintptr_t)x) & 15) == 0) {
        const int c4 = C / 4;
        const int CL = c4 >= 16 ? 16 : (c4 >= 8 ? 8 : 4);     // 64 columns (256 B runs) x 16 row lanes per block
        const int cb4 = cdiv(c4, CL);
        int CLd = CL;
        int sp = cdiv(rows, 8 * (256 / CL));                  // >= 8 rows per thread ...
        if (sp > 64) sp = 64;                                 // ... and at most 64 atomic adds per output element
        if (g_zsg_deterministic) {                            // one block per column group: a single add per element, fixed order
            sp = 1;
            CLd = 4;                                          // (narrow column groups keep some parallelism: C/16 blocks per group)
        }
        const int rpb4 = cdiv(rows, sp);
        if (g_zsg_deterministic) {
            ZSG_LAUNCH(colsum4_kernel, dim3(cdiv(c4, CLd), 1, groups), dim3(256), 0, st, x, gstride, rows, ld, c0, C, out, rpb4, CLd);
            ZSG_CHECK_LAUNCH("colsum");
            return 0;
        }
        ZSG_LAUNCH(colsum4_kernel, dim3(cb4, cdiv(rows, rpb4), groups), dim3(256), 0, st, x, gstride, rows, ld, c0, C, out, rpb4, CL);
        ZSG_CHECK_LAUNCH("colsum");
        return 0;
    }
    int splits = cdiv(rows, 256);
    const int cb = cdiv(C, 64);
    while (splits > 1 && (int64_t)splits * cb * groups > 2048) splits = (splits + 1) / 2;
    if (g_zsg_deterministic) splits = 1;
    const int rpb = cdiv(rows, splits);
    ZSG_LAUNCH(colsum_kernel, dim3(cb, cdiv(rows, rpb), groups), dim3(256), 0, st, x, gstride, rows, ld, c0, C, out, rpb);
    ZSG_CHECK_LAUNCH("colsum");
    return 0;
}

// ---- dst[r][0:dst_ld] = [src[r][0:C] | 0]  (e.g. the 45-channel head gradient -> 48-channel GEMM operand) -------------
__global__ void pad_rows_kernel(const float* __restrict__ src, int64_t rows, int C, int src_ld, float* __restrict__ dst, int dst_ld) {
    ZSG_SET_MAIN_PRIO();
    const int64_t total = rows * dst_ld;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / dst_ld;
        const int c = (int)(i % dst_ld);
        dst[i] = c < C ? src[r * src_ld + c] : 0.f;
    }
}
extern "C" int zsg_pad_rows(const float* src, int64_t rows, int32_t C, int32_t src_ld, float* dst, int32_t dst_ld, void* stream) {
    ZSG_REQUIRE(src && dst && rows > 0 && C > 0 && dst_ld >= C && src_ld >= C, "pad_rows: bad argument");
    ZSG_LAUNCH(pad_rows_kernel, dim3(grid_for(rows * dst_ld)), dim3(256), 0, (hipStream_t)stream, src, rows, C, src_ld, dst, dst_ld);
    ZSG_CHECK_LAUNCH("pad_rows");
    return 0;
}

// Column-group interleave between a compact [rows][groups*k] tensor and a strided one:
//   strided[r*groups*gs + a*gs + off + e]  <->  compact[r*groups*k + a*k + e]      (a < groups, e < k)
// dir 0: compact -> strided (separate att / box head outputs into the [B, A, 5] tensor); dir 1: strided -> compact.
__global__ void interleave_kernel(float* __restrict__ compact, int64_t rows, int groups, int k, float* __restrict__ strided, int gs, int off,
                                  int dir) {
    const int64_t total = rows * groups * k;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(i % k);
        const int64_t ra = i / k;                              // r*groups + a
        float* s = strided + ra * gs + off + e;
        if (dir == 0) *s = compact[i];
        else compact[i] = *s;
    }
}
extern "C" int zsg_interleave(float* compact, int64_t rows, int32_t groups, int32_t k, float* strided, int32_t group_stride, int32_t offset,
                              int32_t dir, void* stream) {
    ZSG_REQUIRE(compact && strided && rows > 0 && groups > 0 && k > 0 && offset >= 0 && offset + k <= group_stride, "interleave: bad argument");
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("interleave", st, 0, (double)rows * groups * k * 8);
    ZSG_LAUNCH(interleave_kernel, dim3(grid_for(rows * groups * k)), dim3(256), 0, st, compact, rows, groups, k, strided, group_stride,
                       offset, dir);
    ZSG_CHECK_LAUNCH("interleave");
    return 0;
}

// ---- head conv0 without the spatially-constant channels --------------------------------------------------------------
// The first head convolution sees [features(256) | language vector(256, constant over the image) | grid(2)].  Its
// language part is sum_tap valid(p,tap) * V[b][tap][co] with V = W_lang * we[b] (a tiny GEMM) and its grid part does not
// depend on the batch index, so only the feature channels go through the big implicit GEMM (half the MACs); these
// three kernels build the additive map and reduce the gradient for the two cheap parts.
//   lang_map:     out[b][p][n] = G[p][n] + sum_{tap valid at p} V[b][n*9 + tap]          (3x3, pad 1)
//   border_sums:  S1[b][n*9+tap] = sum_{p: tap valid} dy[b][p][n]   and its transpose S2[n*9+tap][b] (via nine plain sums)
//   batch_sum:    out[i] = sum_b x[b*stride + i]
__global__ void head_lang_map_kernel(const float* __restrict__ V, const float* __restrict__ G, int B, int h, int w, int N,
                                     float* __restrict__ out) {
    const int n4 = N / 4;
    const int64_t total = (int64_t)B * h * w * n4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % n4) * 4;
        const int64_t pix = i / n4;
        const int x = (int)(pix % w);
        const int y = (int)((pix / w) % h);
        const int b = (int)(pix / ((int64_t)w * h));
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (G) acc = *(const f32x4*)(G + ((int64_t)y * w + x) * N + c);
        const float* v = V + (int64_t)b * N * 9 + (int64_t)c * 9;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            if ((unsigned)(y + r - 1) >= (unsigned)h) continue;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if ((unsigned)(x + q - 1) >= (unsigned)w) continue;
                const int t = r * 3 + q;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += v[e * 9 + t];
            }
        }
        *(f32x4*)(out + i * 4) = acc;
    }
}
extern "C" int zsg_head_lang_map(const float* V, const float* G, int32_t B, int32_t h, int32_t w, int32_t N, float* out, void* stream) {
    ZSG_REQUIRE(V && out && B > 0 && h > 0 && w > 0 && N > 0 && (N % 4) == 0, "head_lang_map: bad argument");
    const int64_t n = (int64_t)B * h * w * (N / 4);
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("head_lang_map", st, 0, (double)B * h * w * N * 4);
    ZSG_LAUNCH(head_lang_map_kernel, dim3(grid_for(n)), dim3(256), 0, st, V, G, B, h, w, N, out);
    ZSG_CHECK_LAUNCH("head_lang_map");
    return 0;
}

// The per-forward input staging of ZSGNet.forward in ONE launch (round 5): the reference moves qvec / qlens to the device and draws the
// LSTM's initial state on the host every forward (utils.py:403-405, mdl.py:279-294); as separate torch copies + the BatchNorm counters'
// `add_` these were five ~5 us operations with 5-20 us between them at the head of every forward (rocprofv3: 45 us).  One kernel:
//   qbuf[b][t][:] = t < T ? qvec[b][t][:] : 0   (the plan's zero-padded token bucket);   qlens (int64 or float) -> float;
//   hc_dst = hc_src (h0 | c0; hc_src may be PINNED HOST memory: read over the bus by this kernel, no copy engine);   nbt[i] += 1.
__global__ void stage_inputs_kernel(const float* __restrict__ qvec, int B, int T, int E, int Tplan, float* __restrict__ qbuf,
                                    const void* __restrict__ qlens, int qlens_f32, float* __restrict__ qlens_dst, const float* __restrict__ hc_src, int hc_n,
                                    float* __restrict__ hc_dst, long long* __restrict__ nbt, int n_nbt) {
    const int64_t nq = (int64_t)B * Tplan * E;
    const int64_t total = nq + B + hc_n + n_nbt;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < nq) {
            const int e = (int)(i % E);
            const int t = (int)((i / E) % Tplan);
            const int b = (int)(i / ((int64_t)E * Tplan));
            qbuf[i] = t < T ? qvec[((int64_t)b * T + t) * E + e] : 0.f;
        } else if (i < nq + B) {
            qlens_dst[i - nq] = qlens_f32 ? ((const float*)qlens)[i - nq] : (float)((const long long*)qlens)[i - nq];
        } else if (i < nq + B + hc_n) {
            hc_dst[i - nq - B] = hc_src[i - nq - B];
        } else {
            nbt[i - nq - B - hc_n] += 1;
        }
    }
}
extern "C" int zsg_stage_inputs(const float* qvec, int32_t B, int32_t T, int32_t E, int32_t Tplan, float* qbuf, const void* qlens, int32_t qlens_f32, float* qlens_dst,
                                const float* hc_src, int32_t hc_n, float* hc_dst, int64_t* nbt, int32_t n_nbt, void* stream) {
    ZSG_REQUIRE(qvec && qbuf && qlens && qlens_dst && B > 0 && T > 0 && T <= Tplan && E > 0 && hc_n >= 0 && n_nbt >= 0 && (!hc_n || (hc_src && hc_dst)) &&
                    (!n_nbt || nbt), "stage_inputs: bad argument");
    const int64_t total = (int64_t)B * Tplan * E + B + hc_n + n_nbt;
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("stage_inputs", st, 0, 8.0 * total);
    ZSG_LAUNCH(stage_inputs_kernel, dim3(grid_for(total)), dim3(256), 0, st, qvec, B, T, E, Tplan, qbuf, qlens, qlens_f32, qlens_dst, hc_src, hc_n, hc_dst,
               (long long*)nbt, n_nbt);
    ZSG_CHECK_LAUNCH("stage_inputs");
    return 0;
}

// All pyramid levels of the map in ONE launch (round 5).  The valid taps of a pixel depend only on which image borders it touches:
// row state (top, bottom) x column state (left, right) = 16 classes, the same for every level, so a block first builds the 16 class
// sums S[cls][n] = sum_{tap valid in cls} V[b][n*9+tap] of ITS image in LDS (9 loads per channel instead of up to 36 scalar loads per
// output element) and then streams out[b][p][n] = G[p][n] + S[cls(p)][n] with 16-byte accesses.  Levels are packed level-major
// (level i of `out` starts at B * N * sum_{j<i} h_j w_j, of G at N * sum_{j<i} h_j w_j — mdl.Lowering.packed).
struct LangMapLevels {
    int nlev;
    int h[ZSG_MAX_SEG], w[ZSG_MAX_SEG];
    int p0[ZSG_MAX_SEG + 1];           // first pixel of level i within an image's pixel list (prefix sums of h * w)
};
__global__ __launch_bounds__(256) void head_lang_map_packed_kernel(const float* __restrict__ V, const float* __restrict__ G, int B, int N, LangMapLevels L,
                                                                   float* __restrict__ out, int blocks_per_image) {
    extern __shared__ __attribute__((aligned(16))) float S[];      // [16][N]
    const int b = blockIdx.x / blocks_per_image, part = blockIdx.x % blocks_per_image;
    for (int n = threadIdx.x; n < N; n += 256) {
        float v[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) v[t] = V[(int64_t)b * N * 9 + (int64_t)n * 9 + t];
#pragma unroll
        for (int cls = 0; cls < 16; ++cls) {       // bit 0: top row, 1: bottom row, 2: left column, 3: right column
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                if ((r == 0 && (cls & 1)) || (r == 2 && (cls & 2))) continue;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    if ((q == 0 && (cls & 4)) || (q == 2 && (cls & 8))) continue;
                    a += v[r * 3 + q];
                }
            }
            S[cls * N + n] = a;
        }
    }
    __syncthreads();
    const int n4 = N / 4, P = L.p0[L.nlev];
    const int64_t total = (int64_t)P * n4;
    for (int64_t i = (int64_t)part * 256 + threadIdx.x; i < total; i += (int64_t)blocks_per_image * 256) {
        const int c = (int)(i % n4) * 4;
        const int p = (int)(i / n4);
        int lv = 0;
#pragma unroll
        for (int j = 1; j < ZSG_MAX_SEG; ++j)
            if (j < L.nlev && p >= L.p0[j]) lv = j;
        const int q = p - L.p0[lv], w = L.w[lv], h = L.h[lv];
        const int y = q / w, x = q - y * w;
        const int cls = (y == 0 ? 1 : 0) | (y == h - 1 ? 2 : 0) | (x == 0 ? 4 : 0) | (x == w - 1 ? 8 : 0);
        f32x4 acc = *(const f32x4*)(S + cls * N + c);
        if (G) acc += *(const f32x4*)(G + ((int64_t)L.p0[lv] + q) * N + c);
        *(f32x4*)(out + ((int64_t)B * L.p0[lv] + (int64_t)b * h * w + q) * N + c) = acc;
    }
}
extern "C" int zsg_head_lang_map_packed(const float* V, const float* G, int32_t B, int32_t nlev, const int32_t* hw, int32_t N, float* out, void* stream) {
    ZSG_REQUIRE(V && out && hw && B > 0 && nlev > 0 && nlev <= ZSG_MAX_SEG && N > 0 && (N % 4) == 0 && N <= 2048, "head_lang_map_packed: bad argument");
    LangMapLevels L;
    memset(&L, 0, sizeof(L));
    L.nlev = nlev;
    for (int i = 0; i < nlev; ++i) {
        ZSG_REQUIRE(hw[2 * i] > 0 && hw[2 * i + 1] > 0, "head_lang_map_packed: level %d is empty", i);
        L.h[i] = hw[2 * i];
        L.w[i] = hw[2 * i + 1];
        L.p0[i + 1] = L.p0[i] + L.h[i] * L.w[i];
    }
    const int64_t per = (int64_t)L.p0[nlev] * (N / 4);
    int bpi = (int)((per + 256 * 8 - 1) / (256 * 8));          // ~8 16-byte elements per thread
    const int cap = (4 * ZSG_NUM_CU + B - 1) / B;
    if (bpi > cap) bpi = cap;
    if (bpi < 1) bpi = 1;
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("head_lang_map_packed", st, 0, (double)B * L.p0[nlev] * N * 4);
    ZSG_LAUNCH(head_lang_map_packed_kernel, dim3(B * bpi), dim3(256), (size_t)16 * N * sizeof(float), st, V, G, B, N, L, out, bpi);
    ZSG_CHECK_LAUNCH("head_lang_map_packed");
    return 0;
}

// The nine validity-masked sums follow by inclusion-exclusion from nine plain sums per (image, channel):
//   Q[0] = all pixels, Q[1]/Q[2] = first / last row, Q[3]/Q[4] = first / last column, Q[5..8] = the four corners;
//   S(r,q) = Q0 - R(r) - C(q) + X(r,q)   with R(0) = first row (tap row 0 reads y-1: invalid at y = 0), R(2) = last row, ...
// so the big pass is ONE per-image column sum with 16-byte loads (it is also the bias gradient), the borders are a few
// pixels, and everything accumulates over pyramid levels (the sums are linear).  Q: [9][B][N], zeroed by the caller.
__global__ void head_image_sums_kernel(const float* __restrict__ dy, int rows, int N, int rows_per_block, float* __restrict__ Q0, int CL) {
    __shared__ f32x4 red[256];
    const int RL = 256 / CL;
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    const int col = (blockIdx.x * CL + cl) * 4;
    const int b = blockIdx.z;
    const int r_begin = blockIdx.y * rows_per_block, r_end = min(rows, r_begin + rows_per_block);
    f32x4 s0 = {0, 0, 0, 0}, s1 = s0;
    if (col < N) {
        const float* base = dy + ((int64_t)b * rows) * N + col;
        int r = r_begin + rl;
        for (; r + RL < r_end; r += 2 * RL) {
            s0 += *(const f32x4*)(base + (int64_t)r * N);
            s1 += *(const f32x4*)(base + (int64_t)(r + RL) * N);
        }
        if (r < r_end) s0 += *(const f32x4*)(base + (int64_t)r * N);
    }
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (rl == 0 && col < N) {
        f32x4 s = red[cl];
        for (int k = 1; k < RL; ++k) s += red[k * CL + cl];
        float* o = Q0 + (int64_t)b * N + col;
#pragma unroll
        for (int e = 0; e < 4; ++e) unsafeAtomicAdd(o + e, s[e]);
    }
}
// one block per (image, border line): first / last row, first / last column (and their end pixels = the corners) of this
// level, added to Q[1..8] (levels run one after another on the stream and a (b, n, quantity) element belongs to one
// thread: plain read-modify-write)
__global__ void head_border_lines_kernel(const float* __restrict__ dy, int B, int h, int w, int N, float* __restrict__ Q) {
    const int b = blockIdx.x, line = blockIdx.y;
    const float* img = dy + (int64_t)b * h * w * N;
    const int64_t qs = (int64_t)B * N;
    const bool is_row = line < 2;
    const int len = is_row ? w : h;
    const int64_t first = is_row ? (line == 0 ? 0 : (int64_t)(h - 1) * w) : (line == 2 ? 0 : w - 1);
    const int64_t step = is_row ? 1 : w;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int i = 0;
        for (; i + 3 < len; i += 4) {
            s0 += img[(first + (int64_t)i * step) * N + n];
            s1 += img[(first + (int64_t)(i + 1) * step) * N + n];
            s2 += img[(first + (int64_t)(i + 2) * step) * N + n];
            s3 += img[(first + (int64_t)(i + 3) * step) * N + n];
        }
        for (; i < len; ++i) s0 += img[(first + (int64_t)i * step) * N + n];
        float* q = Q + (int64_t)b * N + n;
        q[(1 + line) * qs] += (s0 + s1) + (s2 + s3);
        if (is_row) {                                 // corners: (0,0), (0,w-1) from the first row; (h-1,0), (h-1,w-1) from the last
            q[(5 + 2 * line) * qs] += img[first * N + n];
            q[(6 + 2 * line) * qs] += img[(first + w - 1) * N + n];
        }
    }
}
extern "C" int zsg_head_border_sums(const float* dy, int32_t B, int32_t h, int32_t w, int32_t N, float* Q, void* stream) {
    ZSG_REQUIRE(dy && Q && B > 0 && h > 0 && w > 0 && N > 0 && (N % 4) == 0, "head_border_sums: bad argument");
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("head_border_sums", st, 0, (double)B * h * w * N * 4);
    const int rows = h * w, c4 = N / 4;
    const int CL = g_zsg_deterministic ? 4 : (c4 >= 16 ? 16 : (c4 >= 8 ? 8 : 4));
    int sp = cdiv(rows, 4 * (256 / CL));
    if (sp > 32) sp = 32;
    if (g_zsg_deterministic) sp = 1;                          // one block per (image, column group): one add per element
    const int rpb = cdiv(rows, sp);
    ZSG_LAUNCH(head_image_sums_kernel, dim3(cdiv(c4, CL), cdiv(rows, rpb), B), dim3(256), 0, st, dy, rows, N, rpb, Q, CL);
    ZSG_LAUNCH(head_border_lines_kernel, dim3(B, 4), dim3(256), 0, st, dy, B, h, w, N, Q);
    ZSG_CHECK_LAUNCH("head_border_sums");
    return 0;
}
// S1[b][n*9 + r*3+q] = Q0 - R(r) - C(q) + X(r,q), S2 = its [n*9+tap][b] transpose, bias_grad[n] += sum_b Q0[b][n] (optional)
__global__ void head_border_finalize_kernel(const float* __restrict__ Q, int B, int N, float* __restrict__ S1, float* __restrict__ S2,
                                            float* __restrict__ bias_grad) {
    const int64_t qs = (int64_t)B * N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * N; i += gridDim.x * blockDim.x) {
        const int b = i / N, n = i - b * N;
        float q[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) q[k] = Q[k * qs + i];
        const float R[3] = {q[1], 0.f, q[2]}, Cc[3] = {q[3], 0.f, q[4]};
        const float X[3][3] = {{q[5], 0.f, q[6]}, {0.f, 0.f, 0.f}, {q[7], 0.f, q[8]}};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = ((q[0] - R[r]) - Cc[c]) + X[r][c];
                const int t = r * 3 + c;
                S1[((int64_t)b * N + n) * 9 + t] = v;
                S2[((int64_t)n * 9 + t) * B + b] = v;
            }
    }
    if (bias_grad)
        for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
            float s = 0.f;
            for (int b = 0; b < B; ++b) s += Q[(int64_t)b * N + n];
            bias_grad[n] += s;
        }
}
extern "C" int zsg_head_border_finalize(const float* Q, int32_t B, int32_t N, float* S1, float* S2, float* bias_grad, void* stream) {
    ZSG_REQUIRE(Q && S1 && S2 && B > 0 && N > 0, "head_border_finalize: bad argument");
    ZSG_LAUNCH(head_border_finalize_kernel, dim3(cdiv(B * N, 256)), dim3(256), 0, (hipStream_t)stream, Q, B, N, S1, S2, bias_grad);
    ZSG_CHECK_LAUNCH("head_border_finalize");
    return 0;
}

__global__ void batch_sum_kernel(const float* __restrict__ x, int B, int64_t stride4, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < stride4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 s = *(const f32x4*)(x + i * 4);
        for (int b = 1; b < B; ++b) s += *(const f32x4*)(x + (b * stride4 + i) * 4);
        *(f32x4*)(out + i * 4) = s;
    }
}
extern "C" int zsg_batch_sum(const float* x, int32_t B, int64_t stride, float* out, void* stream) {
    ZSG_REQUIRE(x && out && B > 0 && stride > 0 && (stride % 4) == 0, "batch_sum: bad argument");
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("batch_sum", st, 0, (double)B * stride * 4);
    ZSG_LAUNCH(batch_sum_kernel, dim3(grid_for(stride / 4)), dim3(256), 0, st, x, B, stride / 4, out);
    ZSG_CHECK_LAUNCH("batch_sum");
    return 0;
}
