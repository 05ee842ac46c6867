// adam.hip — fused Adam over one flat fp32 parameter buffer (torch.optim.Adam semantics, amsgrad off).  HBM-bound:
// reads p,g,m,v and writes p,m,v once: 28 B per parameter.
#include "common.h"

__global__ void adam_tick_kernel(int* step) { *step += 1; }

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n4, int64_t n, float lr, float b1, float b2, float eps,
                                                   float wd, float gs, const int* __restrict__ step) {
    const int t = *step;                                  // already incremented for this step
    const float bc1 = 1.f - powf(b1, (float)t);
    const float bc2s = sqrtf(1.f - powf(b2, (float)t));
    const float step_size = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 pp = *(const f32x4*)(p + 4 * i);
        f32x4 gg = *(const f32x4*)(g + 4 * i) * gs;
        f32x4 mm = *(const f32x4*)(m + 4 * i);
        f32x4 vv = *(const f32x4*)(v + 4 * i);
        if (wd != 0.f) gg += pp * wd;
        mm = mm * b1 + gg * (1.f - b1);
        vv = vv * b2 + gg * gg * (1.f - b2);
#pragma unroll
        for (int e = 0; e < 4; ++e) pp[e] -= step_size * (mm[e] / (sqrtf(vv[e]) / bc2s + eps));
        *(f32x4*)(p + 4 * i) = pp;
        *(f32x4*)(m + 4 * i) = mm;
        *(f32x4*)(v + 4 * i) = vv;
    }
    // tail (n not a multiple of 4)
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) {
        const int64_t i = 4 * n4 + threadIdx.x;
        float gg = g[i] * gs;
        if (wd != 0.f) gg += p[i] * wd;
        const float mm = m[i] * b1 + gg * (1.f - b1);
        const float vv = v[i] * b2 + gg * gg * (1.f - b2);
        p[i] -= step_size * (mm / (sqrtf(vv) / bc2s + eps));
        m[i] = mm;
        v[i] = vv;
    }
}

extern "C" int zsg_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, float grad_scale, int32_t* step_count, void* stream) {
    ZSG_REQUIRE(p && g && m && v && step_count && n > 0, "adam_step: bad argument");
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("adam_step", st, 0, (double)n * 28);
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, step_count);
    const int64_t n4 = n / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > ZSG_NUM_CU * 8) blocks = ZSG_NUM_CU * 8;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((int)blocks), dim3(256), 0, st, p, g, m, v, n4, n, lr, beta1, beta2, eps, weight_decay, grad_scale,
                       step_count);
    ZSG_CHECK_LAUNCH("adam_step");
    return 0;
}
