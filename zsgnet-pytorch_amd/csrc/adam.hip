// adam.hip — fused Adam over one flat fp32 parameter buffer (torch.optim.Adam semantics, amsgrad off).  HBM-bound:
// reads p,g,m,v and writes p,m,v once: 28 B per parameter.
#include "common.h"

// The step counter lives on the device (the launch is hipGraph-capturable): step[0] = steps taken, step[1] = the ticket of the launch
// in flight — both belong to ONE optimizer (two optimizers stepping on different streams never share a ticket).  Every block reads
// step[0] when it starts and uses t = count + 1; the block that FINISHES last (ticket) publishes t and clears the ticket — by then
// every block has read the old value.  (A separate one-thread "tick" launch ahead of the update was 8 us of dependent launch at the
// end of every step.)

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n4, int64_t n, float lr, float b1, float b2, float eps,
                                                   float wd, float gs, int* step, int tick) {
    const int t = __hip_atomic_load(step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
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
    if (!tick) return;          // (a partial update of the step: the launch that covers the rest publishes the counter)
    __syncthreads();
    if (threadIdx.x == 0 && __hip_atomic_fetch_add(step + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
        __hip_atomic_store(step + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(step, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

static int adam_launch(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                       float weight_decay, float grad_scale, int32_t* step_count, int tick, void* stream) {
    ZSG_REQUIRE(p && g && m && v && step_count && n > 0, "adam_step: bad argument");
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("adam_step", st, 0, (double)n * 28);
    const int64_t n4 = n / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > ZSG_NUM_CU * 8) blocks = ZSG_NUM_CU * 8;
    if (blocks < 1) blocks = 1;
    ZSG_LAUNCH(adam_kernel, dim3((int)blocks), dim3(256), 0, st, p, g, m, v, n4, n, lr, beta1, beta2, eps, weight_decay, grad_scale,
                       step_count, tick);
    ZSG_CHECK_LAUNCH("adam_step");
    return 0;
}

extern "C" int zsg_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, float grad_scale, int32_t* step_count, void* stream) {
    return adam_launch(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, grad_scale, step_count, 1, stream);
}

// One optimizer step as several launches over disjoint ranges of the flat buffer (so that the part whose gradients are complete can
// be updated while the last weight gradients are still being computed): every launch of the step computes with t = counter + 1;
// exactly the LAST one passes publish = 1 and advances the counter.
extern "C" int zsg_adam_step_range(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                                   float weight_decay, float grad_scale, int32_t* step_count, int32_t publish, void* stream) {
    return adam_launch(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, grad_scale, step_count, publish ? 1 : 0, stream);
}
