// Micro-benchmark 2 (developer tool): LDS-DMA operand delivery with a THREE-buffer ring, two tiles in flight, counted vmcnt and raw
// barriers, alone and under fp32 MFMAs; plus what an out-of-range LDS-DMA lane writes.
// build: hipcc -O3 --offload-arch=gfx950 overlap2.hip -o overlap2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) float lds_f32;

template <int MODE>      // 0 mfma only, 1 dma ring only, 2 both
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* __restrict__ out, int steps, int nm, size_t blk_stride) {
    __shared__ __attribute__((aligned(16))) float lds[3][4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src + blockIdx.x * blk_stride), 0, 0x80000000u, 0x00020000);
    f32x16 acc = {0};
    float fa = tid * 1e-9f, fb = 1.0f;
    auto dma = [&](int buf, int s) {
#if __HIP_DEVICE_COMPILE__
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_f32*)(&lds[buf][(j * 4 + wave) * 256]), 16, (int)(((s * 4 + j) * 256 + wave * 64 + lane) * 16), 0, 0, 0);
#endif
    };
    if (MODE != 0) {
        dma(0, 0);
        dma(1, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    for (int s = 0; s < steps; ++s) {
        const int buf = s % 3;
        if (MODE != 0) dma((s + 2) % 3, s + 2);
        f32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
        if (MODE != 0) {
            a = *(const f32x4*)&lds[buf][lane * 4];
            b = *(const f32x4*)&lds[buf][2048 + lane * 4];
        }
        if (MODE == 1) fa += a[0] + b[1];
        else {
            fa += a[0] * 1e-30f; fb += b[0] * 1e-30f;
            for (int i = 0; i < nm; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
        }
        if (MODE != 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float v = fa + fb;
    for (int e = 0; e < 16; ++e) v += acc[e];
    if (v == 12345.678f) out[blockIdx.x * 256 + tid] = v;
}

__global__ void oob_probe(const float* src, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = -7.0f;
    __syncthreads();
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1024u, 0x00020000);      // 1 KB window
#if __HIP_DEVICE_COMPILE__
    // lanes 0..31 in range, lanes 32..63 out of range (offset beyond num_records)
    const unsigned off = (threadIdx.x < 32) ? threadIdx.x * 16 : 0xFFFFFFF0u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_f32*)lds, 16, (int)off, 0, 0, 0);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}

template <int MODE>
float run(const float* src, float* out, int steps, int nm, size_t bs, int grid) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, src, out, steps, nm, bs);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, src, out, steps, nm, bs);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f;
}

int main() {
    const int steps = 256, nm = 16;
    const size_t per_block = (size_t)(steps + 4) * 4096;
    for (int grid : {64, 128, 200, 256, 512}) {
        float *src, *out;
        (void)hipMalloc(&src, per_block * grid * sizeof(float));
        (void)hipMalloc(&out, grid * 256 * sizeof(float));
        (void)hipMemset(src, 0, per_block * grid * sizeof(float));
        float t0 = run<0>(src, out, steps, nm, per_block, grid), t1 = run<1>(src, out, steps, nm, per_block, grid), t2 = run<2>(src, out, steps, nm, per_block, grid);
        printf("grid %3d: per step  mfma %.3f | lds-dma ring (2 tiles ahead) %.3f | both %.3f  (sum %.3f) us\n", grid, t0 / steps, t1 / steps, t2 / steps, (t0 + t1) / steps);
        (void)hipFree(src); (void)hipFree(out);
    }
    float *src, *out, h[256], hs[256];
    (void)hipMalloc(&src, 4096); (void)hipMalloc(&out, 1024);
    for (int i = 0; i < 256; ++i) hs[i] = 1.0f + i;
    (void)hipMemcpy(src, hs, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(oob_probe, dim3(1), dim3(64), 0, 0, src, out);
    (void)hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost);
    printf("oob probe: in-range lane 3 -> %.1f %.1f (expect 13 14); out-of-range lane 40 slot -> %.1f %.1f (was -7: zero = written as 0, -7 = skipped)\n", h[12], h[13], h[160], h[161]);
    return 0;
}
