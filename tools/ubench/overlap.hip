// Micro-benchmark (developer tool): does operand delivery overlap fp32 MFMA execution inside ONE wave per SIMD?
//   mode 0: MFMA only            (NM dependent v_mfma_f32_32x32x2_f32 per step)
//   mode 1: loads via VGPRs      (4 x buffer_load_dwordx4 per lane per step, two steps ahead -> ds_write_b128 -> barrier -> ds_read_b128)
//   mode 2: mode 1 + mode 0 in the same loop (what igemm_kernel does)
//   mode 3: loads via LDS-DMA    (4 x buffer_load_dwordx4 ... lds per wave per step, one step ahead, no VGPR, no ds_write)
//   mode 4: mode 3 + mode 0
// One 256-thread block per CU (grid = 256), every block streams its own 16 KB per step from an L2/MALL-resident buffer.
// build: hipcc -O3 --offload-arch=gfx950 overlap.hip -o overlap ; run: ./overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) float lds_f32;

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* __restrict__ out, int steps, int nm, size_t blk_stride) {
    __shared__ __attribute__((aligned(16))) float lds[2][4096 + 64];      // 2 x 16 KB (+ pad)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src + blockIdx.x * blk_stride), 0, 0x80000000u, 0x00020000);
    f32x16 acc = {0};
    f32x4 r0[4], r1[4];
    float fa = tid * 1e-9f, fb = 1.0f;
    auto load = [&](f32x4 (&r)[4], int s) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((s * 4 + j) * 256 + tid) * 16), 0, 0);
            r[j] = __builtin_bit_cast(f32x4, v);
        }
    };
    auto dma = [&](int buf, int s) {
#if __HIP_DEVICE_COMPILE__
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_f32*)(&lds[buf][(j * 4 + wave) * 256]), 16, (int)(((s * 4 + j) * 256 + wave * 64 + lane) * 16), 0, 0, 0);
#endif
    };
    auto store = [&](int buf, const f32x4 (&r)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *(f32x4*)&lds[buf][(j * 256 + tid) * 4] = r[j];
    };
    auto mfma = [&](int buf) {
        const f32x4 a = *(const f32x4*)&lds[buf][lane * 4], b = *(const f32x4*)&lds[buf][2048 + lane * 4];
        if (MODE == 1 || MODE == 3) { fa += a[0] + b[1]; return; }
        fa += a[0] * 1e-30f; fb += b[0] * 1e-30f;
        for (int i = 0; i < nm; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
    };
    if (MODE == 0) {
        for (int s = 0; s < steps; ++s) {
            for (int i = 0; i < nm; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
            __syncthreads();
        }
    } else if (MODE == 1 || MODE == 2) {
        load(r0, 0); store(0, r0); load(r0, 1);
        __syncthreads();
        for (int s = 0; s < steps; s += 2) {
            load(r1, s + 2); mfma(0); store(1, r0); __syncthreads();
            load(r0, s + 3); mfma(1); store(0, r1); __syncthreads();
        }
    } else {
        dma(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int s = 0; s < steps; s += 2) {
            dma(1, s + 1); mfma(0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();
            dma(0, s + 2); mfma(1); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();
        }
    }
    float v = fa + fb;
    for (int e = 0; e < 16; ++e) v += acc[e];
    if (v == 12345.678f) out[blockIdx.x * 256 + tid] = v;
}

template <int MODE>
float run(const float* src, float* out, int steps, int nm, size_t bs, int grid) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, src, out, steps, nm, bs);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, src, out, steps, nm, bs);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f;
}

int main(int argc, char** argv) {
    const int steps = 256, nm = 16;
    const size_t per_block = (size_t)(steps + 4) * 4096;          // floats streamed per block (16 KB per step)
    for (int grid : {64, 128, 256, 512}) {
        float *src, *out;
        hipMalloc(&src, per_block * grid * sizeof(float));
        hipMalloc(&out, grid * 256 * sizeof(float));
        hipMemset(src, 0, per_block * grid * sizeof(float));
        printf("grid %3d blocks x 256 threads, %d steps, 16 KB + %d MFMA per wave per step (MFMA alone = %.3f us/step at 2.4 GHz)\n", grid, steps, nm, nm * 64 / 2400.0);
        float t0 = run<0>(src, out, steps, nm, per_block, grid), t1 = run<1>(src, out, steps, nm, per_block, grid), t2 = run<2>(src, out, steps, nm, per_block, grid);
        float t3 = run<3>(src, out, steps, nm, per_block, grid), t4 = run<4>(src, out, steps, nm, per_block, grid);
        printf("  per step: mfma %.3f | vgpr-staged loads %.3f | both %.3f  (sum %.3f)   || lds-dma loads %.3f | both %.3f (sum %.3f) us\n", t0 / steps, t1 / steps,
               t2 / steps, (t0 + t1) / steps, t3 / steps, t4 / steps, (t0 + t3) / steps);
        hipFree(src); hipFree(out);
    }
    return 0;
}
