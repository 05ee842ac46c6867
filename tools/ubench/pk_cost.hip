// Micro-benchmark (developer tool, round 6): what does a PACKED fp32 VALU instruction (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32: two
// lanes' worth of arithmetic per instruction) cost a wave next to its own fp32 MFMA stream, against the scalar form (~6 cycles each,
// tools/ubench/mfma_coissue.hip)?  One wave per SIMD, [4 MFMAs + K fillers] per iteration, as part 2 of mfma_coissue.
// build: hipcc -O3 --offload-arch=gfx950 pk_cost.hip -o pk_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
enum { F_FMA = 1, F_PKFMA, F_PKADD, F_PKMUL, F_ADD, F_MOV64 };
static const char* fname[] = {"nothing", "v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_add_f32", "v_mov_b64"};

template <int T, int K>
__global__ __launch_bounds__(256) void k1(float* __restrict__ out, int nm) {
    const int tid = threadIdx.x;
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float fa = tid * 1e-9f, fb = 1.0f;
    float x0 = tid, x1 = 1.f, x2 = 2.f, x3 = 3.f;
    f32x2 p0 = {1.f, 2.f}, p1 = {3.f, 4.f}, p2 = {(float)tid, 1.f}, p3 = {0.5f, 0.25f};
    for (int i = 0; i < nm; i += 4) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, a3, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < K / 2; ++r) {
            if (T == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %1, %2" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
            if (T == F_ADD) asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
            if (T == F_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %3, %3, %1, %2" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
            if (T == F_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %2, %2, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
            if (T == F_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %2, %2, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
            if (T == F_MOV64) asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %2, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
        }
    }
    float v = 0;
    for (int e = 0; e < 16; ++e) v += a0[e] + a1[e] + a2[e] + a3[e];
    v += x0 + x1 + x2 + x3 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1];
    if (v == 12345.678f) out[tid] = v;
}
static float* g_out;
static const int NM = 16384;
template <typename F>
static float timeit(F launch) {
    for (int i = 0; i < 2; ++i) launch();
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}
template <int T, int K>
static void run() {
    const float ms = timeit([&] { hipLaunchKernelGGL((k1<T, K>), dim3(256), dim3(256), 0, 0, g_out, NM); });
    printf("own fillers %-14s K=%2d per 4 MFMAs: %6.2f ns/MFMA (ideal 26.67)\n", fname[T], K, ms * 1e6 / NM);
}
int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    hipMalloc(&g_out, 1 << 16);
    run<F_FMA, 0>();
    run<F_FMA, 8>(); run<F_FMA, 16>(); run<F_FMA, 32>();
    run<F_ADD, 16>();
    run<F_PKFMA, 8>(); run<F_PKFMA, 16>(); run<F_PKFMA, 32>();
    run<F_PKADD, 8>(); run<F_PKADD, 16>(); run<F_PKADD, 32>();
    run<F_PKMUL, 16>();
    run<F_MOV64, 16>();
    return 0;
}
