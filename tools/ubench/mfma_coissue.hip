// Micro-benchmark (developer tool): who can issue what while a wave streams fp32 MFMAs on a SIMD (gfx950)?
//  Part 1 (two waves per SIMD): one 512-thread block per CU; the "matrix" waves (one per SIMD) run NM MFMAs back to back, their
//    SIMD partners spin on ONE instruction class until the matrix waves are done.  Reported: ns per MFMA and partner instructions
//    retired per MFMA.  Variants: which half of the block is the matrix half (older / younger waves), partner at s_setprio 3,
//    MFMA flavour (32x32x2 f32, four independent accumulators | one dependent accumulator | 16x16x4 f32 | 32x32x16 bf16).
//  Part 2 (one wave per SIMD): 256-thread blocks, every wave runs [4 MFMAs + K filler instructions of one class] per iteration:
//    how many fillers hide in the shadow of its OWN fp32 MFMAs?
// build: hipcc -O3 --offload-arch=gfx950 mfma_coissue.hip -o mfma_coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

enum { F_FMA = 1, F_DPP, F_ADD, F_DSR, F_DSW, F_VMEM, F_SALU, F_NOP };
static const char* fname[] = {"nothing", "v_fma_f32", "v_mov_dpp", "v_add_u32", "ds_read_b128", "ds_write_b128", "buffer_load_x4", "s_add_u32", "s_nop"};

#define FILL2(T)                                                                                                                         \
    {                                                                                                                                    \
        if (T == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %1, %2" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));       \
        if (T == F_DPP) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); \
        if (T == F_ADD) asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %0" : "+v"(iv), "+v"(iw));                                  \
        if (T == F_DSR) asm volatile("ds_read_b128 %0, %1\n ds_read_b128 %0, %1 offset:4096" : "=v"(q) : "v"(laddr));                       \
        if (T == F_DSW) asm volatile("ds_write_b128 %1, %0 offset:16384\n ds_write_b128 %1, %0 offset:24576" ::"v"(q), "v"(laddr));         \
        if (T == F_VMEM) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen\n buffer_load_dwordx4 %0, %1, %2, 0 offen offset:1024" : "=v"(q) : "v"(tid * 16), "s"(rs)); \
        if (T == F_SALU) asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1" : "+s"(si) : : "scc");                                              \
        if (T == F_NOP) asm volatile("s_nop 0\n s_nop 0");                                                                                \
    }

// FLAV: 0 f32 32x32x2 x4 accumulators, 1 f32 32x32x2 one accumulator, 2 f32 16x16x4 x4 accumulators, 3 bf16 32x32x16 x4
template <int FLAV>
__device__ __forceinline__ void mfma4(f32x16& a0, f32x16& a1, f32x16& a2, f32x16& a3, f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3, float fa, float fb, bf16x8 ha,
                                      bf16x8 hb) {
    if (FLAV == 0) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, a3, 0, 0, 0);
    } else if (FLAV == 1) {
        for (int i = 0; i < 4; ++i) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, a0, 0, 0, 0);
    } else if (FLAV == 2) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c3, 0, 0, 0);
    } else {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, a3, 0, 0, 0);
    }
}

// ROLE 1: matrix waves = 0-3 (older), 2: matrix waves = 4-7 (younger), 3: as 1 with the partner at s_setprio 3, 4: as 1 with the matrix wave at s_setprio 3
template <int T, int FLAV, int ROLE>
__global__ __launch_bounds__(512) void k2(const float* __restrict__ src, float* __restrict__ out, unsigned* cnt, int nm) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    __shared__ int done;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) done = 0;
    for (int i = tid; i < 8192; i += 512) lds[i] = i * 1e-6f;
    __syncthreads();
    const bool matrix = (ROLE == 2) ? (wave >= 4) : (wave < 4);
    if (matrix) {
        if (ROLE == 4) __builtin_amdgcn_s_setprio(3);
        f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        float fa = tid * 1e-9f, fb = 1.0f;
        bf16x8 ha, hb;
        for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(tid * 1e-3f); hb[i] = (__bf16)1.0f; }
        for (int i = 0; i < nm; i += 4) mfma4<FLAV>(a0, a1, a2, a3, c0, c1, c2, c3, fa, fb, ha, hb);
        float v = 0;
        for (int e = 0; e < 16; ++e) v += a0[e] + a1[e] + a2[e] + a3[e];
        v += c0[0] + c1[1] + c2[2] + c3[3];
        if (lane == 0) __hip_atomic_fetch_add(&done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (v == 12345.678f) out[tid] = v;
    } else {
        if (T == 0) return;
        if (ROLE == 3) __builtin_amdgcn_s_setprio(3);
        unsigned n = 0;
        float x0 = tid, x1 = 1.f, x2 = 2.f, x3 = 3.f;
        f32x4 q = {1, 2, 3, 4};
        int iv = tid, iw = 1, si = blockIdx.x;
        const int laddr = (wave & 3) * 1024 + lane * 16;
        const unsigned long long ba = (unsigned long long)src;
        const i32x4 rs = {(int)(unsigned)ba, (int)((ba >> 32) & 0xffffu), (int)0x80000000u, 0x00020000};
        while (__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4 && n < (1u << 22)) {      // (bounded: an older spinning partner starves the matrix wave)
#pragma unroll
            for (int r = 0; r < 16; ++r) FILL2(T);
            if (T == F_DSR || T == F_VMEM) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            n += 32;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (lane == 0) cnt[blockIdx.x * 4 + (wave & 3)] = n;
        if (x0 + x1 + x2 + x3 + q[0] + q[1] + q[2] + q[3] + iv + iw + si == 12345.678f) out[tid] = x0;
    }
}

// one wave per SIMD: [4 MFMAs + K fillers] per iteration
template <int T, int FLAV, int K>
__global__ __launch_bounds__(256) void k1(const float* __restrict__ src, float* __restrict__ out, int nm) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 8192; i += 256) lds[i] = i * 1e-6f;
    __syncthreads();
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float fa = tid * 1e-9f, fb = 1.0f;
    bf16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(tid * 1e-3f); hb[i] = (__bf16)1.0f; }
    float x0 = tid, x1 = 1.f, x2 = 2.f, x3 = 3.f;
    f32x4 q = {1, 2, 3, 4};
    int iv = tid, iw = 1, si = blockIdx.x;
    const int laddr = (wave & 3) * 1024 + lane * 16;
    const unsigned long long ba = (unsigned long long)src;
    const i32x4 rs = {(int)(unsigned)ba, (int)((ba >> 32) & 0xffffu), (int)0x80000000u, 0x00020000};
    for (int i = 0; i < nm; i += 4) {
        mfma4<FLAV>(a0, a1, a2, a3, c0, c1, c2, c3, fa, fb, ha, hb);
#pragma unroll
        for (int r = 0; r < K / 2; ++r) FILL2(T);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float v = 0;
    for (int e = 0; e < 16; ++e) v += a0[e] + a1[e] + a2[e] + a3[e];
    v += c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + x2 + x3 + q[0] + q[1] + q[2] + q[3] + iv + iw + si;
    if (v == 12345.678f) out[tid] = v;
}

static float *g_src, *g_out;
static unsigned* g_cnt;
static const int NM = 16384;
static const char* flav[] = {"f32 32x32x2 x4acc", "f32 32x32x2 1acc", "f32 16x16x4 x4acc", "bf16 32x32x16 x4acc"};
static const double ideal_ns[] = {64 / 2.4, 64 / 2.4, 32 / 2.4, 32 / 2.4};

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

template <int T, int FLAV, int ROLE>
static void part1() {
    hipMemset(g_cnt, 0, 1024 * 4);
    const float ms = timeit([&] { hipLaunchKernelGGL((k2<T, FLAV, ROLE>), dim3(256), dim3(512), 0, 0, g_src, g_out, g_cnt, NM); });
    static unsigned hn[1024];
    hipMemcpy(hn, g_cnt, sizeof(hn), hipMemcpyDeviceToHost);
    double sn = 0;
    for (int i = 0; i < 1024; ++i) sn += hn[i];
    static const char* roles[] = {"", "matrix=older half", "matrix=younger half", "partner setprio 3", "matrix setprio 3"};
    printf("P1 %-20s %-20s partner %-15s: %6.2f ns/MFMA (ideal %5.2f)  partner instr/MFMA %6.2f\n", flav[FLAV], roles[ROLE], fname[T], ms * 1e6 / NM, ideal_ns[FLAV],
           sn / 1024 / NM);
}
template <int T, int FLAV, int K>
static void part2() {
    const float ms = timeit([&] { hipLaunchKernelGGL((k1<T, FLAV, K>), dim3(256), dim3(256), 0, 0, g_src, g_out, NM); });
    printf("P2 %-20s own fillers %-15s K=%2d per 4 MFMAs: %6.2f ns/MFMA (ideal %5.2f)\n", flav[FLAV], fname[T], K, ms * 1e6 / NM, ideal_ns[FLAV]);
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    hipMalloc(&g_src, 1 << 20); hipMalloc(&g_out, 1 << 16); hipMalloc(&g_cnt, 1024 * 4);
    hipMemset(g_src, 0, 1 << 20);
    part1<0, 0, 1>();
    part1<F_FMA, 0, 1>(); part1<F_DPP, 0, 1>(); part1<F_ADD, 0, 1>(); part1<F_DSR, 0, 1>(); part1<F_DSW, 0, 1>(); part1<F_VMEM, 0, 1>(); part1<F_SALU, 0, 1>(); part1<F_NOP, 0, 1>();
    part1<F_FMA, 0, 2>(); part1<F_DSR, 0, 2>(); part1<F_VMEM, 0, 2>(); part1<F_SALU, 0, 2>();
    part1<F_FMA, 0, 3>(); part1<F_DSR, 0, 3>(); part1<F_VMEM, 0, 3>();
    part1<F_FMA, 0, 4>(); part1<F_DSR, 0, 4>();
    part1<F_FMA, 1, 1>(); part1<F_DSR, 1, 1>(); part1<F_VMEM, 1, 1>();
    part1<0, 2, 1>(); part1<F_FMA, 2, 1>(); part1<F_DSR, 2, 1>(); part1<F_VMEM, 2, 1>();
    part1<0, 3, 1>(); part1<F_FMA, 3, 1>(); part1<F_DSR, 3, 1>(); part1<F_VMEM, 3, 1>();
    part2<F_FMA, 0, 0>();
    part2<F_FMA, 0, 4>(); part2<F_FMA, 0, 8>(); part2<F_FMA, 0, 16>(); part2<F_FMA, 0, 32>();
    part2<F_DPP, 0, 16>(); part2<F_ADD, 0, 16>();
    part2<F_SALU, 0, 8>(); part2<F_SALU, 0, 16>(); part2<F_SALU, 0, 32>(); part2<F_SALU, 0, 64>();
    part2<F_NOP, 0, 16>(); part2<F_NOP, 0, 64>();
    part2<F_DSR, 0, 2>(); part2<F_DSR, 0, 4>(); part2<F_DSR, 0, 8>(); part2<F_DSR, 0, 16>();
    part2<F_DSW, 0, 2>(); part2<F_DSW, 0, 4>(); part2<F_DSW, 0, 8>();
    part2<F_VMEM, 0, 2>(); part2<F_VMEM, 0, 4>(); part2<F_VMEM, 0, 8>();
    part2<F_FMA, 2, 0>(); part2<F_FMA, 2, 8>(); part2<F_FMA, 2, 16>(); part2<F_SALU, 2, 16>(); part2<F_DSR, 2, 8>();
    part2<F_FMA, 3, 0>(); part2<F_FMA, 3, 8>(); part2<F_FMA, 3, 16>(); part2<F_SALU, 3, 16>(); part2<F_DSR, 3, 8>();
    return 0;
}
