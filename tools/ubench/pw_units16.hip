// Micro-benchmark (developer tool, NOT part of libzsg): the filter-resident streaming scheme of csrc/pw.hip on layer1's conv3 shape
// (M = 90000 pixels, K = 64 -> N = 256) with
//   ROWS = 32: units of 32 pixels x 64 channels, v_mfma_f32_32x32x2_f32, 8 waves per workgroup = 2 per SIMD (what pw_kernel<2, 0> runs);
//   ROWS = 16: units of 16 pixels x 64 channels, v_mfma_f32_16x16x4_f32, 16 waves per workgroup = 4 per SIMD — the same 139 KB of LDS
//              (16 wave buffers of 4.35 KB instead of 8 of 8.7 KB), twice the waves to hide one wave's loads / transposition / stores
//              under the others' MFMAs, at twice the fragment reads per MFMA.
// Question for the next round: does the MFMA phase overlap the memory phases with four waves per SIMD?
// build: hipcc -O3 --offload-arch=gfx950 pw_units16.hip -o pw_units16 ; run: ./pw_units16
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define OOB 0xFFFFFFF0u
constexpr int K = 64, N = 256, LDW = 68, TBP = 68;

__device__ __forceinline__ rsrc_t mk(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x80000000u, 0x00020000); }
__device__ __forceinline__ f32x4 ld4(rsrc_t r, unsigned off) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}
__device__ __forceinline__ void st4(rsrc_t r, unsigned off, f32x4 v) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)off, 0, 0);
}
__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int ROWS>
__global__ __launch_bounds__(ROWS == 32 ? 512 : 1024) void kern(const float* src, const float* wt, float* out, int M) {
    constexpr int WAVES = ROWS == 32 ? 8 : 16;
    constexpr int RL = ROWS / 4;               // 16-byte row loads per lane per slice
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* Ws = smem;
    float* Tb = smem + N * LDW + wave * (ROWS * TBP);
    const int cg = lane & 15, rr = lane >> 4;
    const int RT = (M + ROWS - 1) / ROWS, G = gridDim.x;
    auto tile_of = [=](int l) { return (int)blockIdx.x + G * (l >> 2); };      // four 64-channel units per row tile
    const int n0 = (wave & 3) * 64;
    const rsrc_t rs = mk(src), ro = mk(out);
    f32x4 R[2][RL];
    int req = wave;
    auto request = [rs, rr, cg, M, RT, tile_of](f32x4 (&r)[RL], int ru) {
        const int rt = tile_of(ru);
#pragma unroll
        for (int i = 0; i < RL; ++i) {
            const int m = rt * ROWS + rr + 4 * i;
            r[i] = ld4(rs, (rt < RT && m < M) ? 4u * (unsigned)(m * K + 4 * cg) : OOB);
        }
    };
    auto park = [Tb, rr, cg](const f32x4 (&r)[RL]) {
#pragma unroll
        for (int i = 0; i < RL; ++i) *(f32x4*)(Tb + (rr + 4 * i) * TBP + 4 * cg) = r[i];
    };
    request(R[0], req);
    req += WAVES;
    request(R[1], req);
    req += WAVES;
    {
        const rsrc_t rw = mk(wt);
        constexpr int PER = N * 16 / (64 * WAVES);
        f32x4 t[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int idx = tid + j * 64 * WAVES;
            t[j] = ld4(rw, 4u * (unsigned)((idx >> 4) * K + 4 * (idx & 15)));
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int idx = tid + j * 64 * WAVES;
            *(f32x4*)(Ws + (idx >> 4) * LDW + 4 * (idx & 15)) = t[j];
        }
    }
    park(R[0]);
    request(R[0], req);
    req += WAVES;
    __syncthreads();

    int u = wave;
    while (tile_of(u) < RT) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            if (tile_of(u) >= RT) break;
            f32x4(&nxt)[RL] = R[(par + 1) % 2];
            const int m0 = tile_of(u) * ROWS;
            unsigned off[RL];
#pragma unroll
            for (int i = 0; i < RL; ++i) {
                const int m = m0 + rr + 4 * i;
                off[i] = m < M ? 4u * (unsigned)(m * N + n0 + 4 * cg) : OOB;
            }
            f32x4 v[RL];
            wsync();
            if constexpr (ROWS == 32) {
                const int li = lane & 31, lh = lane >> 5;
                f32x16 acc[2];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
                const float* a = Tb + li * TBP + 4 * lh;
                const float* b = Ws + (n0 + li) * LDW + 4 * lh;
                f32x4 fa = *(const f32x4*)a, fb[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = *(const f32x4*)(b + j * 32 * LDW);
#pragma unroll
                for (int kq = 0; kq < 8; ++kq) {
                    f32x4 fan = fa, fbn[2] = {fb[0], fb[1]};
                    if (kq < 7) {
                        fan = *(const f32x4*)(a + (kq + 1) * 8);
#pragma unroll
                        for (int j = 0; j < 2; ++j) fbn[j] = *(const f32x4*)(b + j * 32 * LDW + (kq + 1) * 8);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j][e], fa[e], acc[j], 0, 0, 0);
                    fa = fan;
                    fb[0] = fbn[0];
                    fb[1] = fbn[1];
                }
                wsync();
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *(f32x4*)(Tb + li * TBP + jj * 32 + 8 * q + 4 * lh) = (f32x4){acc[jj][4 * q], acc[jj][4 * q + 1], acc[jj][4 * q + 2], acc[jj][4 * q + 3]};
            } else {
                const int li = lane & 15, lg = lane >> 4;
                f32x4 acc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const float* a = Tb + li * TBP + 4 * lg;
                const float* b = Ws + (n0 + li) * LDW + 4 * lg;
                f32x4 fa = *(const f32x4*)a, fb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = *(const f32x4*)(b + j * 16 * LDW);
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) {
                    f32x4 fan = fa, fbn[4] = {fb[0], fb[1], fb[2], fb[3]};
                    if (kq < 3) {
                        fan = *(const f32x4*)(a + (kq + 1) * 16);
#pragma unroll
                        for (int j = 0; j < 4; ++j) fbn[j] = *(const f32x4*)(b + j * 16 * LDW + (kq + 1) * 16);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j][e], fa[e], acc[j], 0, 0, 0);
                    fa = fan;
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[j] = fbn[j];
                }
                wsync();
                // lane (i = pixel, g): acc[j] = channels j * 16 + 4 g .. + 3 of pixel i
#pragma unroll
                for (int j = 0; j < 4; ++j) *(f32x4*)(Tb + li * TBP + j * 16 + 4 * lg) = acc[j];
            }
            wsync();
#pragma unroll
            for (int i = 0; i < RL; ++i) v[i] = *(const f32x4*)(Tb + (rr + 4 * i) * TBP + 4 * cg);
            wsync();
            park(nxt);
            request(nxt, req);
            req += WAVES;
#pragma unroll
            for (int i = 0; i < RL; ++i) st4(ro, off[i], v[i]);
            u += WAVES;
        }
    }
}

template <int ROWS>
static void run(const float* src, const float* wt, float* out, int M, const std::vector<float>& hs, const std::vector<float>& hw) {
    constexpr int WAVES = ROWS == 32 ? 8 : 16;
    const size_t lds = (size_t)(N * LDW + WAVES * ROWS * TBP) * sizeof(float);
    (void)hipFuncSetAttribute((const void*)kern<ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipMemset(out, 0xff, (size_t)M * N * sizeof(float));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern<ROWS>, dim3(256), dim3(64 * WAVES), lds, 0, src, wt, out, M);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        printf("ROWS=%d: %s\n", ROWS, hipGetErrorString(e));
        return;
    }
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    float single[15], ms;
    for (int i = 0; i < 15; ++i) {                     // single launches, device idle before each (as tools/pw_bench.py)
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL(kern<ROWS>, dim3(256), dim3(64 * WAVES), lds, 0, src, wt, out, M);
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&single[i], a, b);
    }
    for (int i = 0; i < 15; ++i)
        for (int j = i + 1; j < 15; ++j)
            if (single[j] < single[i]) { float t = single[i]; single[i] = single[j]; single[j] = t; }
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern<ROWS>, dim3(256), dim3(64 * WAVES), lds, 0, src, wt, out, M);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    (void)hipEventElapsedTime(&ms, a, b);
    // check 97 sampled rows (incl. the last) against a double-precision host product
    double worst = 0;
    std::vector<float> row(N);
    for (int s = 0; s < 97; ++s) {
        const int m = s == 96 ? M - 1 : (int)(((long long)s * 928187) % M);
        (void)hipMemcpy(row.data(), out + (size_t)m * N, N * sizeof(float), hipMemcpyDeviceToHost);
        for (int n = 0; n < N; ++n) {
            double r = 0;
            for (int k = 0; k < K; ++k) r += (double)hs[(size_t)m * K + k] * hw[(size_t)n * K + k];
            worst = fmax(worst, fabs(r - row[n]));
        }
    }
    printf("ROWS=%2d (%2d waves / workgroup): single launch %.1f us (median of 15), back to back %.1f us per launch; max |err| on 97 rows %.2e\n", ROWS, WAVES,
           single[7] * 1e3, ms / 20 * 1e3, worst);
}

int main() {
    const int M = 90000;
    std::vector<float> hs((size_t)M * K), hw((size_t)N * K);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hs) v = rnd();
    for (auto& v : hw) v = rnd() * 0.25f;
    float *src, *wt, *out;
    (void)hipMalloc(&src, hs.size() * sizeof(float));
    (void)hipMalloc(&wt, hw.size() * sizeof(float));
    (void)hipMalloc(&out, (size_t)M * N * sizeof(float));
    (void)hipMemcpy(src, hs.data(), hs.size() * sizeof(float), hipMemcpyHostToDevice);
    (void)hipMemcpy(wt, hw.data(), hw.size() * sizeof(float), hipMemcpyHostToDevice);
    printf("filter-resident streaming GEMM, M = %d, K = %d -> N = %d (18.7 us of fp32 MFMA, 115 MB of HBM traffic)\n", M, K, N);
    run<32>(src, wt, out, M, hs, hw);
    run<16>(src, wt, out, M, hs, hw);
    run<32>(src, wt, out, M, hs, hw);
    run<16>(src, wt, out, M, hs, hw);
    return 0;
}
