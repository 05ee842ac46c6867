// Micro-benchmark (developer tool): the BatchNorm finalize launch (partial rows -> per-channel statistics).  88 of these sit on the
// step's dependent chain at 6-7.6 us each (profiles/r03_rocprofv3_kernel_stats_serial.csv); how short can the launch be?
// Every timed finalize follows a producer launch that rewrites the partial rows from all CUs (as the convolution epilogue does), so
// the rows come from HBM / the Infinity Cache, not a warm L2.  Reported: (producer + finalize) - producer, per launch.
// build: hipcc -O3 --offload-arch=gfx950 bn_finalize.hip -o bn_finalize
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ void producer(float* part, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) part[i] = v + (float)(i & 1023) * 1e-3f;
}

// V0: the shipped kernel — one wave per channel quad, four quads per block, rows strided by 64, "#pragma unroll 4".
__global__ __launch_bounds__(256) void fin_v0(const float* __restrict__ part, int chunks, int C, double rows, float* mean, float* invstd) {
    const int c = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    const int lane = threadIdx.x & 63;
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    if (c < C) {
#pragma unroll 4
        for (int k = lane; k < chunks; k += 64) {
            const f32x4 a = *(const f32x4*)(part + (size_t)k * 2 * C + c);
            const f32x4 b = *(const f32x4*)(part + (size_t)k * 2 * C + C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[e] += (double)a[e]; ss[e] += (double)b[e]; }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { s[e] = wave_sum_d(s[e]); ss[e] = wave_sum_d(ss[e]); }
    const int e = lane;
    if (e >= 4 || c + e >= C) return;
    const double se = (e == 0) ? s[0] : ((e == 1) ? s[1] : ((e == 2) ? s[2] : s[3]));
    const double sse = (e == 0) ? ss[0] : ((e == 1) ? ss[1] : ((e == 2) ? ss[2] : ss[3]));
    const double m = se / rows;
    double var = sse / rows - m * m;
    mean[c + e] = (float)m;
    invstd[c + e] = (float)(1.0 / sqrt(var + 1e-5));
}

// V1<W, U>: W waves per channel quad (one block per quad), every lane issues U row pairs before it adds anything (one memory
// round trip for chunks <= 64*W*U), DPP-free wave butterfly, W partial sums combined through LDS by lanes 0..3 of wave 0.
template <int W, int U>
__global__ __launch_bounds__(64 * W) void fin_v1(const float* __restrict__ part, int chunks, int C, double rows, float* mean, float* invstd) {
    __shared__ double red[W][8];
    const int c = blockIdx.x * 4;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    for (int k0 = t; k0 < chunks; k0 += 64 * W * U) {
        f32x4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u * 64 * W;
            const int kk = k < chunks ? k : chunks - 1;
            a[u] = *(const f32x4*)(part + (size_t)kk * 2 * C + c);
            b[u] = *(const f32x4*)(part + (size_t)kk * 2 * C + C + c);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = k0 + u * 64 * W < chunks;
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[e] += ok ? (double)a[u][e] : 0.0; ss[e] += ok ? (double)b[u][e] : 0.0; }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { s[e] = wave_sum_d(s[e]); ss[e] = wave_sum_d(ss[e]); }
    if (lane == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[wave][e] = s[e]; red[wave][4 + e] = ss[e]; }
    }
    __syncthreads();
    if (t >= 4) return;
    double se = 0, sse = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) { se += red[w][t]; sse += red[w][4 + t]; }
    const double m = se / rows;
    double var = sse / rows - m * m;
    mean[c + t] = (float)m;
    invstd[c + t] = (float)(1.0 / sqrt(var + 1e-5));
}

// V2<W, U>: rows reduced in fp32 pairs first?  No: same as V1 but the wave butterfly runs on TWO packed values per lane pair —
// lanes reduce 8 doubles with 3 exchange steps by halving the value set each step (transpose-reduce): 8+4+2+1 exchanges instead of 8*6.
template <int W, int U>
__global__ __launch_bounds__(64 * W) void fin_v2(const float* __restrict__ part, int chunks, int C, double rows, float* mean, float* invstd) {
    __shared__ double red[W][8];
    const int c = blockIdx.x * 4;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k0 = t; k0 < chunks; k0 += 64 * W * U) {
        f32x4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u * 64 * W;
            const int kk = k < chunks ? k : chunks - 1;
            a[u] = *(const f32x4*)(part + (size_t)kk * 2 * C + c);
            b[u] = *(const f32x4*)(part + (size_t)kk * 2 * C + C + c);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = k0 + u * 64 * W < chunks;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += ok ? (double)a[u][e] : 0.0; v[4 + e] += ok ? (double)b[u][e] : 0.0; }
        }
    }
    // transpose-reduce: after step j (xor 1, 2, 4) each lane keeps half of its values, summed with its partner's copy of them.
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
    // lane l now holds value index ((l&1)*4 + ((l>>1)&1)*2 + ((l>>2)&1)) summed over its 8-lane group; sum the 8 groups.
    w1 += __shfl_xor(w1, 8, 64);
    w1 += __shfl_xor(w1, 16, 64);
    w1 += __shfl_xor(w1, 32, 64);
    if (lane < 8) red[wave][(lane & 1) * 4 + ((lane >> 1) & 1) * 2 + ((lane >> 2) & 1)] = w1;
    __syncthreads();
    if (t >= 4) return;
    double se = 0, sse = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) { se += red[w][t]; sse += red[w][4 + t]; }
    const double m = se / rows;
    double var = sse / rows - m * m;
    mean[c + t] = (float)m;
    invstd[c + t] = (float)(1.0 / sqrt(var + 1e-5));
}

__global__ void empty_kernel(float* p) { if (p == nullptr) p[0] = 0; }

template <typename F>
static float chain_ms(F launch, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const int shapes[][2] = {{704, 64}, {704, 256}, {361, 128}, {181, 512}, {361, 512}, {91, 256}, {91, 1024}, {181, 128}, {1408, 64}};
    float *part, *mean, *invstd;
    hipMalloc(&part, (size_t)1408 * 2 * 2048 * 4);
    hipMalloc(&mean, 2048 * 4); hipMalloc(&invstd, 2048 * 4);
    std::vector<float> ref(2048), got(2048);
    const int R = 200;
    {
        const float e = chain_ms([&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0, mean); }, R);
        printf("empty kernel in a chain: %.2f us per launch\n", e * 1e3);
    }
    for (auto& sh : shapes) {
        const int chunks = sh[0], C = sh[1];
        const size_t n = (size_t)chunks * 2 * C;
        auto prod = [&] { hipLaunchKernelGGL(producer, dim3(512), dim3(256), 0, 0, part, n, 0.5f); };
        const float base = chain_ms(prod, R);
        auto run = [&](const char* name, auto fin) {
            hipMemset(mean, 0, C * 4);
            const float t = chain_ms([&] { prod(); fin(); }, R);
            hipMemcpy(got.data(), mean, C * 4, hipMemcpyDeviceToHost);
            double d = 0;
            for (int i = 0; i < C; ++i) d = fmax(d, fabs((double)got[i] - ref[i]));
            const float hot = chain_ms(fin, R);
            printf("  %-18s after producer %6.2f us   back-to-back (warm) %6.2f us   max|mean - v0| %.1e\n", name, (t - base) * 1e3, hot * 1e3, d);
        };
        printf("chunks=%d C=%d (%.2f MB of partial rows; producer alone %.2f us)\n", chunks, C, n * 4 / 1e6, base * 1e3);
        prod();
        hipLaunchKernelGGL(fin_v0, dim3((C + 15) / 16), dim3(256), 0, 0, part, chunks, C, 1e4, mean, invstd);
        hipMemcpy(ref.data(), mean, C * 4, hipMemcpyDeviceToHost);
        run("v0 (shipped)", [&] { hipLaunchKernelGGL(fin_v0, dim3((C + 15) / 16), dim3(256), 0, 0, part, chunks, C, 1e4, mean, invstd); });
        run("v1 W=4 U=3", [&] { hipLaunchKernelGGL((fin_v1<4, 3>), dim3(C / 4), dim3(256), 0, 0, part, chunks, C, 1e4, mean, invstd); });
        run("v1 W=8 U=2", [&] { hipLaunchKernelGGL((fin_v1<8, 2>), dim3(C / 4), dim3(512), 0, 0, part, chunks, C, 1e4, mean, invstd); });
        run("v1 W=16 U=1", [&] { hipLaunchKernelGGL((fin_v1<16, 1>), dim3(C / 4), dim3(1024), 0, 0, part, chunks, C, 1e4, mean, invstd); });
        run("v1 W=2 U=6", [&] { hipLaunchKernelGGL((fin_v1<2, 6>), dim3(C / 4), dim3(128), 0, 0, part, chunks, C, 1e4, mean, invstd); });
        run("v1 W=1 U=12", [&] { hipLaunchKernelGGL((fin_v1<1, 12>), dim3(C / 4), dim3(64), 0, 0, part, chunks, C, 1e4, mean, invstd); });
        run("v2 W=4 U=3", [&] { hipLaunchKernelGGL((fin_v2<4, 3>), dim3(C / 4), dim3(256), 0, 0, part, chunks, C, 1e4, mean, invstd); });
        run("v2 W=8 U=2", [&] { hipLaunchKernelGGL((fin_v2<8, 2>), dim3(C / 4), dim3(512), 0, 0, part, chunks, C, 1e4, mean, invstd); });
        run("v2 W=2 U=6", [&] { hipLaunchKernelGGL((fin_v2<2, 6>), dim3(C / 4), dim3(128), 0, 0, part, chunks, C, 1e4, mean, invstd); });
    }
    return 0;
}
