// any_order.hip — does hipExtAnyOrderLaunch (no barrier bit on the dispatch packet) let two INDEPENDENT kernels of one stream overlap on
// gfx950?  (hip_ext.h says the flag is not supported on GFX9xx boards; measured rather than believed.)
// Four launches of a small-grid kernel that spins ~100 us: in order they take ~4 x 100 us, overlapped ~100 us.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
__global__ void spin(long long cycles, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = 1;
}
int main() {
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    int* sink;
    hipMalloc(&sink, 4);
    const long long cyc = 10000;      // 100 MHz wall clock: 100 us
    for (int flag = 0; flag < 2; ++flag)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a, st);
            for (int i = 0; i < 4; ++i) hipExtLaunchKernelGGL(spin, dim3(8), dim3(64), 0, st, nullptr, nullptr, flag ? hipExtAnyOrderLaunch : 0, cyc, sink);
            hipEventRecord(b, st);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            printf("flags=%d  4 launches of a ~100 us 8-block kernel: %.1f us\n", flag, ms * 1e3f);
        }
    return 0;
}
