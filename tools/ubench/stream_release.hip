// Micro-benchmark (developer tool): when does the main stream's NEXT kernel start after it has released work to a side stream
// (event record + hipStreamWaitEvent), depending on what the side stream then runs?  Every kernel stamps its own start / end with
// the 100 MHz constant clock, so the timeline is the GPU's, not the host's.
// build: hipcc -O3 --offload-arch=gfx950 stream_release.hip -o stream_release
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <string.h>
#include <vector>

__global__ void spin(unsigned long long* stamp, int slot, long ticks) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) atomicMin(&stamp[2 * slot], t0);
    while ((long)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) atomicMax(&stamp[2 * slot + 1], wall_clock64());
}

// write-heavy side kernel: streams `bytes` of stores, `passes` times (dirty lines in every L2 while the main stream records its event)
__global__ void writer(unsigned long long* stamp, int slot, float* dst, size_t n4, int passes) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) atomicMin(&stamp[2 * slot], t0);
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int p = 0; p < passes; ++p)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) ((f4*)dst)[i] = f4{(float)p, 1, 2, 3};
    if (threadIdx.x == 0) atomicMax(&stamp[2 * slot + 1], wall_clock64());
}

static hipStream_t st[2] = {nullptr, nullptr};      // (created once: streams created later may share a hardware queue with these)

static void run_writer_case(const char* title, unsigned flags, int wgrid) {
    unsigned long long* stamp;
    float* big;
    hipMalloc(&stamp, 64 * 16);
    hipMalloc(&big, (size_t)512 << 20);
    hipEvent_t ev;
    hipEventCreateWithFlags(&ev, flags);
    std::vector<unsigned long long> h(128);
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 64; ++i) { h[2 * i] = ~0ull; h[2 * i + 1] = 0; }
        hipMemcpy(stamp, h.data(), 64 * 16, hipMemcpyHostToDevice);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(writer, dim3(wgrid), dim3(256), 0, st[1], stamp, 1, big, (size_t)(512 << 20) / 16, 4);       // side: busy writing
        hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st[0], stamp, 0, 100L * 100);                              // main: A
        hipEventRecord(ev, st[0]);
        hipStreamWaitEvent(st[1], ev, 0);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[1], stamp, 2, 20L * 100);                                  // side: S (behind W)
        hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st[0], stamp, 3, 50L * 100);                               // main: B
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), stamp, 64 * 16, hipMemcpyDeviceToHost);
    const unsigned long long t0 = h[0];
    const char* nm[4] = {"A main (spin)", "W side (writes 2 GB)", "S side (after release)", "B main (spin)"};
    printf("%s\n", title);
    for (int i = 0; i < 4; ++i) printf("   %-24s start %8.1f us  end %8.1f us\n", nm[i], ((long long)h[2 * i] - (long long)t0) / 100.0, ((long long)h[2 * i + 1] - (long long)t0) / 100.0);
    printf("   => B starts %.1f us after A ends\n", ((long long)h[6] - (long long)h[1]) / 100.0);
    hipFree(big); hipFree(stamp); hipEventDestroy(ev);
}

// release WITHOUT a marker packet: A carries the event as its own completion signal (hipExtLaunchKernelGGL stopEvent), the side
// stream waits for that; mode 0 = event record (marker), 1 = stop event on A, 2 = no release at all (reference)
static void run_stopevent_case(const char* title, int mode, int n_pairs) {
    unsigned long long* stamp;
    hipMalloc(&stamp, 64 * 16);
    hipEvent_t ev[16];
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    std::vector<unsigned long long> h(128);
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 64; ++i) { h[2 * i] = ~0ull; h[2 * i + 1] = 0; }
        hipMemcpy(stamp, h.data(), 64 * 16, hipMemcpyHostToDevice);
        hipDeviceSynchronize();
        for (int i = 0; i < n_pairs; ++i) {          // main: A_i (30 us) [release] ; side: S_i (10 us, one block)
            if (mode == 1)
                hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st[0], nullptr, ev[i], 0, stamp, 2 * i, 30L * 100);
            else
                hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st[0], stamp, 2 * i, 30L * 100);
            if (mode == 0) hipEventRecord(ev[i], st[0]);
            if (mode != 2) hipStreamWaitEvent(st[1], ev[i], 0);
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[1], stamp, 2 * i + 1, 10L * 100);
        }
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), stamp, 64 * 16, hipMemcpyDeviceToHost);
    printf("%s\n", title);
    double gap = 0, lat = 0;
    for (int i = 1; i < n_pairs; ++i) gap += ((long long)h[2 * (2 * i)] - (long long)h[2 * (2 * i - 2) + 1]) / 100.0;
    for (int i = 0; i < n_pairs; ++i) lat += ((long long)h[2 * (2 * i + 1)] - (long long)h[2 * (2 * i) + 1]) / 100.0;
    printf("   main stream: A_i end -> A_i+1 start %.2f us (mean of %d);  side stream: S_i starts %.2f us after A_i ends;  whole chain %.1f us\n",
           gap / (n_pairs - 1), n_pairs - 1, lat / n_pairs, ((long long)h[2 * (2 * n_pairs - 2) + 1] - (long long)h[0]) / 100.0);
    hipFree(stamp);
    for (auto& e : ev) hipEventDestroy(e);
}

// release through MEMORY: the last block of A writes a flag (atomic ticket), the side stream waits with hipStreamWaitValue32 — no
// packet of the main stream's queue takes part.  mode 0: flag in hipMallocSignalMemory, 1: flag in plain device memory
__global__ void spin_release(unsigned long long* stamp, int slot, long ticks, unsigned* ticket, unsigned* flag, unsigned seq) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) atomicMin(&stamp[2 * slot], t0);
    while ((long)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) atomicMax(&stamp[2 * slot + 1], wall_clock64());
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
            *ticket = 0;
            __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

static void run_waitvalue_case(const char* title, int mode, int n_pairs) {
    int can = 0;
    hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    unsigned long long* stamp;
    unsigned *ticket, *flag = nullptr;
    hipMalloc(&stamp, 64 * 16);
    hipMalloc(&ticket, 64);
    hipMemset(ticket, 0, 64);
    hipError_t e = mode == 0 ? hipExtMallocWithFlags((void**)&flag, 64, hipMallocSignalMemory) : hipMalloc(&flag, 64);
    printf("%s\n   (hipDeviceAttributeCanUseStreamWaitValue = %d, flag allocation: %s)\n", title, can, hipGetErrorString(e));
    if (e != hipSuccess || !can) return;
    hipMemset(flag, 0, 8);
    std::vector<unsigned long long> h(128);
    unsigned seq = 0;
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 64; ++i) { h[2 * i] = ~0ull; h[2 * i + 1] = 0; }
        hipMemcpy(stamp, h.data(), 64 * 16, hipMemcpyHostToDevice);
        hipDeviceSynchronize();
        for (int i = 0; i < n_pairs; ++i) {
            ++seq;
            hipLaunchKernelGGL(spin_release, dim3(256), dim3(256), 0, st[0], stamp, 2 * i, 30L * 100, ticket, flag, seq);
            hipError_t w = hipStreamWaitValue32(st[1], flag, seq, hipStreamWaitValueGte, 0xffffffffu);
            if (w != hipSuccess) { printf("   hipStreamWaitValue32: %s\n", hipGetErrorString(w)); return; }
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[1], stamp, 2 * i + 1, 10L * 100);
        }
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), stamp, 64 * 16, hipMemcpyDeviceToHost);
    double gap = 0, lat = 0;
    for (int i = 1; i < n_pairs; ++i) gap += ((long long)h[2 * (2 * i)] - (long long)h[2 * (2 * i - 2) + 1]) / 100.0;
    for (int i = 0; i < n_pairs; ++i) lat += ((long long)h[2 * (2 * i + 1)] - (long long)h[2 * (2 * i) + 1]) / 100.0;
    printf("   main stream: A_i end -> A_i+1 start %.2f us (mean of %d);  side stream: S_i starts %.2f us after A_i ends;  whole chain %.1f us\n",
           gap / (n_pairs - 1), n_pairs - 1, lat / n_pairs, ((long long)h[2 * (2 * n_pairs - 2) + 1] - (long long)h[0]) / 100.0);
}

struct K { const char* name; int stream; int grid, block; long us; bool memset_before; };

static void run_case(const char* title, const std::vector<K>& ks, int release_after, bool release_at_all) {
    static unsigned long long* stamp = nullptr;
    static float* scratch = nullptr;
    static hipEvent_t ev;
    if (!st[0]) {
        hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking);
        hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking);
        hipMalloc(&stamp, 64 * 16);
        hipMalloc(&scratch, 64 << 20);
        hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    }
    std::vector<unsigned long long> h(128);
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 64; ++i) { h[2 * i] = ~0ull; h[2 * i + 1] = 0; }
        hipMemcpy(stamp, h.data(), 64 * 16, hipMemcpyHostToDevice);
        hipDeviceSynchronize();
        for (size_t i = 0; i < ks.size(); ++i) {
            const K& k = ks[i];
            if (k.memset_before) hipMemsetAsync(scratch, 0, 4 << 20, st[k.stream]);
            hipLaunchKernelGGL(spin, dim3(k.grid), dim3(k.block), 0, st[k.stream], stamp, (int)i, k.us * 100);
            if ((int)i == release_after && release_at_all) {
                hipEventRecord(ev, st[0]);
                hipStreamWaitEvent(st[1], ev, 0);
            }
        }
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), stamp, 64 * 16, hipMemcpyDeviceToHost);
    printf("%s\n", title);
    const unsigned long long t0 = h[0];
    for (size_t i = 0; i < ks.size(); ++i)
        printf("   %-22s stream %d grid %5d x %4d : start %8.1f us  end %8.1f us\n", ks[i].name, ks[i].stream, ks[i].grid, ks[i].block,
               (h[2 * i] - t0) / 100.0, (h[2 * i + 1] - t0) / 100.0);
}

int main() {
    // main: A (100 us) -> release -> B (50 us);  side: S after the release
    run_case("1. side runs ONE small block for 200 us",
             {{"A main", 0, 256, 256, 100, false}, {"S side", 1, 1, 64, 200, false}, {"B main", 0, 256, 256, 50, false}}, 0, true);
    run_case("2. side runs 2048 blocks x 1024 threads (fills every CU) for 200 us each",
             {{"A main", 0, 256, 256, 100, false}, {"S side", 1, 2048, 1024, 200, false}, {"B main", 0, 256, 256, 50, false}}, 0, true);
    run_case("3. side runs a 4 MB hipMemsetAsync, then one small block for 200 us",
             {{"A main", 0, 256, 256, 100, false}, {"S side", 1, 1, 64, 200, true}, {"B main", 0, 256, 256, 50, false}}, 0, true);
    run_case("4. as 1, ten small side kernels of 20 us",
             {{"A main", 0, 256, 256, 100, false}, {"S0", 1, 1, 64, 20, false}, {"S1", 1, 1, 64, 20, false}, {"S2", 1, 1, 64, 20, false},
              {"S3", 1, 1, 64, 20, false}, {"S4", 1, 1, 64, 20, false}, {"S5", 1, 1, 64, 20, false}, {"S6", 1, 1, 64, 20, false},
              {"S7", 1, 1, 64, 20, false}, {"S8", 1, 1, 64, 20, false}, {"S9", 1, 1, 64, 20, false}, {"B main", 0, 256, 256, 50, false}}, 0, true);
    run_case("5. as 3 with ten memset + kernel pairs on the side stream",
             {{"A main", 0, 256, 256, 100, false}, {"S0", 1, 1, 64, 20, true}, {"S1", 1, 1, 64, 20, true}, {"S2", 1, 1, 64, 20, true},
              {"S3", 1, 1, 64, 20, true}, {"S4", 1, 1, 64, 20, true}, {"S5", 1, 1, 64, 20, true}, {"S6", 1, 1, 64, 20, true},
              {"S7", 1, 1, 64, 20, true}, {"S8", 1, 1, 64, 20, true}, {"S9", 1, 1, 64, 20, true}, {"B main", 0, 256, 256, 50, false}}, 0, true);
    run_case("6. no release at all (independent streams), side: one small block 200 us",
             {{"A main", 0, 256, 256, 100, false}, {"S side", 1, 1, 64, 200, false}, {"B main", 0, 256, 256, 50, false}}, 0, false);
    run_case("7. B is a 1024-thread kernel, side: 512 blocks x 256 threads for 200 us",
             {{"A main", 0, 256, 256, 100, false}, {"S side", 1, 512, 256, 200, false}, {"B main", 0, 256, 1024, 50, false}}, 0, true);
    run_stopevent_case("13. 12 x (main A 30 us -> release -> side S 10 us): release = hipEventRecord + hipStreamWaitEvent", 0, 12);
    run_stopevent_case("14. same, release = A's own completion (hipExtLaunchKernelGGL stopEvent) + hipStreamWaitEvent", 1, 12);
    run_stopevent_case("15. same, no release (independent streams)", 2, 12);
    run_waitvalue_case("16. release through memory: last block of A writes a flag in SIGNAL memory, side stream hipStreamWaitValue32", 0, 12);
    run_waitvalue_case("17. same, flag in plain device memory", 1, 12);
    run_writer_case("8. side stream is WRITING (2 GB in flight) while main records the event; event flags: DisableTiming", hipEventDisableTiming, 2048);
    run_writer_case("9. same, event flags: DisableTiming | ReleaseToDevice", hipEventDisableTiming | hipEventReleaseToDevice, 2048);
    run_writer_case("10. same, event flags: DisableTiming | DisableSystemFence", hipEventDisableTiming | hipEventDisableSystemFence, 2048);
    run_writer_case("11. as 8 with the writer on 64 blocks only", hipEventDisableTiming, 64);
    run_writer_case("12. as 9 with the writer on 64 blocks only", hipEventDisableTiming | hipEventReleaseToDevice, 64);
    return 0;
}
