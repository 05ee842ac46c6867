// common.h — shared host/device helpers for libzsg (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "zsg.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// BatchNorm-backward statistics fused into the epilogue of the data-gradient convolution that completes dout (the gradient
// w.r.t. the BatchNorm's ReLU output): per output tile, per channel, (sum g, sum g * xhat) with g = dout * relu-bit and
// xhat = (x - mean) * invstd — the pass zsg_bn_backward otherwise makes over dout and x.  x / mask are indexed with the
// OUTPUT element offsets of the convolution (the same dense [rows][C] layout as dout).
struct BnbDev {
    const float* x;               // BatchNorm input (the forward convolution's output); nullptr = no fusion
    const float* mean;
    const float* invstd;
    const unsigned char* mask;    // 4 ReLU bits per 16-byte group (bn_apply's relu_mask) or nullptr
    int store_masked;             // zsg_conv_desc.epi_flags bit 0: the STORED value is the masked gradient g (round 6)
};

// pw.hip: the filter-resident streaming kernel behind zsg_conv_igemm's tile_hint BM = 32 (uw = the hint's BN field)
int zsg_conv_pw_launch(const zsg_conv_desc* d, int uw, const float* src, const float* wt, float* out, const float* bias, const float* add_src,
                       const float* mask_src, float* bn_partials, const BnbDev* bnb, hipStream_t st);

// mx.hip: the filter-resident streaming kernel of the network's first convolution (merge_x descriptors) behind tile_hint BM = 32
bool zsg_conv_mx_ok(const zsg_conv_desc* d, const char** why);
int zsg_conv_mx_groups(const zsg_conv_desc* d);
int zsg_conv_mx_launch(const zsg_conv_desc* d, const float* src, const float* wt, float* out, const float* bias, const float* add_src,
                       const float* mask_src, float* bn_partials, hipStream_t st);

// Wave priority of the kernels of the step's dependent (main-stream) chain: convolutions, BatchNorm passes.  The weight-gradient kernels
// (side stream) stay at 0, so where a CU holds blocks of both streams the SIMD's arbiter serves the critical chain first.
// Round 6: a RUN-TIME switch (zsg_set_main_priority; default ZSG_MAIN_PRIO = 3).  s_setprio takes an immediate, so the kernels branch on a
// flag word: one __constant__ int per translation unit (no relocatable device code: every .hip file is its own code object), set by a
// per-file setter that registers itself with api.cpp.  The read is one scalar load next to the kernel-argument loads.
#ifndef ZSG_MAIN_PRIO
#define ZSG_MAIN_PRIO 3
#endif
typedef int (*zsg_prio_setter_t)(int);
void zsg_register_prio_setter(zsg_prio_setter_t fn);
#define ZSG_DEFINE_PRIO_FLAG()                                                                                                      \
    __constant__ int zsg_prio_flag_c = ZSG_MAIN_PRIO;                                                                               \
    static int zsg_prio_set_(int v) {                                                                                               \
        return hipMemcpyToSymbol(HIP_SYMBOL(zsg_prio_flag_c), &v, sizeof(int), 0, hipMemcpyHostToDevice) == hipSuccess ? 0 : -3;      \
    }                                                                                                                               \
    namespace {                                                                                                                     \
    struct ZsgPrioReg {                                                                                                             \
        ZsgPrioReg() { zsg_register_prio_setter(zsg_prio_set_); }                                                                   \
    } zsg_prio_reg_;                                                                                                                \
    }
#define ZSG_SET_MAIN_PRIO() do { if (zsg_prio_flag_c) __builtin_amdgcn_s_setprio(3); } while (0)

#define ZSG_WAVE 64
#define ZSG_NUM_CU 256
#define ZSG_NUM_XCD 8
#define ZSG_MAX_DEV 64      // per-device caches of kernel attributes (one process may drive several GPUs)

// ---- error plumbing -------------------------------------------------------------------------------------------
void zsg_set_error(const char* fmt, ...);
#define ZSG_FAIL(code, ...)          \
    do {                             \
        zsg_set_error(__VA_ARGS__);  \
        return (code);               \
    } while (0)
#define ZSG_REQUIRE(cond, ...)                   \
    do {                                         \
        if (!(cond)) ZSG_FAIL(-1, __VA_ARGS__);  \
    } while (0)
#define ZSG_CHECK_LAUNCH(name)                                                            \
    do {                                                                                  \
        hipError_t e__ = hipGetLastError();                                               \
        if (e__ != hipSuccess) ZSG_FAIL(-3, "%s: launch failed: %s", name, hipGetErrorString(e__)); \
    } while (0)

// ---- kernel launches ------------------------------------------------------------------------------------------
// Every kernel launch of libzsg goes through ZSG_LAUNCH.  When the calling thread has armed a completion event
// (zsg_set_completion_event), the launch carries it as the dispatch packet's own completion signal (hipExtLaunchKernelGGL's stop
// event): another stream can then wait for this launch with zsg_stream_wait_event WITHOUT an event-record marker packet in this
// stream's queue (measured, tools/ubench/stream_release.hip: the next kernel of the releasing stream starts 3.4 us after this one
// ends instead of 7.7 us behind a marker).  A call that launches several kernels re-arms the same event on each: the last wins.
extern thread_local hipEvent_t zsg_tls_completion_event;
extern thread_local int zsg_tls_completion_uses;
#define ZSG_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                                   \
    do {                                                                                                                      \
        if (zsg_tls_completion_event) {                                                                                       \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, nullptr, zsg_tls_completion_event, 0, __VA_ARGS__);      \
            ++zsg_tls_completion_uses;                                                                                        \
        } else                                                                                                                  \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                              \
    } while (0)

// ---- per-stream scratch (api.cpp; zsg_set_stream_workspace) ----------------------------------------------------------------
// The caller registers one scratch buffer per stream; launches of one stream are ordered, so they can share it.  The first
// ZSG_SK_FLAG_BYTES are hand-off flags (zeroed at registration, left zero by every launch), the rest partial-tile slots.
#define ZSG_SK_FLAG_BYTES 16384
#define ZSG_SK_ERR_WORD (ZSG_SK_FLAG_BYTES / 4 - 1)      // the last flag word: set when a stream-K finisher gave up waiting (bounded poll)
void* zsg_stream_workspace(hipStream_t st, size_t* bytes);

// ---- per-launch profiling (prof.cpp) --------------------------------------------------------------------------
struct ZsgProfScope {
    int slot;
    hipStream_t st;
    ZsgProfScope(const char* name, hipStream_t s, double flops, double bytes);
    ~ZsgProfScope();
};
extern int g_zsg_prof_on;
extern int g_zsg_deterministic;     // zsg_set_deterministic: reductions never combine partial sums with fp32 atomics
#define ZSG_PROF(name, stream, flops, bytes) ZsgProfScope prof__(name, (hipStream_t)(stream), (flops), (bytes))

// ALGORITHMIC HBM bytes of one convolution launch: every operand element and every output element once (input pixels x channels,
// the filter, the output; + the output again when an add_src / accumulate operand is read).  What bench.py divides the measured
// FETCH / WRITE traffic of a launch by (roofline.traffic_over_algorithmic).
static inline double zsg_conv_alg_bytes(const zsg_conv_desc* d, bool reads_out) {
    double e = (double)d->N * d->seg[0].ty.n * d->seg[0].tx.n * d->C;
    for (int s = 0; s < d->nseg; ++s) {
        const zsg_seg& a = d->seg[s];
        e += (double)d->B * a.src_H * a.src_W * d->C + (double)d->B * a.rows_y * a.rows_x * d->N * (reads_out ? 2.0 : 1.0);
    }
    return 4.0 * e;
}

// ---- device helpers -------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Raw buffer loads: the descriptor bounds-checks every lane (offset >= 2 GB window -> zeros, no fault), which is how
// the gather kernels implement zero padding / tails without branches around their loads (guide T8/T20).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x80000000u, 0x00020000);
}
__device__ __forceinline__ rsrc_t make_rsrc_n(const void* base, unsigned bytes) {     // a window of `bytes` bytes: lanes beyond it read zeros
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(rsrc_t r, unsigned byte_off) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ float buf_load1(rsrc_t r, unsigned byte_off) {
    unsigned v = __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0);
    return __builtin_bit_cast(float, v);
}
#define ZSG_OOB 0xFFFFFFF0u

// XCD-aware, bijective remap of a linear block id: blocks that land on one XCD (id % 8) get a contiguous chunk of
// the logical grid so that neighbouring tiles share that XCD's L2 (guide T1).
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int q = nblocks / ZSG_NUM_XCD, r = nblocks % ZSG_NUM_XCD;
    const int xcd = bid % ZSG_NUM_XCD, idx = bid / ZSG_NUM_XCD;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
