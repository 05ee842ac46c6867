// bn_tail.h — BatchNorm statistics finalised INSIDE the producing convolution, by the workgroup of each column block's last row tile (round 5).
//
// A convolution that feeds a train-mode BatchNorm writes per-tile partial rows [m_tiles][2][N] (sum, sum of squares — or, for the data
// gradient that completes a BatchNorm's dout, sum g and sum g * xhat).  Until round 4 a separate finalize launch (or, for few rows,
// every block of the apply pass) reduced them: 81 dependent ~5 us launches per step on the conv -> BatchNorm -> conv chain, or a
// prologue that re-read rows x C x 8 bytes in EVERY apply block (measured: 11.8 us against 6.9 us for the plain apply of a 5776 x 256
// tensor).  Here the tiles of one column block announce their partial row on a counter; the workgroup of the column block's last row
// tile waits for them, reduces the rows of that column block in a FIXED order in fp64 (deterministic, whatever the arrival order) and
// publishes mean / invstd / running statistics (forward) or the backward coefficients, d(gamma), d(beta).  The apply pass that follows
// is the plain one.
//
// Protocol (MI355X_MICROARCH.md, inter-workgroup visibility; cdna_hip_programming.md Guideline 16 in its counter form).  The first
// version of this round had EVERY tile take a returning ticket and the last arriver reduce: correct, but every workgroup then stood
// ~2-3 us behind its drain + the atomic's round trip before it could free its CU slot — on multi-round grids (736 tiles for a
// 5776 x 1024 output) that cost 15 us per launch, more than the finalize launch it replaced (profiles/r05_bn_tail_ab.txt).  Now:
//   producers (every tile but the column block's LAST row tile): partial row with write-through (sc1) stores; at the very end of the
//             workgroup each storing wave `s_waitcnt vmcnt(0)` (its stores have long been issued) and one lane fires a NON-returning
//             relaxed agent-scope add on the column block's counter — nobody waits for anything.
//   reducer  (the workgroup of the column block's last row tile — dispatched among the last of the grid): finishes its own tile, then
//             polls the counter (relaxed sc1 loads + s_sleep) until every other storing wave has arrived, reads the rows with sc1
//             loads (they bypass this CU's L1; the rows never sat dirty in another XCD's L2), reduces, publishes, and stores 0 into the
//             counter (zero again when the kernel ends: no memset per call).
// The producers do not depend on the reducer, so the poll cannot deadlock: it holds one CU slot while the rest of the grid drains
// (every producer only has to be DISPATCHED; a grid larger than the resident capacity drains in dispatch order — the reducer is the column
// block's LAST row tile, dispatched among the last).  Visibility rests on the pairing the guide prescribes for payloads of this size
// (MI355X_MICROARCH.md, inter-workgroup visibility: `sc1` write-through stores AND `sc1` loads on both sides, the producer's stores
// drained with s_waitcnt vmcnt(0) before its arrival) — not on fences, which write back / invalidate a whole XCD L2 (5-30x slower,
// DESIGN.md section 8 round 2).  The poll is bounded (below).
// The output stores of a tile are never waited for on their own account (nobody reads them in this launch).
//
// Results do not depend on dispatch order or XCD placement; counters must be zero at entry (the host allocates them zeroed, one word
// per column block and per use site) and are zero at exit.
#pragma once
#include "common.h"

struct BnTail {
    unsigned* tickets;        // arrival counters, [column blocks of the launch]; nullptr = no in-kernel finalize
    // mode 0: forward statistics
    float* mean;
    float* invstd;
    float* rmean;             // running statistics or nullptr
    float* rvar;
    // mode 1: backward sums -> coef[0][c] = sum g / n, coef[1][c] = sum g * xhat / n; dgamma / dbeta written or accumulated
    float* coef;
    float* dgamma;
    float* dbeta;
    float momentum, eps;
    int accumulate;
    int mode;
    long long rows;           // pixels the statistics are over (n)
};

#define BN_TAIL_MAX_ROWS 128      // a column block's partial rows one workgroup reduces in one memory round trip (<= 64 KB at 64 columns)

typedef unsigned __attribute__((address_space(1))) bn_gu32;

__device__ __forceinline__ void bn_tail_store(float* p, float v) {      // 4-byte write-through store (the epilogue's natural width)
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Producer side, at the very end of the workgroup, executed by the waves that stored the partial row (whole waves): drain, then one
// lane per wave adds 1 (non-returning).  The reducer expects (nrows - 1) * storing_waves arrivals.
__device__ __forceinline__ void bn_tail_arrive(unsigned* counter) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) (void)__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Reducer side: all threads of the workgroup that owns the column block's LAST row tile, uniform control flow, after its own partial row
// has been stored (sc1) and its output stores issued.  `sh` = LDS scratch of >= (NT / (BN / 2)) * 2 * BN doubles (may alias anything the
// block no longer needs).  part: [nrows][2][N] partial rows; this block's columns are [n0, n0 + BN).
template <int NT, int BN>
__device__ __forceinline__ void bn_tail_reduce(const BnTail& t, unsigned* counter, const float* part, int nrows, int N, int n0, double* sh) {
    constexpr int CG = BN / 4;            // 16-byte column groups
    constexpr int RL = NT / (2 * CG);     // row lanes (each (kind, column group) is read by RL threads)
    constexpr int SW = BN / 64 > 0 ? BN / 64 : 1;      // storing waves per producer workgroup
    static_assert(RL >= 1 && NT % (2 * CG) == 0, "thread count vs column block");
    const int tid = threadIdx.x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this workgroup's own partial row (every storing wave drains)
    __syncthreads();
    // Bounded poll (~1-2 s): a lost arrival — an aborted launch that left the word nonzero, a producer workgroup that never
    // became resident — must not hang the training step.  The reducer then gives up, re-zeroes the counter and POISONS what it
    // publishes (NaN statistics / coefficients): the loss turns NaN at once and the trainer's NaN handling names the step, instead of a
    // silent wait.  sh[0] carries the verdict to the other threads.
    if (tid == 0) {
        const unsigned expect = (unsigned)(nrows - 1) * SW;
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != expect && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(8);
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // zero again for the next launch
        *(volatile int*)sh = spins >= (1 << 20) ? 1 : 0;
    }
    __syncthreads();
    const bool poisoned = *(volatile int*)sh != 0;
    __syncthreads();                      // (sh is reused below)
    const int cg = tid % CG, kind = (tid / CG) & 1, rl = tid / (2 * CG);
    const int n = n0 + 4 * cg;
    double v[4] = {0, 0, 0, 0};
    {
        // sc1 loads (aux 16): served by L2 / memory, never by this CU's L1 — the rows were written by other CUs during this launch
        const rsrc_t rs = make_rsrc(part);
        constexpr int U = 16;
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        for (int k0 = rl; k0 < nrows; k0 += U * RL) {
            f32x4 a[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + u * RL;
                const bool ok = (k < nrows) & (n < N);
                const unsigned off = ok ? 4u * (unsigned)((k * 2 + kind) * N + n) : ZSG_OOB;
                a[u] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 16));
            }
#pragma unroll
            for (int u = 0; u < U; ++u)           // (rows past the end were read as zeros: adding +0.0 keeps the order of the sum fixed)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (double)a[u][e];
        }
    }
    // row lanes are combined through LDS in row-lane order (fixed)
#pragma unroll
    for (int e = 0; e < 4; ++e) sh[(rl * 2 + kind) * BN + 4 * cg + e] = v[e];
    __syncthreads();
    if (tid < BN && n0 + tid < N) {
        double s = 0, ss = 0;
#pragma unroll 4
        for (int r = 0; r < RL; ++r) {
            s += sh[(r * 2 + 0) * BN + tid];
            ss += sh[(r * 2 + 1) * BN + tid];
        }
        const int c = n0 + tid;
        const double cnt = (double)t.rows;
        if (poisoned) s = ss = __builtin_nan("");
        if (t.mode == 0) {
            const double m = s / cnt;
            double var = ss / cnt - m * m;
            if (var < 0) var = 0;
            t.mean[c] = (float)m;
            t.invstd[c] = (float)(1.0 / sqrt(var + (double)t.eps));
            if (t.rmean) t.rmean[c] = (1.f - t.momentum) * t.rmean[c] + t.momentum * (float)m;
            if (t.rvar) t.rvar[c] = (1.f - t.momentum) * t.rvar[c] + t.momentum * (float)(cnt > 1 ? var * cnt / (cnt - 1) : var);
        } else {
            t.coef[c] = (float)(s / cnt);
            t.coef[N + c] = (float)(ss / cnt);
            if (t.dbeta) t.dbeta[c] = (t.accumulate ? t.dbeta[c] : 0.f) + (float)s;
            if (t.dgamma) t.dgamma[c] = (t.accumulate ? t.dgamma[c] : 0.f) + (float)ss;
        }
    }
}
