// pw.hip — filter-resident streaming kernel for the 1x1 / stride-1 convolutions whose filter (whole, or cut into 2 / 4 / 8
// panels of output channels) fits a CU's LDS: the bottleneck 1x1 layers of the trunk's first stages (fpn_resnet.py:66-72,86-100:
// layer1's conv1 64/256 -> 64, conv3 64 -> 256 and projection shortcut 64 -> 256 at M = B x 75 x 75 pixels; layer2's conv3
// 128 -> 512 at M = B x 38 x 38), forward and data gradient, fp32 MFMA, gfx950.  Reached through zsg_conv_igemm /
// zsg_conv_igemm_bnb with tile_hint BM = 32 (BN = unit width); the host autotuner times it next to the implicit-GEMM tiles.
//
// Why a kernel of its own.  As tiles of the implicit GEMM (igemm.hip) these launches run at HALF their roofline (DESIGN 8,
// profiles/r03_shortk_gemm.txt: 64 -> 256 takes 47-51 us for 18.7 us of MFMA work and ~21 us of HBM traffic): a block lives for
// two K steps, so its prologue, two latency-exposed tile loads, three barriers and the LDS-transposed epilogue are paid per 64
// MFMAs per wave and nothing overlaps them.  Here a workgroup is PERSISTENT (one per CU) and its eight waves are AUTONOMOUS:
//   * the filter panel [NB][K] (<= 67 KB) is parked in LDS once per workgroup;
//   * a wave walks "units" of 32 pixel rows x (32 NJ) output channels.  It stages its own 32 x 64 slice of the source through a
//     wave-private LDS buffer (coalesced 16-byte loads -> ds_write_b128 -> ds_read_b128 fragments: the LDS queue of a wave is
//     in-order, so no barrier — not even a workgroup one — separates its writes from its reads), multiplies it against the
//     resident panel, transposes the accumulators through the same buffer and stores whole 256-byte row runs;
//   * the loads of the next slice(s) are in flight during the MFMAs, and no barrier ties the waves together.
// The MFMA operands are swapped against igemm.hip (filter rows first): the 32x32 accumulator of lane (i, h) then holds, for
// PIXEL i, the output channels 8 q + 4 h .. + 3 (q = 0..3) — four consecutive channels per register quad, i.e. the transposition
// writes 16-byte units.  The K sum of an output runs in the same order as in the 64x64 tile: results are bit-identical to it.
//
// Epilogue terms, BatchNorm-statistics partials and BatchNorm-backward partials are those of igemm.hip's vectorised epilogue,
// with ONE partial row per workgroup row group (a wave accumulates over all its units, the workgroup reduces its waves in a fixed
// order: deterministic): zsg_conv_igemm_partial_rows() tells the caller how many rows a launch writes.
//
// Measured (profiles/r03_pw_microbench.txt, _ablation.txt, _pmc.txt, _ab.txt): 64 -> 256 @ M = 90000 51.6 -> 39.7 us per launch
// (MFMA pipe busy 0.365 -> 0.445), 128 -> 512 @ M = 23104 39.5 -> 35.2; the input-dominated 256 -> 64 only ties (39.9 vs 39.8);
// the training step 14.01 -> 13.91 ms.  What is left (ablation): ~14 us of launch latency + panel prologue + loop skeleton, and an
// MFMA phase (19.3 us) that ADDS to the 7.6 us of loads + stores of the SIMD's other wave instead of hiding them.
#include "common.h"

#define PW_TB 68          // floats per row of a wave's tile buffer: 64 + 4 = 17 x 16 B (odd: conflict-free b128 rows)
#define PW_WAVES 8
#define PW_LDS_MAX (160 * 1024)
#include <stdlib.h>

ZSG_DEFINE_PRIO_FLAG()

struct PwParams {
    const float* src;
    const float* wt;
    float* out;
    const float* bias;
    const float* add_src;
    const float* mask_src;
    float* stats;         // [row groups][2][N] partial rows (BatchNorm statistics, or BatchNorm-backward sums when bnb.x)
    BnbDev bnb;
    int M, K, N;
    int src_ld, out_ld, wt_ld, wc0;
    int src_off, out_off;
    int relu;
    int NS;               // column splits of a workgroup's panel: NB = NS x 32 NJ
    int NCB, NB;          // column blocks: the filter is cut into NCB panels of NB = N / NCB rows, one per workgroup
    int rt;               // row tiles: ceil(M / 32)
    int ns_shift;         // log2(NS)
};

// LDS traffic of ONE wave is processed in issue order, so a ds_read behind a ds_write of the same wave sees the data whichever
// lane wrote it: all that is needed between the two is that the COMPILER keeps the order (wavefront-scope fences emit no
// instruction on gfx950).
__device__ __forceinline__ void pw_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#define PW_MFMA(c, a, b) (c) = __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define PW_FRAG(ptr) (*(const f32x4*)(ptr))

__device__ __forceinline__ void buf_store4(rsrc_t r, unsigned byte_off, f32x4 v) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)byte_off, 0, 0);
}

// MODE 0: bias / ReLU / BatchNorm-statistics partials;  1: + add_src and / or mask_src;  2: BatchNorm-backward partials (+ add_src)
//
// Software pipeline of a wave, in "steps" of (unit, 64-deep K chunk):
//     Tb holds step s (parked);  register stage(s) hold steps s+1 (.. s+PF), requested from memory PF steps ahead.
//     step s:  fragments + MFMAs on Tb  ->  [last chunk of the unit: epilogue]  ->  park step s+1 in Tb  ->  request step s+1+PF.
// The park of the NEXT unit's first chunk sits inside the last epilogue pass, between that pass's LDS reads and its global
// stores: in the wave's memory queue the loads it waits for are then older than every store still in flight, so the counted
// wait does not include the stores' acknowledgements (a wave that had to drain its own stores before it could start the next
// unit lost ~1 us per unit).  PF = 2 register stages where a step is short (NJ <= 2: 32 / 64 MFMAs), 1 for NJ = 4 (128 MFMAs).
template <int NJ, int MODE>
__global__ __launch_bounds__(64 * PW_WAVES) void pw_kernel(const PwParams p) {
    ZSG_SET_MAIN_PRIO();
    constexpr int UW = 32 * NJ;             // unit width (output channels)
    constexpr int NP = (NJ + 1) / 2;        // epilogue passes of 64 columns
    constexpr int PF = NJ <= 2 ? 2 : 1;     // register stages
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = p.K, LDW = K + 4;         // filter row pitch: K/4 + 1 16-byte units (odd for K = 64, 128, 256 ...)
    float* Ws = smem;                                        // [NB][LDW]
    float* Tb = smem + p.NB * LDW + wave * (32 * PW_TB);     // this wave's [32][PW_TB]
    const int li = lane & 31, lh = lane >> 5;                // MFMA fragment coordinates
    const int cg = lane & 15, rr = lane >> 4;                // staging / epilogue coordinates: 16-byte column group, row class
    // Units: row tile rt = rgid + RG * j belongs to this workgroup's row group (j = 0, 1, ..), its NS column splits are the
    // workgroup's local units l = j * NS + ns, and wave w takes l = w, w + 8, ..: every workgroup gets the same number of row tiles
    // (+-1), a SIMD (waves s and s + 4) the same number of units (+-1), a wave keeps its column split (8 % NS == 0), and both
    // splits of a row tile read the source rows from the same CU's L1 / the same XCD's L2.
    // Column blocks (NCB > 1: the filter does not fit as a whole): gridDim.x = RG x NCB, the NCB workgroups of a row group walk the
    // same row tiles, each against its own panel of NB filter rows — on the same XCD where the grid allows (blockIdx % 8 picks the
    // XCD), so that the source rows are fetched into one L2.
    const int RT = p.rt, nsh = p.ns_shift, NCB = p.NCB, NB = p.NB;
    const int RG = (int)gridDim.x / NCB;
    int cb, rgid;
    if ((int)gridDim.x % (ZSG_NUM_XCD * NCB) == 0) {
        const int k = (int)blockIdx.x / ZSG_NUM_XCD;
        cb = k % NCB;
        rgid = (int)blockIdx.x % ZSG_NUM_XCD + ZSG_NUM_XCD * (k / NCB);
    } else {
        cb = (int)blockIdx.x % NCB;
        rgid = (int)blockIdx.x / NCB;
    }
    auto tile_of = [=](int l) { return rgid + RG * (l >> nsh); };
    const int NS = p.NS;
    const int n0l = (wave % NS) * UW;                        // this wave's columns: within the panel,
    const int n0 = cb * NB + n0l;                            // and of the output
    const int nkc = K >> 6;
    const int M = p.M;
    const rsrc_t rs = make_rsrc(p.src);

    // this wave's bias columns (MODE 2 fetches mean / invstd per pass instead: registers).  Requested FIRST: the filter panel's
    // counted waits below then cover them, and the streaming loop never waits for them again.
    f32x4 cv0[NP];
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
        const int cl = jp * 64 + 4 * cg;
        cv0[jp] = buf_load4(make_rsrc(p.bias ? p.bias : p.src), (MODE != 2 && p.bias && cl < UW) ? 4u * (unsigned)(n0 + cl) : ZSG_OOB);
    }
    f32x4 R[PF][8];
    int req_u = wave, req_k = 0;              // the next step to request (plain locals: advanced by PW_ADVANCE, never captured)
    // (past the last unit: out-of-range offsets, zeros, no memory traffic)
    auto request = [rs, rr, cg, M, RT, tile_of, &p](f32x4 (&r)[8], int ru, int rk) {
        const int rt = tile_of(ru);
        const int m0 = rt * 32;
        const bool live = rt < RT;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + rr + 4 * i;
            const bool ok = live & (m < M);
            r[i] = buf_load4(rs, ok ? 4u * (unsigned)(p.src_off + m * p.src_ld + rk * 64 + 4 * cg) : ZSG_OOB);
        }
    };
#define PW_ADVANCE()           \
    if (++req_k == nkc) {      \
        req_k = 0;             \
        req_u += PW_WAVES;     \
    }
    auto park = [Tb, rr, cg](const f32x4 (&r)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *(f32x4*)(Tb + (rr + 4 * i) * PW_TB + 4 * cg) = r[i];
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) {
        request(R[s], req_u, req_k);
        PW_ADVANCE();
    }

    {   // the filter panel, once per workgroup: eight 16-byte loads in flight per thread
        const rsrc_t rw = make_rsrc(p.wt);
        const int kg = K >> 2, total = NB * kg;
        for (int base = 0; base < total; base += 8 * 64 * PW_WAVES) {
            f32x4 t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = base + tid + j * 64 * PW_WAVES;
                const int n = idx / kg, g = idx - n * kg;
                t[j] = buf_load4(rw, idx < total ? 4u * (unsigned)((cb * NB + n) * p.wt_ld + p.wc0 + 4 * g) : ZSG_OOB);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = base + tid + j * 64 * PW_WAVES;
                const int n = idx / kg, g = idx - n * kg;
                if (idx < total) *(f32x4*)(Ws + n * LDW + 4 * g) = t[j];
            }
        }
    }
    park(R[0]);
    request(R[0], req_u, req_k);
    PW_ADVANCE();
    __syncthreads();
    // (The two waves of a SIMD are NOT decoupled by a start offset: while one streams fp32 MFMAs the other issues nothing at all on
    // gfx950, tools/ubench/mfma_coissue.hip — offsets of 2 000 - 8 000 cycles measured neutral to slower, profiles/r03_pw_stagger.txt.)

    const bool has_add = p.add_src != nullptr, has_mask = p.mask_src != nullptr, has_bits = p.bnb.mask != nullptr;
    const rsrc_t rs_out = make_rsrc(p.out);
    const rsrc_t rs_add = make_rsrc(has_add ? p.add_src : p.src), rs_mask = make_rsrc(has_mask ? p.mask_src : p.src);
    const rsrc_t rs_x = make_rsrc(MODE == 2 ? p.bnb.x : p.src), rs_bits = make_rsrc(has_bits ? (const void*)p.bnb.mask : (const void*)p.src);
    // per-column operands of this wave's column split (fixed for the whole launch)
    const rsrc_t rs_mean = make_rsrc(MODE == 2 ? p.bnb.mean : p.src), rs_istd = make_rsrc(MODE == 2 ? p.bnb.invstd : p.src);
    f32x4 s1[NP], s2[NP];
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
        s1[jp] = (f32x4){0.f, 0.f, 0.f, 0.f};
        s2[jp] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    f32x16 acc[NJ];
    int u = wave, kc = 0;                    // the step parked in Tb (u: local unit index)
    while (tile_of(u) < RT) {
#pragma unroll
      for (int par = 0; par < PF; ++par) {   // (unrolled: the register stage of the step after this one is a compile-time index)
        if (tile_of(u) >= RT) break;
        f32x4 (&nxt)[8] = R[(par + 1) % PF];
        if (kc == 0) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        }
        pw_wave_sync();
        {
            const float* a = Tb + li * PW_TB + 4 * lh;
            const float* b = Ws + (n0l + li) * LDW + kc * 64 + 4 * lh;
            if constexpr (NJ <= 2) {
                // few MFMAs per fragment set: the reads of kq + 1 are issued ahead of kq's MFMAs
                f32x4 fa = PW_FRAG(a), fb[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[j] = PW_FRAG(b + j * 32 * LDW);
#pragma unroll
                for (int kq = 0; kq < 8; ++kq) {
                    f32x4 fan = fa, fbn[NJ];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) fbn[j] = fb[j];
                    if (kq < 7) {
                        fan = PW_FRAG(a + (kq + 1) * 8);
#pragma unroll
                        for (int j = 0; j < NJ; ++j) fbn[j] = PW_FRAG(b + j * 32 * LDW + (kq + 1) * 8);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) PW_MFMA(acc[j], fb[j][e], fa[e]);
                    fa = fan;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) fb[j] = fbn[j];
                }
            } else {
#pragma unroll
                for (int kq = 0; kq < 8; ++kq) {
                    const f32x4 fa = PW_FRAG(a + kq * 8);
                    f32x4 fb[NJ];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) fb[j] = PW_FRAG(b + j * 32 * LDW + kq * 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) PW_MFMA(acc[j], fb[j][e], fa[e]);
                }
            }
        }
        pw_wave_sync();
        if (kc + 1 < nkc) {                  // inside a unit: park the next chunk, request the one PF steps on
            park(nxt);
            request(nxt, req_u, req_k);
            PW_ADVANCE();
            ++kc;
            continue;
        }
        // ---- epilogue: 64 output channels per pass through the wave's buffer.  Branch-free: every global access is a raw buffer
        // access whose offset is out of range for dead rows / columns / absent operands (loads return zeros, stores are dropped),
        // all loads of a pass are issued before the transposition, all its stores back to back after the arithmetic.
        const int m0 = tile_of(u) * 32;
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
            const int cl = jp * 64 + 4 * cg;             // column within the unit
            const bool colok = cl < UW;
            const int n = n0 + cl;
            // (NJ = 4 with the BatchNorm-backward operands: two half-batches of four rows, or the registers do not fit)
            constexpr int HB = (NJ == 4 && MODE == 2) ? 2 : 1, RH = 8 / HB;
#pragma unroll
            for (int h = 0; h < HB; ++h) {
                unsigned off[RH];
#pragma unroll
                for (int i = 0; i < RH; ++i) {
                    const int m = m0 + rr + 4 * (h * RH + i);
                    off[i] = (colok & (m < M)) ? 4u * (unsigned)(p.out_off + m * p.out_ld + n) : ZSG_OOB;
                }
                f32x4 xa[RH], xb[RH];                    // MODE 1: add_src, mask_src;  MODE 2: add_src, bn_x
                unsigned mb[RH];
                f32x4 mu, is;
                if (MODE == 2) {      // cold reads: in flight while the accumulators go through LDS
                    mu = buf_load4(rs_mean, colok ? 4u * (unsigned)n : ZSG_OOB);
                    is = buf_load4(rs_istd, colok ? 4u * (unsigned)n : ZSG_OOB);
#pragma unroll
                    for (int i = 0; i < RH; ++i) {
                        xa[i] = buf_load4(rs_add, has_add ? off[i] : ZSG_OOB);
                        xb[i] = buf_load4(rs_x, off[i]);
                        mb[i] = has_bits ? (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_bits, (int)(off[i] == ZSG_OOB ? ZSG_OOB : off[i] >> 4), 0, 0) : 0xfu;
                    }
                }
                if (MODE == 1) {
#pragma unroll
                    for (int i = 0; i < RH; ++i) {
                        xa[i] = buf_load4(rs_add, has_add ? off[i] : ZSG_OOB);
                        xb[i] = buf_load4(rs_mask, has_mask ? off[i] : ZSG_OOB);
                    }
                }
                if (h == 0) {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        if (2 * jp + jj < NJ) {
                            const f32x16& c = acc[(2 * jp + jj) < NJ ? (2 * jp + jj) : 0];
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                *(f32x4*)(Tb + li * PW_TB + jj * 32 + 8 * q + 4 * lh) = (f32x4){c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};
                        }
                    }
                    pw_wave_sync();
                }
                f32x4 v[RH];
#pragma unroll
                for (int i = 0; i < RH; ++i) v[i] = *(const f32x4*)(Tb + (rr + 4 * (h * RH + i)) * PW_TB + 4 * cg);
                if (h == HB - 1) {
                    pw_wave_sync();
                    if (jp == NP - 1) {                  // the buffer is free: the next unit's first chunk moves in (see above)
                        park(nxt);
                        request(nxt, req_u, req_k);
                        PW_ADVANCE();
                    }
                }
                if (MODE == 2) {
#pragma unroll
                    for (int i = 0; i < RH; ++i) {
                        v[i] += xa[i];
                        f32x4 g = v[i];
#pragma unroll
                        for (int e = 0; e < 4; ++e) g[e] = ((mb[i] >> e) & 1u) ? g[e] : 0.f;
                        // (no row test — measured: eight compare + branch pairs per pass cost 3-5 us per launch.  A dead row's
                        // accumulator, add_src, x and ReLU bits are all exact zeros (out-of-range loads), so it adds 0 * finite;
                        // a dead column's sums are never written out)
                        s1[jp] += g;
                        s2[jp] += g * ((xb[i] - mu) * is);
                        if (p.bnb.store_masked) v[i] = g;      // (wave-uniform: the stored dout is the masked gradient)
                    }
                } else {
                    if (MODE == 0 && p.stats) {          // (plain convolution: the host excludes bias / add / ReLU here)
#pragma unroll
                        for (int i = 0; i < RH; ++i) {    // (dead rows hold exact zeros, dead columns are never written out: no test)
                            s1[jp] += v[i];
                            s2[jp] += v[i] * v[i];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < RH; ++i) {
                        v[i] += cv0[jp];
                        if (MODE == 1) v[i] += xa[i];
                        if (p.relu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[i][e] = fmaxf(v[i][e], 0.f);
                        }
                        if (MODE == 1 && has_mask) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[i][e] = xb[i][e] > 0.f ? v[i][e] : 0.f;
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < RH; ++i) buf_store4(rs_out, off[i], v[i]);
            }
        }
        kc = 0;
        u += PW_WAVES;
      }
    }
#undef PW_ADVANCE

    // ---- one partial row per workgroup: lanes of a column group, then the waves of a column split, in a fixed order ------------
    if (p.stats) {
        __syncthreads();                    // every wave has left the streaming loop: the filter panel is no longer needed
        float* red = smem;                  // [PW_WAVES][2][UW]
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
            f32x4 a = s1[jp], b = s2[jp];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] += __shfl_xor(a[e], 16, 64);
                a[e] += __shfl_xor(a[e], 32, 64);
                b[e] += __shfl_xor(b[e], 16, 64);
                b[e] += __shfl_xor(b[e], 32, 64);
            }
            const int cl = jp * 64 + 4 * cg;
            if (rr == 0 && cl < UW) {
                *(f32x4*)(red + (wave * 2 + 0) * UW + cl) = a;
                *(f32x4*)(red + (wave * 2 + 1) * UW + cl) = b;
            }
        }
        __syncthreads();
        for (int n = tid; n < NB; n += 64 * PW_WAVES) {      // (n: column within the panel)
            const int ns = n / UW, cl = n - ns * UW;
            float a = 0.f, b = 0.f;
            for (int w = ns; w < PW_WAVES; w += NS) {
                a += red[(w * 2 + 0) * UW + cl];
                b += red[(w * 2 + 1) * UW + cl];
            }
            float* o = p.stats + (size_t)rgid * 2 * p.N + cb * NB;
            o[n] = a;
            o[p.N + n] = b;
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------

// geometry the kernel covers: ONE dense 1x1 / stride-1 segment (row m of the GEMM = pixel m of both tensors), K a multiple of 64,
// the filter cut into ncb = 1, 2, 4 or 8 panels of nb = N / ncb rows such that a panel + the eight tile buffers fit the CU's LDS
// and a panel is 1, 2, 4 or 8 units of `uw` columns (the smallest such ncb is used)
static bool pw_geometry_ok(const zsg_conv_desc* d, int uw, const char** why, int* ncb_out = nullptr) {
    static const char* msg;
    const char*& w = why ? *why : msg;
    if (d->nseg != 1 || d->merge_x) { w = "one segment, no merge_x"; return false; }
    const zsg_seg& s = d->seg[0];
    if (s.ty.n != 1 || s.tx.n != 1 || s.ty.d0 != 0 || s.tx.d0 != 0 || s.sy != 1 || s.sx != 1 || s.osy != 1 || s.osx != 1 || s.opy != 0 || s.opx != 0) {
        w = "1x1, stride 1, no padding";
        return false;
    }
    if (s.src_H != s.rows_y || s.src_W != s.rows_x || s.out_W != s.rows_x || s.src_bstride != (int64_t)s.rows_y * s.rows_x * d->src_ld ||
        s.out_bstride != (int64_t)s.rows_y * s.rows_x * d->out_ld) {
        w = "dense pixel rows in both tensors";
        return false;
    }
    if (uw != 32 && uw != 64 && uw != 128) { w = "unit width 32 / 64 / 128"; return false; }
    if (d->C <= 0 || (d->C % 64) != 0 || d->N <= 0 || (d->N % uw) != 0) { w = "C % 64 == 0 and N % unit width == 0"; return false; }
    int ncb = 0;
    for (int c = 1; c <= 8 && !ncb; c *= 2) {
        if (d->N % c) break;
        const int nb = d->N / c;
        if (nb % uw) break;
        const int ns = nb / uw;
        if ((ns == 1 || ns == 2 || ns == 4 || ns == 8) && ((size_t)nb * (d->C + 4) + (size_t)PW_WAVES * 32 * PW_TB) * sizeof(float) <= PW_LDS_MAX) ncb = c;
    }
    if (!ncb) { w = "a filter panel of 1, 2, 4 or 8 units (N / 1, 2, 4 or 8 rows) that fits the LDS"; return false; }
    if ((d->src_ld % 4) || (d->out_ld % 4) || (s.src_off % 4) || (s.out_off % 4) || (d->wt_ld % 4) || (d->wc0 % 4)) { w = "16-byte aligned rows"; return false; }
    const int64_t rows = (int64_t)d->B * s.rows_y * s.rows_x;
    if (rows <= 0 || rows >= (1ll << 30) || s.src_off + rows * d->src_ld >= (1ll << 29) || s.out_off + rows * d->out_ld >= (1ll << 29)) {
        w = "tensor exceeds 2^29 elements";
        return false;
    }
    if (ncb_out) *ncb_out = ncb;
    return true;
}

// row groups of a launch (= rows of BatchNorm partials it writes); the grid is row groups x column blocks, at most one workgroup per CU
static int pw_row_groups(const zsg_conv_desc* d, int uw, int ncb) {
    const int64_t rows = (int64_t)d->B * d->seg[0].rows_y * d->seg[0].rows_x;
    const int units = cdiv(rows, 32) * (d->N / ncb / uw);      // per column block
    const int g = cdiv(units, PW_WAVES), gmax = ZSG_NUM_CU / ncb;
    return g < gmax ? g : gmax;
}

// rows of BatchNorm partials a zsg_conv_igemm / zsg_conv_igemm_bnb launch with this descriptor (and its tile_hint) writes
extern "C" int32_t zsg_conv_igemm_partial_rows(const zsg_conv_desc* d) {
    if (!d || !d->tile_hint) return -1;
    const int bm = d->tile_hint & 0xff, bn = (d->tile_hint >> 8) & 0xff;
    if (bm == 32 && d->merge_x) return zsg_conv_mx_ok(d, nullptr) ? zsg_conv_mx_groups(d) : -1;
    if (bm == 32) {
        int ncb = 0;
        return pw_geometry_ok(d, bn, nullptr, &ncb) ? pw_row_groups(d, bn, ncb) : -1;
    }
    if (bm <= 0) return -1;
    int64_t t = 0;
    for (int s = 0; s < d->nseg; ++s) t += cdiv((int64_t)d->B * d->seg[s].rows_y * d->seg[s].rows_x, bm);
    return (int32_t)t;
}

template <int NJ, int MODE>
static int pw_launch2(const PwParams& p, int grid, size_t lds, hipStream_t st, double flops, const char* kname) {
    static bool attr_done[ZSG_MAX_DEV] = {};       // per device (a process may drive several GPUs); idempotent
    int dev = 0;
    (void)hipGetDevice(&dev);
    ZSG_REQUIRE(dev >= 0 && dev < ZSG_MAX_DEV, "conv_pw: device %d", dev);
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)pw_kernel<NJ, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS_MAX);
        if (e != hipSuccess) ZSG_FAIL(-3, "conv_pw: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done[dev] = true;
    }
    ZSG_PROF(kname, st, flops, 4.0 * ((double)p.M * (p.K + p.N * (p.add_src ? 2.0 : 1.0)) + (double)p.N * p.K));
    ZSG_LAUNCH((pw_kernel<NJ, MODE>), dim3(grid), dim3(64 * PW_WAVES), lds, st, p);
    ZSG_CHECK_LAUNCH("conv_pw");
    return 0;
}
template <int NJ>
static int pw_launch1(const PwParams& p, int grid, size_t lds, hipStream_t st, double flops, const char* k0, const char* k1, const char* k2) {
    if (p.bnb.x) {
        // (128-channel units have no BatchNorm-backward form: x / add / bits of a pass next to 64 accumulator registers spill 126
        // registers; zsg_conv_pw_launch runs such launches as 64-channel units and refuses the one geometry that cannot)
        if constexpr (NJ == 4) ZSG_FAIL(-1, "conv_igemm_bnb: the streaming 1x1 kernel has no 128-channel-unit variant for this geometry");
        else return pw_launch2<NJ, 2>(p, grid, lds, st, flops, k2);
    }
    if (p.add_src || p.mask_src) return pw_launch2<NJ, 1>(p, grid, lds, st, flops, k1);
    return pw_launch2<NJ, 0>(p, grid, lds, st, flops, k0);
}

// called by conv_igemm_impl (igemm.hip) for tile_hint BM == 32: uw = the hint's BN field
int zsg_conv_pw_launch(const zsg_conv_desc* d, int uw_hint, const float* src, const float* wt, float* out, const float* bias, const float* add_src,
                       const float* mask_src, float* bn_partials, const BnbDev* bnb, hipStream_t st) {
    int uw = uw_hint;
    const char* why = "";
    int ncb = 0;
    ZSG_REQUIRE(pw_geometry_ok(d, uw, &why, &ncb), "conv_igemm: the streaming 1x1 kernel (tile_hint BM = 32) needs %s", why);
    const uintptr_t al = (uintptr_t)src | (uintptr_t)wt | (uintptr_t)out | (uintptr_t)bias | (uintptr_t)add_src | (uintptr_t)mask_src;
    ZSG_REQUIRE((al & 15) == 0, "conv_igemm: the streaming 1x1 kernel needs 16-byte aligned operands");
    PwParams p;
    memset(&p, 0, sizeof(p));
    const zsg_seg& s = d->seg[0];
    p.src = src; p.wt = wt; p.out = out; p.bias = bias; p.add_src = add_src; p.mask_src = mask_src; p.stats = bn_partials;
    p.M = (int)((int64_t)d->B * s.rows_y * s.rows_x);
    p.K = d->C; p.N = d->N; p.src_ld = d->src_ld; p.out_ld = d->out_ld; p.wt_ld = d->wt_ld;
    p.wc0 = d->wc0 + (s.ty.w0 * d->wS + s.tx.w0) * d->wC;          // (the one tap's position in a weight row)
    p.src_off = (int)s.src_off; p.out_off = (int)s.out_off; p.relu = d->relu;
    p.NCB = ncb;
    p.NB = d->N / ncb;
    p.NS = p.NB / uw;
    p.rt = cdiv(p.M, 32);
    p.ns_shift = p.NS == 8 ? 3 : p.NS == 4 ? 2 : p.NS == 2 ? 1 : 0;
    if (bnb) {
        ZSG_REQUIRE(bn_partials && bnb->x && bnb->mean && bnb->invstd && !bias && !d->relu && !mask_src, "conv_igemm_bnb: bad argument");
        ZSG_REQUIRE((((uintptr_t)bnb->x | (uintptr_t)bnb->mean | (uintptr_t)bnb->invstd) & 15) == 0, "conv_igemm_bnb: operands not 16-byte aligned");
        p.bnb = *bnb;
        p.bnb.store_masked = d->epi_flags & 1;
    } else if (bn_partials) {
        ZSG_REQUIRE(!bias && !add_src && !d->relu && !mask_src, "conv_igemm: BN-statistics fusion needs a plain (bias-free) convolution");
    }
    const int grid = pw_row_groups(d, uw, ncb) * ncb;      // (from the hint's unit width: what zsg_conv_igemm_partial_rows promised)
    // BatchNorm-backward partials with 128-channel units: x / add / bits of a pass + 64 accumulator registers do not fit next to
    // each other (the half-batch form of MODE 2 measured 105 us where the 64-channel units take 51) — run the 64-channel units on
    // the SAME grid (the unit dealing works for any grid; the row count of the partials is unchanged).
    if (bnb && uw == 128 && p.NS * 2 <= 8) {
        uw = 64;
        p.NS *= 2;
        p.ns_shift += 1;
    }
    const size_t lds = ((size_t)p.NB * (d->C + 4) + (size_t)PW_WAVES * 32 * PW_TB) * sizeof(float);
    const double flops = 2.0 * p.M * (double)d->N * d->C;
    if (uw == 128) return pw_launch1<4>(p, grid, lds, st, flops, "pw_kernel<4, 0>", "pw_kernel<4, 1>", "pw_kernel<4, 2>");
    if (uw == 64) return pw_launch1<2>(p, grid, lds, st, flops, "pw_kernel<2, 0>", "pw_kernel<2, 1>", "pw_kernel<2, 2>");
    return pw_launch1<1>(p, grid, lds, st, flops, "pw_kernel<1, 0>", "pw_kernel<1, 1>", "pw_kernel<1, 2>");
}
