// wino.hip — Winograd F(2x2, 3x3) convolution (forward and stride-1 data gradient) on fp32 MFMA, NHWC, gfx950.
//
// For the 3x3 / stride 1 / pad 1 convolutions (72 % of the network's multiply-adds: the shared head, the FPN output
// convolutions, every bottleneck's conv2) the direct implicit GEMM spends 9 MACs per (pixel, cin, cout); the minimal
// filtering algorithm F(2x2,3x3) spends 16 per 2x2 output tile = 4 per pixel (2.25x fewer) at fp32 (transform constants
// are 0, +-1, +-1/2: exact scalings, a handful of extra roundings per output).
//
//   Y(2x2) = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A          d: 4x4 input patch, g: 3x3 filter
//
// GEMM view: 16 independent GEMMs ("positions" p = j*4 + i of the 4x4 transformed domain) that share M / N / K:
//   M = tiles (segment, b, ty, tx)   A_p[tile][c] = (B^T d B)[i][j]      transformed ON THE FLY while staging to LDS
//   N = output channels              B_p[n][c]    = (G g G^T)[i][j]      pre-transformed once per step (zsg_wino_weights)
//   K = input channels, 8 per LDS stage
// Work split: a workgroup owns (32*TM tiles) x (32*TN channels); wave (wm, wn, ph) owns one 32x32 sub-block and the
// 8 positions with i in {2ph, 2ph+1} (8 x 16 accumulator registers); the two position halves are combined in the
// epilogue (the output transform is linear), which also undoes the tile -> 2x2 pixel mapping through LDS so that every
// lane stores 16 contiguous bytes.
// A loader: lane (tile, 4-channel group g, patch row q) loads the 4 pixels of its patch row (4 x 16 B), applies the row
// transform in registers and gets the column transform from its quad neighbours with DPP quad_perm (no LDS round trip),
// (one v_fmac with a DPP operand per value), then writes V[q][0..3] with four conflict-free ds_write_b128.  B loader: 16-byte copies of the pre-chunked U image
// [c/8][p][n][8] (contiguous 2 KB runs).  LDS rows are 8 floats; the two 16-byte halves of a row are XOR-swizzled with
// bit 3 of the row index, which makes the ds_read_b128 fragment reads (lane = row, half = k-group) conflict-free without
// padding.  K permutation as in igemm.hip: value e of a lane's b128 feeds the e-th of four MFMAs.
#include "common.h"
#include "bn_tail.h"

ZSG_DEFINE_PRIO_FLAG()

#define WN_CK 8

struct WnSegDev {
    int tiles_y, tiles_x, tiles;   // tile grid per image; tiles = B * tiles_y * tiles_x
    int blk0;                      // first M block of the segment
    int H, W;
    int src_off, src_bstride, out_off, out_bstride;   // elements
};

struct WnParams {
    const float* src;
    const float* U;
    float* out;
    const float* bias;
    const float* add_src;
    const float* mask_src;
    float* stats;        // optional BatchNorm partials [m_blocks][2][N]
    BnbDev bnb;          // bnb.x != nullptr: `stats` receives BatchNorm-BACKWARD partials of the stored values (common.h)
    BnTail tail;         // tail.tickets != nullptr: the last-arriving block of a column block finalises the statistics (bn_tail.h)
    int C, N, Npad, src_ld, out_ld, relu, nseg;
    int m_blocks, n_blocks, splits, chunks, vec, add_is_out;
    double alg_bytes;    // host only: algorithmic HBM bytes of the launch (profile)
    // stream-K (SK kernels only; tile_hint bits 28-29, as igemm.hip): sk_grid workgroups share the (block tile, 8-channel chunk) units
    // evenly, sk_per consecutive units each; the workgroup that holds a tile's first chunk completes it
    int sk_grid, sk_per;
    float* sk_ws;        // one partial OUTPUT tile [TB * 4 pixels][BN] per workgroup (the output transform is linear: partial sums of Y add)
    unsigned* sk_flags;  // one word per workgroup: "its partial tile is published"; zero at entry, zero at exit
    WnSegDev seg[ZSG_MAX_SEG];
};

typedef __attribute__((address_space(3))) float lds_f32;

__device__ __forceinline__ float quad_other(float v) {
    // quad_perm [2,2,1,1]: the patch row this lane's column transform combines its own row with
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x5A, 0xf, 0xf, true));
}

// PS position groups per 32x32 sub-block: wave (ph, wm, wn) holds the 16/PS positions with i in {2ph, 2ph+1} (PS = 2) or
// i = ph (PS = 4).  PS = 4 doubles the waves per SIMD for the same tile (64 instead of 128 accumulator registers each).
//
// SK (stream-K, round 6): the bottlenecks' conv2 at 19^2 / 10^2 are grids of 184 / 104 blocks of 32 tiles x 64 channels on 256 CUs — one
// round with a quarter / more than half of the chip idle.  With SK the launch has 256 workgroups and the (block tile, chunk) units are
// dealt evenly in tile-major order, exactly as in igemm.hip: a workgroup's range is at most the TAIL of one tile (a producer pass: K
// loop, output transform, the partial OUTPUT tile published write-through, flag) followed by the HEAD of the next (the finishing pass:
// K loop, output transform, + the partial tiles of the workgroups behind it in workgroup order — deterministic — and the complete
// epilogue).  What is exchanged is the 2x2-pixel output tile (32 or 64 KB), not the 16 transformed-domain accumulators (4x as much):
// the output transform is linear.  Producers never wait, so the finisher's poll cannot deadlock.
template <int TM, int TN, int PS, bool SK = false>
__global__ __launch_bounds__(64 * PS * TM * TN, 2) void wino_kernel(const WnParams p) {
    ZSG_SET_MAIN_PRIO();
    constexpr int NT = 64 * PS * TM * TN;        // PS position groups x TM x TN waves
    constexpr int NP = 16 / PS;                  // positions (accumulator tiles) per wave
    constexpr int TB = 32 * TM, BN = 32 * TN;
    constexpr int SA = TB * 8 + 8;               // floats between positions of the A stage (+8: conflict-free quad writes)
    constexpr int SB = BN * 8;
    constexpr int IA = (TB * 8 + NT - 1) / NT;   // A loader items (tile, g, q) per thread (NT > TB*8: only the first TB*8 threads load)
    constexpr bool A_ALL = (TB * 8 % NT) == 0;
    constexpr int IB = 32 * BN / NT;             // B loader 16-byte copies per thread
    static_assert(IB >= 1 && (A_ALL || IA == 1), "tile too small for the thread count");
    const bool a_thr = A_ALL || (int)threadIdx.x < TB * 8;      // wave-uniform
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                            // [2][16][SA]
    float* Bs = smem + 2 * 16 * SA;              // [2][16][SB]
    int* rowinfo = (int*)(smem + 2 * 16 * (SA + SB));   // [TB][2]: output offset of pixel (2ty, 2tx) | -1 ; validity bits

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int ph = wave / (TM * TN);
    const int wmn = wave % (TM * TN);
    const int wm = wmn / TN, wn = wmn % TN;
    const int li = lane & 31, lh = lane >> 5;

    const int n_mn = p.m_blocks * p.n_blocks;
    const int src_ld = p.src_ld;
    // SK: this workgroup's range of (tile, chunk) units, [sk_u, sk_u1) in tile-major order (see igemm.hip)
    int sk_l = 0, sk_u = 0, sk_u1 = 0;
    if constexpr (SK) {
        sk_l = xcd_remap(blockIdx.x, p.sk_grid);
        sk_u = sk_l * p.sk_per;
        sk_u1 = min(n_mn * p.chunks, sk_u + p.sk_per);
        if (sk_u >= sk_u1) return;
    }
    // the tile of the current pass (what the epilogue behind the pass loop completes)
    int split = 0, mb = 0, nb = 0, n0 = 0;
    int sk_t = 0, sk_k0 = 0, sk_k1 = 0;          // SK: the tile of this pass and its chunks [sk_k0, sk_k1)
    WnSegDev sg;
    f32x16 acc[NP];
    constexpr int LDC = BN + 4;
    float* ct = smem;                               // [TB*4][LDC] — the output tile; reuses the staging area
    float* red = smem + TB * 4 * LDC;               // [2][RPP][BN] (statistics)
    constexpr int E_CG = BN / 4, E_RPP = NT / E_CG, E_NR = TB * 4 / E_RPP;
    static_assert((TB * 4) % E_RPP == 0, "epilogue row passes");
    f32x4 xpre[E_NR], apre[E_NR];
    unsigned mpre[E_NR];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  // One pass, unless SK (producer pass, then the finishing pass: the epilogue stays OUTSIDE the loop, see igemm.hip)
  for (;;) {
    int bid;
    if constexpr (SK) {
        sk_t = sk_u / p.chunks;
        sk_k0 = sk_u - sk_t * p.chunks;
        sk_k1 = min(p.chunks, sk_k0 + (sk_u1 - sk_u));
        bid = sk_t;
    } else {
        split = blockIdx.x / n_mn;
        bid = xcd_remap(blockIdx.x - split * n_mn, n_mn);
    }
    mb = bid / p.n_blocks;
    nb = bid % p.n_blocks;
    int si = 0;
#pragma unroll
    for (int s = 1; s < ZSG_MAX_SEG; ++s)
        if (s < p.nseg && mb >= p.seg[s].blk0) si = s;
    sg = p.seg[si];
    const int m0 = (mb - sg.blk0) * TB;
    n0 = nb * BN;

    // ---- loader state (fixed over the K loop) ------------------------------------------------------------------------
    const int q = tid & 3, g = (tid >> 2) & 1;
    // Loop-invariant per-lane byte offsets (zero padding = out-of-range offset, baked in); the chunk's channel offset travels in an
    // SGPR (the buffer instructions' soffset), so a load costs NO address arithmetic in the K loop.  On gfx950 nothing co-issues
    // with a SIMD's fp32 MFMA stream (tools/ubench/mfma_coissue.hip: a partner wave retires ~0 instructions while its SIMD mate
    // multiplies, a wave's own VALU costs ~6 cycles each on top of its MFMAs), so every instruction of this loop is paid in full.
    unsigned a_voff[IA][4], a_voff_t[IA][4];       // a_voff_t: the last chunk of a C % 8 == 4 tensor (its upper channel quad is dead)
    int a_lds[IA];
    const bool c_tail = (p.C & 4) != 0;
#pragma unroll
    for (int ia = 0; ia < IA; ++ia) {
        const int t = ((tid + NT * ia) >> 3) % TB;
        const int m = m0 + t;
        const bool ok = a_thr & (m < sg.tiles);
        const int mm = ok ? m : 0;
        const int per = sg.tiles_y * sg.tiles_x;
        const int b = mm / per;
        const int rem = mm - b * per;
        const int ty = rem / sg.tiles_x;
        const int tx = rem - ty * sg.tiles_x;
        const int y = 2 * ty - 1 + q, x0 = 2 * tx - 1;
        const bool rok = ok & ((unsigned)y < (unsigned)sg.H);
        const int base = sg.src_off + b * sg.src_bstride + (y * sg.W + x0) * src_ld + 4 * g;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool pok = rok & ((unsigned)(x0 + c) < (unsigned)sg.W);
            a_voff[ia][c] = pok ? 4u * (unsigned)(base + c * src_ld) : ZSG_OOB;
            a_voff_t[ia][c] = (pok && !(c_tail && g)) ? a_voff[ia][c] : ZSG_OOB;
        }
        a_lds[ia] = q * SA + t * 8 + 4 * (g ^ ((t >> 3) & 1));
        if (q == 0 && g == 0 && a_thr) {
            rowinfo[2 * t] = ok ? sg.out_off + b * sg.out_bstride + ((2 * ty) * sg.W + 2 * tx) * p.out_ld : -1;
            rowinfo[2 * t + 1] = ((2 * tx + 1 < sg.W) ? 1 : 0) | ((2 * ty + 1 < sg.H) ? 2 : 0);
        }
    }
    // B loader: LDS-DMA (buffer_load ... lds): one wave-instruction lands 1 KB = 32 rows of one position, lane-linear, so
    // the row swizzle is applied on the SOURCE side (lane l fills slot (row l>>1, half l&1) with half (l&1)^swz(row)).
    // The LDS destination is wave-uniform by construction: computed from a readfirstlane'd wave index so that it lives in SGPRs
    // (M0 is then one s_add away, no v_readfirstlane per piece).
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned b_voff[IB];
    int b_dst[IB];
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) {
        const int kb = wave_u + (NT / 64) * ib;        // 1 KB piece index: pos * (BN/32) + row group
        const int pos = kb / (BN / 32), rg = kb % (BN / 32);
        const int r = rg * 32 + (lane >> 1);
        b_voff[ib] = 4u * (unsigned)((pos * p.Npad + n0 + r) * 8 + 4 * ((lane & 1) ^ ((r >> 3) & 1)));
        b_dst[ib] = pos * SB + rg * 256;               // floats, wave-uniform (SGPR)
    }
    const rsrc_t rsrc_a = make_rsrc(p.src);
    const rsrc_t rsrc_b = make_rsrc(p.U);
    const int ustep = 16 * p.Npad * 8;           // U elements per 8-channel chunk

    int c0 = 0, nc = p.chunks;
    if (p.splits > 1) {
        const int per = (p.chunks + p.splits - 1) / p.splits;
        c0 = min(split * per, p.chunks);
        nc = min(per, p.chunks - c0);
    }
    if constexpr (SK) {
        c0 = sk_k0;
        nc = sk_k1 - sk_k0;
    }

    f32x4 ra[IA][4];
    auto load_a = [&](int c, bool live) {          // c, live: wave-uniform
        if (!a_thr || !live) return;      // (a dead prefetch leaves ra as it is: it is stored into the idle buffer, never read)
        const int so = c * (WN_CK * 4);      // bytes, SGPR
        if (!(c_tail && c == p.chunks - 1)) {
#pragma unroll
            for (int ia = 0; ia < IA; ++ia)
#pragma unroll
                for (int col = 0; col < 4; ++col)
                    ra[ia][col] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)a_voff[ia][col], so, 0));
        } else {
#pragma unroll
            for (int ia = 0; ia < IA; ++ia)
#pragma unroll
                for (int col = 0; col < 4; ++col)
                    ra[ia][col] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)a_voff_t[ia][col], so, 0));
        }
    };
    auto load_b = [&](int c, int buf) {              // straight into LDS, no registers
#if __HIP_DEVICE_COMPILE__      // (the host pass of hipcc cannot type-check the LDS address-space cast; it never runs this body)
        lds_f32* b = (lds_f32*)(Bs + buf * 16 * SB);
        const int so = c * (ustep * 4);   // bytes, SGPR
#pragma unroll
        for (int ib = 0; ib < IB; ++ib)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, b + b_dst[ib], 16, (int)b_voff[ib], so, 0, 0);
#endif
    };
    // Column transform B^T across the quad (lane q holds row q of d B): V[0] = t0 - t2, V[1] = t1 + t2, V[2] = t2 - t1 and
    // V'[3] = t3 - t1 = -V[3] (zsg_wino_weights negates row 3 of U to match): every lane computes own + sgn * other with
    // ONE cross-lane operand, i.e. one v_fmac_f32 with a DPP source per value.
    const float sgn = (q == 1) ? 1.f : -1.f;
    auto store_a = [&](int buf) {
        if (!a_thr) return;
        float* a = As + buf * 16 * SA;
#pragma unroll
        for (int ia = 0; ia < IA; ++ia) {
            f32x4 t[4];                            // row transform (d B): this lane's patch row q
            t[0] = ra[ia][0] - ra[ia][2];
            t[1] = ra[ia][1] + ra[ia][2];
            t[2] = ra[ia][2] - ra[ia][1];
            t[3] = ra[ia][1] - ra[ia][3];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(quad_other(t[j][e]), sgn, t[j][e]);
                *(f32x4*)(a + a_lds[ia] + j * 4 * SA) = v;
            }
        }
    };

#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    if (nc > 0) {
        load_b(c0, 0);
        load_a(c0, true);
        store_a(0);
        if (ph & 1) load_a(c0 + 1, nc > 1);        // the late groups transform FIRST in every chunk: their next chunk is prefetched here
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int pbase = (PS == 2) ? 2 * ph : ph;
    const int frag_a = (wm * 32 + li) * 8 + 4 * (lh ^ ((li >> 3) & 1)) + pbase * SA;
    const int frag_b = (wn * 32 + li) * 8 + 4 * (lh ^ ((li >> 3) & 1)) + pbase * SB;
    auto mfma_pos = [&](const float* a, const float* b, int pl) {       // position p = j*4 + i: PS 2: pl = j*2 + il, i = 2ph + il; PS 4: pl = j, i = ph
        const int po = (PS == 2) ? (pl >> 1) * 4 + (pl & 1) : pl * 4;
        const f32x4 fa = *(const f32x4*)(a + po * SA);
        const f32x4 fb = *(const f32x4*)(b + po * SB);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[pl] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], acc[pl], 0, 0, 0);
    };
    // One chunk = MFMAs on buffer it&1 + (A transform, B LDS-DMA) of chunk it+1 into the other buffer; the two are
    // independent between two barriers, so the position groups — whose waves share the SIMDs (wave w, w + TM*TN, ...) —
    // alternate their order: while one group transforms (VALU / LDS writes) its SIMD partner issues MFMAs.
    //   early groups (ph even): loads of chunk it+1 | MFMAs | transform it+1        (load -> use: the MFMA phase)
    //   late  groups (ph odd) : transform it+1 (loaded one chunk ago) | loads of chunk it+2 | MFMAs
    constexpr int NP1 = NP - NP / 4;               // MFMA positions before the transform may start interleaving
    if ((ph & 1) == 0) {
        for (int it = 0; it < nc; ++it) {
            const float* a = As + (it & 1) * 16 * SA + frag_a;
            const float* b = Bs + (it & 1) * 16 * SB + frag_b;
            if (it + 1 < nc) load_b(c0 + it + 1, (it + 1) & 1);
            load_a(c0 + it + 1, it + 1 < nc);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pl = 0; pl < NP1; ++pl) mfma_pos(a, b, pl);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pl = NP1; pl < NP; ++pl) mfma_pos(a, b, pl);
            store_a((it + 1) & 1);    // (after the last chunk: into the idle buffer, never read)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else {
        for (int it = 0; it < nc; ++it) {
            const float* a = As + (it & 1) * 16 * SA + frag_a;
            const float* b = Bs + (it & 1) * 16 * SB + frag_b;
            if (it + 1 < nc) load_b(c0 + it + 1, (it + 1) & 1);
            store_a((it + 1) & 1);
            load_a(c0 + it + 2, it + 2 < nc);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) mfma_pos(a, b, pl);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    // ---- output transform: this wave's share of Y = A^T M A (A^T = [[1,1,1,0],[0,1,-1,-1]]) ---------------------------------
    // z[i][b] = sum_j A^T[b][j] M[i][j];  Y[a][b] = sum_i A^T[a][i] z[i][b].  The position groups add their partial Y through
    // LDS one after the other (fixed order).
    // BatchNorm-backward fusion: this thread's x values and ReLU bits (cold HBM reads) are requested before the output transform
    if (p.bnb.x && (!SK || sk_k0 == 0)) {
        const int en = n0 + 4 * (tid % E_CG);
#pragma unroll
        for (int i = 0; i < E_NR; ++i) {
            const int row = tid / E_CG + E_RPP * i;
            const int tl = row >> 2, px = row & 3;
            const int ro = rowinfo[2 * tl], fl = rowinfo[2 * tl + 1];
            const bool ok = !(ro < 0 || ((px & 1) && !(fl & 1)) || ((px & 2) && !(fl & 2))) & (en < p.N);
            const size_t o = ok ? (size_t)(ro + ((px >> 1) * sg.W + (px & 1)) * p.out_ld) + en : 0;
            xpre[i] = *(const f32x4*)(p.bnb.x + o);
            mpre[i] = p.bnb.mask ? p.bnb.mask[o >> 2] : 0xfu;
            if (p.add_src) apre[i] = *(const f32x4*)(p.add_src + o);
        }
    }
    if (PS == 4) {
        // Group ph holds row i = ph of M: z0 = m0 + m1 + m2, z1 = m1 - m2 - m3 along j, and Y = A^T z with A^T[.][i] = (1,0), (1,1),
        // (1,-1), (0,-1): pixels (0,1) get z(0) + z(1) + z(2), pixels (2,3) get z(1) - z(2) - z(3).  Three rounds in which TWO groups
        // work on disjoint pixel rows (fixed order: deterministic) — A: group 0 stores (0,1), group 3 stores (2,3); B: group 1 adds
        // to (0,1), group 2 to (2,3); C: group 1 adds to (2,3), group 2 to (0,1) — instead of four rounds of one group touching all
        // four pixels (the other twelve waves idle at the barrier): 160 instead of 448 LDS operations on the critical path.
#pragma unroll
        for (int rnd = 0; rnd < 3; ++rnd) {
            const bool lo = (rnd == 0) ? (ph == 0) : ((rnd == 1) ? (ph == 1) : (ph == 2));      // this group writes pixels (0,1)
            const bool hi = (rnd == 0) ? (ph == 3) : ((rnd == 1) ? (ph == 2) : (ph == 1));      // this group writes pixels (2,3)
            if (lo | hi) {
                const float sgn = (hi && ph != 1) ? -1.f : 1.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float z0 = sgn * (acc[0][e] + acc[1][e] + acc[2][e]), z1 = sgn * (acc[1][e] - acc[2][e] - acc[3][e]);
                    const int tl = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    float* o = ct + (tl * 4 + (hi ? 2 : 0)) * LDC + wn * 32 + li;
                    if (rnd == 0) {
                        o[0] = z0;
                        o[LDC] = z1;
                    } else {
                        o[0] += z0;
                        o[LDC] += z1;
                    }
                }
            }
            __syncthreads();
        }
    } else {
#pragma unroll
        for (int hh = 0; hh < PS; ++hh) {
            if (ph == hh) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float y[4];
                    const float z00 = acc[0][e] + acc[2][e] + acc[4][e], z01 = acc[2][e] - acc[4][e] - acc[6][e];
                    const float z10 = acc[1][e] + acc[3][e] + acc[5][e], z11 = acc[3][e] - acc[5][e] - acc[7][e];
                    if (hh == 0) {                  // rows i = 0, 1
                        y[0] = z00 + z10; y[1] = z01 + z11; y[2] = z10; y[3] = z11;
                    } else {                        // rows i = 2, 3
                        y[0] = z00; y[1] = z01; y[2] = -z00 - z10; y[3] = -z01 - z11;
                    }
                    const int tl = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    float* o = ct + (tl * 4) * LDC + wn * 32 + li;
#pragma unroll
                    for (int px = 0; px < 4; ++px) {
                        if (hh == 0) o[px * LDC] = y[px];
                        else o[px * LDC] += y[px];
                    }
                }
            }
            __syncthreads();
        }
    }

    // ---- stream-K hand-off, producer side: the partial output tile is published, then the next pass -------------------------------
    if constexpr (!SK) break;
    else {
        if (sk_k0 == 0) break;                        // the head of a tile (or a whole tile): the finishing pass
        const rsrc_t rs = make_rsrc(p.sk_ws + (size_t)sk_l * (TB * 4 * BN));
        const int cg = tid % E_CG, rr = tid / E_CG;
#pragma unroll
        for (int i = 0; i < E_NR; ++i) {
            const int row = rr + E_RPP * i;
            const f32x4 v = *(const f32x4*)(ct + row * LDC + 4 * cg);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, 4 * (row * BN + 4 * cg), 0, 16);      // sc1: write-through
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave drains its own stores ...
        __syncthreads();                                      // ... before the one flag store (also: ct / the staging area are free again)
        if (tid == 0) __hip_atomic_store(p.sk_flags + sk_l, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sk_u += sk_k1 - sk_k0;
        if (sk_u >= sk_u1) return;                    // this workgroup's range ended inside the tile
    }
  }
    // ---- stream-K hand-off, finisher side: + the partial tiles of the workgroups sk_l + 1 .. last, in that order -------------------------
    if constexpr (SK) {
        if (sk_k1 < p.chunks) {
            const int last = ((sk_t + 1) * p.chunks - 1) / p.sk_per;
            if (tid == 0) {
                for (int j = sk_l + 1; j <= last; ++j) {
                    int spins = 0;      // bounded poll (~1 s), see igemm.hip: a lost producer poisons the launch (ZSG_SK_ERR_WORD), it does not hang it
                    while (__hip_atomic_load(p.sk_flags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(4);
                    if (spins >= (1 << 20)) __hip_atomic_store(p.sk_flags + ZSG_SK_ERR_WORD, 0xdeadu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(p.sk_flags + j, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // zero again for the next launch
                }
            }
            __syncthreads();
            const int cg = tid % E_CG, rr = tid / E_CG;
            for (int j = sk_l + 1; j <= last; ++j) {              // fixed order: deterministic
                const rsrc_t rs = make_rsrc(p.sk_ws + (size_t)j * (TB * 4 * BN));
                f32x4 v[E_NR];
#pragma unroll
                for (int i = 0; i < E_NR; ++i)
                    v[i] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs, 4 * ((rr + E_RPP * i) * BN + 4 * cg), 0, 16));
#pragma unroll
                for (int i = 0; i < E_NR; ++i) *(f32x4*)(ct + (rr + E_RPP * i) * LDC + 4 * cg) += v[i];      // (each thread owns its (row, column group) cells)
            }
            __syncthreads();
        }
    }

    // ---- epilogue: bias, residual / accumulate, relu, relu-mask, BatchNorm partial statistics ---------------------------
    const bool vec_ok = p.vec && (p.splits == 1);
    if (vec_ok) {
        constexpr int CG = BN / 4;
        constexpr int RPP = NT / CG;
        const int cg = tid % CG, rr = tid / CG;
        const int n = n0 + 4 * cg;
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
        const bool bnb = p.bnb.x != nullptr;
        const bool tail_on = p.tail.tickets != nullptr;      // (host: only with this vectorised epilogue, unsplit)
        static_assert(E_NR == TB * 4 / RPP && E_CG == CG, "one row schedule for the prefetch and both epilogue passes");
        if (n < p.N) {
            f32x4 bv = {0.f, 0.f, 0.f, 0.f}, mu = {0.f, 0.f, 0.f, 0.f}, is = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) bv = *(const f32x4*)(p.bias + n);
            if (bnb) {
                mu = *(const f32x4*)(p.bnb.mean + n);
                is = *(const f32x4*)(p.bnb.invstd + n);
            }
            if (bnb) {                                // (bias / ReLU / float mask are excluded by the host for this mode)
                // pass 1: the values this thread will store (kept in apre[]) and their share of the two sums; the stores follow the
                // partial row and the ticket (pass 2)
#pragma unroll
                for (int i = 0; i < E_NR; ++i) {
                    const int row = rr + RPP * i;
                    const int tl = row >> 2, px = row & 3;
                    const int ro = rowinfo[2 * tl], fl = rowinfo[2 * tl + 1];
                    const bool dead = ro < 0 || ((px & 1) && !(fl & 1)) || ((px & 2) && !(fl & 2));
                    mpre[i] = dead ? 0u : (mpre[i] | 0x100u);        // bit 8: this row is stored
                    if (dead) continue;
                    f32x4 v = *(const f32x4*)(ct + row * LDC + 4 * cg);
                    if (p.add_src) v += apre[i];
                    f32x4 g = v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] = ((mpre[i] >> e) & 1u) ? g[e] : 0.f;
                    apre[i] = p.bnb.store_masked ? g : v;
                    s1 += g;
                    s2 += g * ((xpre[i] - mu) * is);
                }
            } else if (tail_on) {
                // forward statistics with the in-kernel finalize: pass 1 reads this thread's rows (kept in apre[]) and sums them
                // (plain convolution: no bias / residual / ReLU here, the host excludes them)
#pragma unroll
                for (int i = 0; i < E_NR; ++i) {
                    const int row = rr + RPP * i;
                    const int tl = row >> 2, px = row & 3;
                    const int ro = rowinfo[2 * tl], fl = rowinfo[2 * tl + 1];
                    const bool dead = ro < 0 || ((px & 1) && !(fl & 1)) || ((px & 2) && !(fl & 2));
                    mpre[i] = dead ? 0u : 0x100u;
                    if (dead) continue;
                    const f32x4 v = *(const f32x4*)(ct + row * LDC + 4 * cg);
                    apre[i] = v;
                    s1 += v;
                    s2 += v * v;
                }
            } else
#pragma unroll 4
            for (int row = rr; row < TB * 4; row += RPP) {
                const int tl = row >> 2, px = row & 3;
                const int ro = rowinfo[2 * tl], fl = rowinfo[2 * tl + 1];
                if (ro < 0 || ((px & 1) && !(fl & 1)) || ((px & 2) && !(fl & 2))) continue;
                const size_t o = (size_t)(ro + ((px >> 1) * sg.W + (px & 1)) * p.out_ld) + n;
                f32x4 v = *(const f32x4*)(ct + row * LDC + 4 * cg);
                s1 += v;
                s2 += v * v;
                v += bv;
                if (p.add_src) v += *(const f32x4*)(p.add_src + o);
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (p.mask_src) {
                    const f32x4 m = *(const f32x4*)(p.mask_src + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = m[e] > 0.f ? v[e] : 0.f;
                }
                *(f32x4*)(p.out + o) = v;
            }
        }
        if (p.stats) {                               // fixed-order (deterministic) reduction over the RPP row lanes
            *(f32x4*)(red + rr * BN + 4 * cg) = s1;
            *(f32x4*)(red + (RPP + rr) * BN + 4 * cg) = s2;
            __syncthreads();
            if (tid < BN && n0 + tid < p.N) {
                float a1 = 0.f, a2 = 0.f;
#pragma unroll 8
                for (int r = 0; r < RPP; ++r) {
                    a1 += red[r * BN + tid];
                    a2 += red[(RPP + r) * BN + tid];
                }
                float* o = p.stats + (size_t)mb * 2 * p.N;
                if (tail_on) {                        // write-through: another CU's workgroup reduces the rows inside this launch
                    bn_tail_store(o + n0 + tid, a1);
                    bn_tail_store(o + p.N + n0 + tid, a2);
                } else {
                    o[n0 + tid] = a1;
                    o[p.N + n0 + tid] = a2;
                }
            }
        }
        if ((bnb || tail_on) && n < p.N) {           // pass 2: the output stores
#pragma unroll
            for (int i = 0; i < E_NR; ++i) {
                if (!(mpre[i] & 0x100u)) continue;
                const int row = rr + RPP * i;
                const int tl = row >> 2, px = row & 3;
                const size_t o = (size_t)(rowinfo[2 * tl] + ((px >> 1) * sg.W + (px & 1)) * p.out_ld) + n;
                *(f32x4*)(p.out + o) = apre[i];
            }
        }
        if (tail_on) {
            // in-kernel BatchNorm finalize (bn_tail.h): producers drain and fire their arrival, the column block's last row block reduces
            if (mb != p.m_blocks - 1) {
                if (tid < BN) bn_tail_arrive(p.tail.tickets + nb);
            } else {
                __syncthreads();                     // ct / red are free
                bn_tail_reduce<NT, BN>(p.tail, p.tail.tickets + nb, p.stats, p.m_blocks, p.N, n0, (double*)smem);
            }
        }
        return;
    }
    // scalar path: ragged channel counts (the 45-channel head output) and split-K (fp32 atomics; linear terms only)
    for (int idx = tid; idx < TB * 4 * BN; idx += NT) {
        const int row = idx / BN, col = idx - row * BN;
        const int n = n0 + col;
        const int tl = row >> 2, px = row & 3;
        const int ro = rowinfo[2 * tl], fl = rowinfo[2 * tl + 1];
        if (n >= p.N || ro < 0 || ((px & 1) && !(fl & 1)) || ((px & 2) && !(fl & 2))) continue;
        const size_t o = (size_t)(ro + ((px >> 1) * sg.W + (px & 1)) * p.out_ld) + n;
        float v = ct[row * LDC + col];
        if (p.splits > 1) {
            if (split == 0) {
                if (p.bias) v += p.bias[n];
                if (p.add_src && !p.add_is_out) v += p.add_src[o];
            }
            if (p.mask_src) v = (p.mask_src[o] > 0.f) ? v : 0.f;
            unsafeAtomicAdd(p.out + o, v);
        } else {
            if (p.bias) v += p.bias[n];
            if (p.add_src) v += p.add_src[o];
            if (p.relu) v = fmaxf(v, 0.f);
            if (p.mask_src) v = (p.mask_src[o] > 0.f) ? v : 0.f;
            p.out[o] = v;
        }
    }
}

// ---- weight transform U = G g G^T, all layers of a step in one launch -------------------------------------------------
// job: source rows [N][9 taps][src_tap_ld] with row stride src_row_ld (a channel window of an OHWI weight, or a dgrad weight
// image), flip = 1 rotates the filter by 180 degrees (data gradient).  dst: [chunks][16][Npad][8], p = j*4 + i.
struct WnWJob {
    int64_t src, dst;            // absolute device addresses
    int32_t N, C, src_row_ld, src_tap_ld, flip, Npad, chunks, blk0;
};

__global__ __launch_bounds__(256) void wino_weight_kernel(const WnWJob* jobs, int njobs) {
    int ji = 0;
    for (int s = 1; s < njobs; ++s)
        if ((int)blockIdx.x >= jobs[s].blk0) ji = s;
    const WnWJob jb = jobs[ji];
    const int id = ((int)blockIdx.x - jb.blk0) * 256 + threadIdx.x;      // (chunk, n, cc), cc fastest
    const int cc = id & 7;
    const int n = (id >> 3) % jb.Npad;
    const int chunk = (id >> 3) / jb.Npad;
    if (chunk >= jb.chunks) return;
    const int c = chunk * 8 + cc;
    float gk[3][3];
    const bool ok = (n < jb.N) & (c < jb.C);
    const float* src = (const float*)jb.src + (size_t)n * jb.src_row_ld + c;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int tap = jb.flip ? (2 - a) * 3 + (2 - b) : a * 3 + b;
            gk[a][b] = ok ? src[(size_t)tap * jb.src_tap_ld] : 0.f;
        }
    // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
    float tg[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        tg[0][b] = gk[0][b];
        tg[1][b] = 0.5f * (gk[0][b] + gk[1][b] + gk[2][b]);
        tg[2][b] = 0.5f * (gk[0][b] - gk[1][b] + gk[2][b]);
        tg[3][b] = gk[2][b];
    }
    float* dst = (float*)jb.dst + ((size_t)chunk * 16 * jb.Npad + n) * 8 + cc;
    const size_t ps = (size_t)jb.Npad * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float u0 = tg[i][0];
        const float u1 = 0.5f * (tg[i][0] + tg[i][1] + tg[i][2]);
        const float u2 = 0.5f * (tg[i][0] - tg[i][1] + tg[i][2]);
        const float u3 = tg[i][2];
        const float sg = (i == 3) ? -1.f : 1.f;      // the kernel's input transform produces -V for row 3 (see store_a)
        dst[(0 * 4 + i) * ps] = sg * u0;
        dst[(1 * 4 + i) * ps] = sg * u1;
        dst[(2 * 4 + i) * ps] = sg * u2;
        dst[(3 * 4 + i) * ps] = sg * u3;
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------

extern "C" int64_t zsg_wino_u_elems(int32_t C, int32_t N) {
    const int64_t chunks = (C + WN_CK - 1) / WN_CK, npad = (N + 63) / 64 * 64;
    return chunks * 16 * npad * 8;
}

extern "C" int zsg_wino_weights(const void* jobs_dev, int32_t njobs, int32_t total_blocks, void* stream) {
    ZSG_REQUIRE(jobs_dev && njobs > 0 && total_blocks > 0, "wino_weights: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("wino_weight_kernel", st, 0, 0);
    ZSG_LAUNCH(wino_weight_kernel, dim3(total_blocks), dim3(256), 0, st, (const WnWJob*)jobs_dev, njobs);
    ZSG_CHECK_LAUNCH("wino_weights");
    return 0;
}

template <int TM, int TN, int PS, bool SK = false>
static int wino_launch(const WnParams& p, hipStream_t st, double flops, const char* kname) {
    constexpr int TB = 32 * TM, BN = 32 * TN, NT = 64 * PS * TM * TN;
    constexpr int SA = TB * 8 + 8, SB = BN * 8;
    constexpr size_t stage = (size_t)2 * 16 * (SA + SB) * sizeof(float);
    constexpr size_t epi = ((size_t)TB * 4 * (BN + 4) + 2 * (NT / (BN / 4)) * BN) * sizeof(float);
    static_assert(epi <= stage, "the epilogue's transposed tile + statistics rows reuse the K-loop staging area (rowinfo sits behind it)");
    static_assert(stage + TB * 2 * sizeof(int) <= 160 * 1024, "staging area exceeds a CU's LDS");
    const size_t lds = stage + TB * 2 * sizeof(int);
    static bool attr_done[ZSG_MAX_DEV] = {};       // per device (a benign race sets it twice)
    int dev = 0;
    (void)hipGetDevice(&dev);
    ZSG_REQUIRE(dev >= 0 && dev < ZSG_MAX_DEV, "conv_wino: device %d", dev);
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)wino_kernel<TM, TN, PS, SK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) ZSG_FAIL(-3, "wino: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done[dev] = true;
    }
    ZSG_PROF(kname, st, flops, p.alg_bytes);
    ZSG_LAUNCH((wino_kernel<TM, TN, PS, SK>), dim3(SK ? p.sk_grid : p.m_blocks * p.n_blocks * p.splits), dim3(NT), lds, st, p);
    ZSG_CHECK_LAUNCH("conv_wino");
    return 0;
}
// stream-K launch (profile name = kname + "+sk"): 256 workgroups (these tiles fill a CU's LDS: one workgroup per CU), the partial
// output tiles and flags in the stream's scratch (zsg_set_stream_workspace)
template <int TM, int TN, int PS>
static int wino_launch_sk(WnParams& p, hipStream_t st, double flops, const char* kname) {
    constexpr int TB = 32 * TM, BN = 32 * TN;
    size_t ws_bytes = 0;
    char* ws = (char*)zsg_stream_workspace(st, &ws_bytes);
    ZSG_REQUIRE(ws, "conv_wino: a stream-K tile hint needs zsg_set_stream_workspace() for this stream");
    const int tiles = p.m_blocks * p.n_blocks, grid = ZSG_NUM_CU;
    ZSG_REQUIRE(tiles <= grid, "conv_wino: stream-K is for grids below one round (%d tiles, %d workgroups)", tiles, grid);
    if ((size_t)ZSG_SK_FLAG_BYTES + (size_t)grid * TB * 4 * BN * sizeof(float) > ws_bytes)
        ZSG_FAIL(-2, "conv_wino: stream-K needs %zu bytes of stream workspace (%zu registered)", (size_t)ZSG_SK_FLAG_BYTES + (size_t)grid * TB * 4 * BN * sizeof(float), ws_bytes);
    p.sk_grid = grid;
    p.sk_per = cdiv((int64_t)tiles * p.chunks, grid);
    p.sk_flags = (unsigned*)ws;
    p.sk_ws = (float*)(ws + ZSG_SK_FLAG_BYTES);
    static char nm[96];
    snprintf(nm, sizeof(nm), "%s+sk", kname);
    return wino_launch<TM, TN, PS, true>(p, st, flops, nm);
}
// tile_hint = TB | (BN << 8) | (split_k << 16) | (four position groups << 24), TB (tiles per block) and BN in {32, 64};
// 0 = 64x64, two position groups, no split
static int conv_wino_impl(const zsg_conv_desc* d, const float* src, const float* U, float* out, const float* bias,
                          const float* add_src, const float* mask_src, float* bn_partials, const BnbDev* bnb, void* stream,
                          const BnTail* tail = nullptr) {
    ZSG_REQUIRE(d && src && U && out, "conv_wino: null argument");
    ZSG_REQUIRE(d->nseg >= 1 && d->nseg <= ZSG_MAX_SEG, "conv_wino: nseg=%d", d->nseg);
    ZSG_REQUIRE(d->C > 0 && (d->C % 4) == 0 && (d->src_ld % 4) == 0, "conv_wino: C=%d src_ld=%d must be multiples of 4", d->C, d->src_ld);
    ZSG_REQUIRE(d->wR == 3 && d->wS == 3 && !d->merge_x, "conv_wino: 3x3 filters only");
    int TB = d->tile_hint & 0xff, BN = (d->tile_hint >> 8) & 0xff, splits = (d->tile_hint >> 16) & 0xff;
    if (!d->tile_hint) { TB = 64; BN = 64; }
    if (splits < 1) splits = 1;
    ZSG_REQUIRE((TB == 32 || TB == 64) && (BN == 32 || BN == 64), "conv_wino: unsupported tile %dx%d", TB, BN);
    WnParams p;
    memset(&p, 0, sizeof(p));
    p.src = src; p.U = U; p.out = out; p.bias = bias; p.add_src = add_src; p.mask_src = mask_src; p.stats = bn_partials;
    p.C = d->C; p.N = d->N; p.Npad = (d->N + 63) / 64 * 64; p.src_ld = d->src_ld; p.out_ld = d->out_ld; p.relu = d->relu;
    p.nseg = d->nseg; p.chunks = (d->C + WN_CK - 1) / WN_CK; p.splits = splits;
    p.add_is_out = (add_src == out) ? 1 : 0;
    p.alg_bytes = zsg_conv_alg_bytes(d, add_src != nullptr);
    int blocks = 0;
    double fl = 0;
    bool v = (d->out_ld % 4) == 0 && (d->N % 4) == 0;
    for (int s = 0; s < d->nseg; ++s) {
        const zsg_seg& a = d->seg[s];
        // a centred 3x3 window at unit stride: forward (d0 = -1, step +1) or data gradient (d0 = +1, step -1; U holds
        // the rotated filter)
        ZSG_REQUIRE(a.ty.n == 3 && a.tx.n == 3 && a.sy == 1 && a.sx == 1 && a.osy == 1 && a.osx == 1 && a.opy == 0 && a.opx == 0 &&
                        a.ty.d0 == -a.ty.dstep && a.tx.d0 == -a.tx.dstep && (a.ty.dstep == 1 || a.ty.dstep == -1) && a.tx.dstep == a.ty.dstep,
                    "conv_wino: seg %d is not a 3x3 / stride 1 / pad 1 convolution", s);
        ZSG_REQUIRE(a.rows_y == a.src_H && a.rows_x == a.src_W && a.out_W == a.rows_x, "conv_wino: seg %d: output grid must equal the input grid", s);
        const int64_t tiles = (int64_t)d->B * ((a.src_H + 1) / 2) * ((a.src_W + 1) / 2);
        ZSG_REQUIRE(tiles > 0 && tiles < (1ll << 28), "conv_wino: seg %d tiles=%lld", s, (long long)tiles);
        ZSG_REQUIRE(a.src_off + (int64_t)d->B * a.src_bstride < (1ll << 29) && a.out_off + (int64_t)d->B * a.out_bstride < (1ll << 29),
                    "conv_wino: tensor exceeds 2^29 elements (2 GB window)");
        ZSG_REQUIRE((a.src_off % 4) == 0 && (a.src_bstride % 4) == 0, "conv_wino: seg %d source not 16-byte aligned", s);
        WnSegDev& o = p.seg[s];
        o.tiles_y = (a.src_H + 1) / 2; o.tiles_x = (a.src_W + 1) / 2; o.tiles = (int)tiles; o.blk0 = blocks;
        o.H = a.src_H; o.W = a.src_W;
        o.src_off = (int)a.src_off; o.src_bstride = (int)a.src_bstride; o.out_off = (int)a.out_off; o.out_bstride = (int)a.out_bstride;
        blocks += cdiv(tiles, TB);
        fl += 2.0 * d->B * a.src_H * a.src_W * d->N * 9.0 * d->C;
        v = v && (a.out_off % 4) == 0 && (a.out_bstride % 4) == 0;
    }
    ZSG_REQUIRE((int64_t)p.chunks * 16 * p.Npad * 8 < (1ll << 29), "conv_wino: transformed weights exceed 2^29 elements");
    p.m_blocks = blocks;
    p.n_blocks = cdiv(d->N, BN);
    const uintptr_t al = (uintptr_t)out | (uintptr_t)bias | (uintptr_t)add_src | (uintptr_t)mask_src;
    p.vec = (v && (al & 15) == 0) ? 1 : 0;
    if (bnb) {
        ZSG_REQUIRE(bn_partials && bnb->x && bnb->mean && bnb->invstd, "conv_wino_bnb: null argument");
        ZSG_REQUIRE(splits == 1 && p.vec && !bias && !d->relu && !mask_src, "conv_wino_bnb: needs an unsplit, bias-free convolution with 16-byte addressable output rows");
        ZSG_REQUIRE((((uintptr_t)bnb->x | (uintptr_t)bnb->mean | (uintptr_t)bnb->invstd) & 15) == 0, "conv_wino_bnb: operands not 16-byte aligned");
        p.bnb = *bnb;
        p.bnb.store_masked = d->epi_flags & 1;
    } else if (bn_partials) {
        ZSG_REQUIRE(splits == 1 && !bias && !add_src && !d->relu && p.vec, "conv_wino: BN-statistics fusion needs a plain (bias-free, unsplit, 16-byte addressable) convolution");
    }
    if (tail) {
        ZSG_REQUIRE(tail->tickets && bn_partials && splits == 1 && p.vec && !mask_src && p.m_blocks <= BN_TAIL_MAX_ROWS,
                    "conv_wino: in-kernel BatchNorm finalize needs fused partials, the vectorised epilogue and at most %d row blocks (%d)",
                    BN_TAIL_MAX_ROWS, p.m_blocks);
        ZSG_REQUIRE((((uintptr_t)bn_partials) & 15) == 0 && (d->N % 4) == 0, "conv_wino: partial rows not 16-byte aligned");
        p.tail = *tail;
        int64_t rows = 0;
        for (int s = 0; s < d->nseg; ++s) rows += (int64_t)d->B * d->seg[s].rows_y * d->seg[s].rows_x;
        p.tail.rows = rows;
    }
    hipStream_t st = (hipStream_t)stream;
    if (splits > 1) {
        ZSG_REQUIRE(!d->relu && d->nseg == 1 && d->out_ld == d->N && d->seg[0].out_bstride == (int64_t)d->seg[0].rows_y * d->seg[0].rows_x * d->N,
                    "conv_wino: split-K needs a single dense segment without ReLU");
        if (splits > p.chunks) p.splits = splits = p.chunks;
        if (!p.add_is_out) {
            hipError_t e = hipMemsetAsync(out + d->seg[0].out_off, 0, (size_t)d->B * d->seg[0].out_bstride * sizeof(float), st);
            if (e != hipSuccess) ZSG_FAIL(-3, "conv_wino: memset: %s", hipGetErrorString(e));
        }
    }
    if (d->tile_hint && ((d->tile_hint >> 28) & 3)) {      // stream-K (one workgroup per CU whatever the field says: these tiles fill a CU's LDS)
        ZSG_REQUIRE(splits == 1 && p.vec && d->nseg == 1, "conv_wino: stream-K needs one segment, no split-K, 16-byte addressable output rows");
        const bool ps4 = (d->tile_hint >> 24) & 1;
        // (only the 32-tile x 64-channel, four-group tile: inside the pass loop the 64-tile kernels spill — 96 registers at the four-group
        // tile's 128-register budget, 97 at the two-group tile's 256 — and a spilling instantiation is not shipped, tools/kernel_resources.py)
        if (TB == 32 && BN == 64 && ps4) return wino_launch_sk<1, 2, 4>(p, st, fl, "wino_kernel<1, 2, 4>");
        ZSG_FAIL(-1, "conv_wino: no stream-K variant for tile %dx%d (four position groups %d)", TB, BN, (int)ps4);
    }
    if ((d->tile_hint >> 24) & 1) {            // four position groups: twice the waves per SIMD
        if (TB == 64 && BN == 64) return wino_launch<2, 2, 4>(p, st, fl, "wino_kernel<2, 2, 4>");
        if (TB == 32 && BN == 64) return wino_launch<1, 2, 4>(p, st, fl, "wino_kernel<1, 2, 4>");
        if (TB == 64 && BN == 32) return wino_launch<2, 1, 4>(p, st, fl, "wino_kernel<2, 1, 4>");
        return wino_launch<1, 1, 4>(p, st, fl, "wino_kernel<1, 1, 4>");
    }
    if (TB == 64 && BN == 64) return wino_launch<2, 2, 2>(p, st, fl, "wino_kernel<2, 2, 2>");
    if (TB == 32 && BN == 64) return wino_launch<1, 2, 2>(p, st, fl, "wino_kernel<1, 2, 2>");
    if (TB == 64 && BN == 32) return wino_launch<2, 1, 2>(p, st, fl, "wino_kernel<2, 1, 2>");
    return wino_launch<1, 1, 2>(p, st, fl, "wino_kernel<1, 1, 2>");
}

extern "C" int zsg_conv_wino(const zsg_conv_desc* d, const float* src, const float* U, float* out, const float* bias,
                             const float* add_src, const float* mask_src, float* bn_partials, void* stream) {
    return conv_wino_impl(d, src, U, out, bias, add_src, mask_src, bn_partials, nullptr, stream);
}

// see zsg_conv_igemm_bnb: the data gradient that completes a BatchNorm's dout also emits that BatchNorm's backward partials
extern "C" int zsg_conv_wino_bnb(const zsg_conv_desc* d, const float* src, const float* U, float* out, const float* add_src,
                                 const float* bn_x, const float* bn_mean, const float* bn_invstd, const uint8_t* bn_relu_mask,
                                 float* partials, void* stream) {
    BnbDev b = {bn_x, bn_mean, bn_invstd, bn_relu_mask};
    return conv_wino_impl(d, src, U, out, nullptr, add_src, nullptr, partials, &b, stream);
}

// The Winograd counterparts of zsg_conv_igemm_bnstat / zsg_conv_igemm_bnb_tail (igemm.hip): the last block of a column block finalises
// the BatchNorm statistics / backward sums inside the launch (bn_tail.h); tickets: zsg_conv_bn_tail_tickets(d, 1) zeroed words.
extern "C" int zsg_conv_wino_bnstat(const zsg_conv_desc* d, const float* src, const float* U, float* out, float* partials, uint32_t* tickets,
                                    float* mean, float* invstd, float* running_mean, float* running_var, float momentum, float eps, void* stream) {
    ZSG_REQUIRE(tickets && mean && invstd && partials, "conv_wino_bnstat: null argument");
    BnTail t;
    memset(&t, 0, sizeof(t));
    t.tickets = tickets; t.mean = mean; t.invstd = invstd; t.rmean = running_mean; t.rvar = running_var; t.momentum = momentum; t.eps = eps; t.mode = 0;
    return conv_wino_impl(d, src, U, out, nullptr, nullptr, nullptr, partials, nullptr, stream, &t);
}
extern "C" int zsg_conv_wino_bnb_tail(const zsg_conv_desc* d, const float* src, const float* U, float* out, const float* add_src,
                                      const float* bn_x, const float* bn_mean, const float* bn_invstd, const uint8_t* bn_relu_mask,
                                      float* partials, uint32_t* tickets, float* coef, float* dgamma, float* dbeta, int32_t accumulate,
                                      void* stream) {
    ZSG_REQUIRE(tickets && coef && partials, "conv_wino_bnb_tail: null argument");
    BnbDev b = {bn_x, bn_mean, bn_invstd, bn_relu_mask};
    BnTail t;
    memset(&t, 0, sizeof(t));
    t.tickets = tickets; t.coef = coef; t.dgamma = dgamma; t.dbeta = dbeta; t.accumulate = accumulate ? 1 : 0; t.mode = 1;
    return conv_wino_impl(d, src, U, out, nullptr, add_src, nullptr, partials, &b, stream, &t);
}

