// wino4.hip — Winograd F(4x4, 3x3) convolution (forward and stride-1 data gradient) on fp32 MFMA, NHWC, gfx950 (round 5).
//
// For the 3x3 / stride 1 / pad 1 convolutions BEHIND the last BatchNorm of the network — the pyramid's output convolutions and the
// shared head (fpn_resnet.py:157-172, mdl.py:211-244) — where F(4x4,3x3)'s larger rounding error stays a rounding error: end to end
// 1.2e-4 on outputs of magnitude 30 at the benchmark shape (tools/wino_f4_e2e.py, profiles/r05_wino_f4_gate.txt; the same algorithm on
// layer1's conv2, in front of 49 train-mode BatchNorms, moves the outputs by 2.4e-3 and is therefore never offered there).
//
//   Y(4x4) = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A        d: 6x6 input patch, g: 3x3 filter
//
// Interpolation points 0, 1, -1, 1/2, -2, inf (the mixed set of Barabasz et al.: max error / max|y| 3.4e-6 on head-like operands
// against 6.5e-6 for the textbook 0, +-1, +-2 — tools/wino_f4_error.py; on the GPU at C = 256, 38 x 38: 1.08e-5 with the textbook
// points, which missed the 1e-5 gate of VERDICT r04 item 7; the 1-D input transform costs 16 instead of 12 operations).
// 36 multiplies per 16 outputs = 2.25 per pixel and (cin, cout) pair instead of 4 (F(2x2,3x3)) or 9 (direct).
// GEMM view: 36 independent GEMMs ("positions" p = i*6 + j) sharing M = 4x4 output tiles, N = output channels, K = input channels.
// A workgroup owns 32 tiles x 64 channels; its twelve waves are (n half) x (row i of the 6x6 position grid): six 32x32 accumulators
// (96 registers) each, three waves per SIMD.  One K chunk = 8 channels.
//   B operand (transformed filter): every (position, n half) belongs to exactly ONE wave, so the fragments go from global memory
//     (L2: the image is shared by all row blocks) straight into registers — U is stored [chunk][position][n][8], a lane's 16 bytes ARE
//     its MFMA fragment — one chunk ahead, refilled in place behind the MFMAs that consumed them.  No LDS, no barrier for B.
//   A operand: every lane owns (tile, patch row, channel PAIR): six 8-byte loads per chunk (12 registers), requested one interval + one
//     chunk ahead; chunk k+1 is transformed while chunk k multiplies:
//     interval k: MFMAs of chunk k | COLUMN transform B^T (.) of chunk k+1 in place | ROW transform d B of chunk k+2 (registers -> a
//     third A stage) | request the patch rows of chunk k+3                                                                  | ONE barrier
//     (all 768 lanes transform: lane = (tile, channel pair, patch row or column); on gfx950 nothing co-issues with a SIMD's fp32 MFMA
//     stream — tools/ubench/mfma_coissue.hip — so the transform's VALU time ADDS to the MFMA time on each SIMD; the interleave only
//     keeps every SIMD supplied with both kinds of work between two barriers.)
// Epilogue: every wave contracts its row of M along j (z = M[i][.] A), the six rows are combined along i through LDS in a fixed order
// (Y[a][.] += A^T[a][i] z), then the same 16-byte store pass as wino.hip (bias, residual / accumulate, ReLU, ReLU mask).
// History of the round (profiles/r05_wino4_microbench.txt): a first, strictly phased version with the filter slice staged through a
// single LDS stage ran 5 us per chunk (177 us for P3_2 against 140 us for F(2x2,3x3)); patch rows by LDS-DMA made hipcc drain every
// register load at the top of each chunk (vmcnt(0)), and hand-counted inline-asm loads around that cost 70 spilled registers.
#include "common.h"
#include "zsg_wino4.h"      // (the experiment's own declarations: not part of include/zsg.h)

#define W4_CK 8
#define W4_TB 32
#define W4_BN 64
#define W4_NT 768
#define W4_SA (W4_TB * 8 + 8)      // floats between positions of an A stage
#define W4_LDC (W4_BN + 4)

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct W4SegDev {
    int tiles_y, tiles_x, tiles;   // 4x4-tile grid per image; tiles = B * tiles_y * tiles_x
    int blk0;                      // first M block of the segment
    int H, W;
    int src_off, src_bstride, out_off, out_bstride;   // elements
};

struct W4Params {
    const float* src;
    const float* U;
    float* out;
    const float* bias;
    const float* add_src;
    const float* mask_src;
    int C, N, Npad, src_ld, out_ld, relu, nseg;
    int m_blocks, n_blocks, chunks, vec;
    double alg_bytes;    // host only
    W4SegDev seg[ZSG_MAX_SEG];
};

// 1-D input transform B^T d, points (0, 1, -1, 1/2, -2, inf):
//   [1 -3/2 -2 3/2 1 0], [0 -1 1/2 5/2 1 0], [0 1 -5/2 1/2 1 0], [0 -2 -1 2 1 0], [0 1/2 -1 -1/2 1 0], [0 1 -3/2 -2 3/2 1]
// every output is handed to ST(index, value) as soon as it exists (few live registers: the kernel runs at its register limit)
#define W4_BT_EMIT(d, ST)                                               \
    {                                                                   \
        const f32x2 e__ = (d)[3] - (d)[1];                              \
        const f32x2 c__ = (d)[4] - (d)[2];                              \
        ST(0, ((d)[0] + (d)[4]) + 1.5f * e__ - 2.f * (d)[2]);           \
        ST(5, ((d)[1] + (d)[5]) + 1.5f * c__ - 2.f * (d)[3]);           \
        ST(3, c__ + 2.f * e__);                                         \
        ST(4, c__ - 0.5f * e__);                                        \
        ST(1, ((d)[4] - (d)[1]) + 0.5f * (d)[2] + 2.5f * (d)[3]);       \
        ST(2, ((d)[4] + (d)[1]) - 2.5f * (d)[2] + 0.5f * (d)[3]);       \
    }

// 1-D output transform A^T m (rows of A^T: [1 1 1 1 1 0], [0 1 -1 1/2 -2 0], [0 1 1 1/4 4 0], [0 1 -1 1/8 -8 1])
__device__ __forceinline__ void w4_at(float m0, float m1, float m2, float m3, float m4, float m5, float (&z)[4]) {
    const float s12 = m1 + m2, d12 = m1 - m2;
    z[0] = m0 + s12 + m3 + m4;
    z[1] = d12 + 0.5f * m3 - 2.f * m4;
    z[2] = s12 + 0.25f * m3 + 4.f * m4;
    z[3] = d12 + 0.125f * m3 - 8.f * m4 + m5;
}

__global__ __launch_bounds__(W4_NT, 3) void wino4_kernel(const W4Params p) {
    constexpr int NT = W4_NT, TB = W4_TB, BN = W4_BN, SA = W4_SA, LDC = W4_LDC;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                             // [3][36][SA]: three A stages (multiply | column-transform | row-transform)
    constexpr int STAGE = 3 * 36 * SA;
    constexpr int EPI = TB * 16 * LDC;            // the epilogue's transposed tile [TB*16][LDC] reuses the stages
    int* rowinfo = (int*)(smem + (EPI > STAGE ? EPI : STAGE));   // [TB][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);      // (wave-uniform values the compiler can keep in SGPRs)
    const int wn = wave_u & 1, wi = wave_u >> 1;  // n half; row i of the 6x6 position grid
    const int li = lane & 31, lh = lane >> 5;

    const int bid = xcd_remap(blockIdx.x, p.m_blocks * p.n_blocks);
    const int mb = bid / p.n_blocks, nb = bid % p.n_blocks;      // (channel-block-major order — an XCD's chunk shares one filter slice — measured the same)
    int si = 0;
#pragma unroll
    for (int s = 1; s < ZSG_MAX_SEG; ++s)
        if (s < p.nseg && mb >= p.seg[s].blk0) si = s;
    const W4SegDev sg = p.seg[si];
    const int m0 = (mb - sg.blk0) * TB;
    const int n0 = nb * BN;

    // ---- transform-lane state (all 768 lanes): (tile tt, patch row / column tq, channel quad tg, half th), th fastest: the four lanes of
    // a pixel's chunk read 32 contiguous bytes
    const int th = tid & 1, tg = (tid >> 1) & 1, tq = (tid >> 2) % 6, tt = (tid >> 2) / 6;
    unsigned a_base = 0, a_ok = 0;           // byte offset of the patch row's first pixel (this lane's channel pair); bit c: pixel c is inside the image
    const bool c_tail = (p.C & 4) != 0;      // the last chunk of a C % 8 == 4 tensor: its upper channel quad is dead
    {
        const int m = m0 + tt;
        const bool ok = m < sg.tiles;
        const int mm = ok ? m : 0;
        const int per = sg.tiles_y * sg.tiles_x;
        const int b = mm / per;
        const int rem = mm - b * per;
        const int ty = rem / sg.tiles_x;
        const int tx = rem - ty * sg.tiles_x;
        const int y = 4 * ty - 1 + tq, x0 = 4 * tx - 1;
        const bool rok = ok & ((unsigned)y < (unsigned)sg.H);
        a_base = 4u * (unsigned)(sg.src_off + b * sg.src_bstride + (y * sg.W + x0) * p.src_ld + 4 * tg + 2 * th);      // (only used where the pixel's bit is set)
#pragma unroll
        for (int c = 0; c < 6; ++c) a_ok |= (rok & ((unsigned)(x0 + c) < (unsigned)sg.W)) ? (1u << c) : 0u;
        if (tq == 0 && tg == 0 && th == 0) {
            rowinfo[2 * tt] = ok ? sg.out_off + b * sg.out_bstride + ((4 * ty) * sg.W + 4 * tx) * p.out_ld : -1;
            rowinfo[2 * tt + 1] = min(4, sg.W - 4 * tx) | (min(4, sg.H - 4 * ty) << 8);      // valid columns | valid rows
        }
    }
    const int a_pix = 4 * p.src_ld;               // bytes between the pixels of a patch row
    const int t_slot = tt * 8 + 4 * (tg ^ ((tt >> 3) & 1)) + 2 * th;    // + position * SA (16-byte halves of a row XOR-swizzled by bit 3 of the
                                                                        //   tile: conflict-free ds_read_b128 fragments without padding)
    // ---- B fragments: this wave's six positions wi*6 + pl, its 32 channels, k = 4 lh .. 4 lh + 3 of the chunk --------------------------
    const unsigned b_voff = 4u * (unsigned)((n0 + wn * 32 + li) * 8 + 4 * lh);
    const int b_pos = p.Npad * 8 * 4;             // bytes between positions
    const int b_chunk = 36 * b_pos;               // bytes between chunks
    const rsrc_t rsrc_a = make_rsrc(p.src);
    const rsrc_t rsrc_b = make_rsrc(p.U);
    const int nc = p.chunks;

    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    // (macros, not lambdas: arrays captured by reference in a lambda next to barriers end up in scratch memory with hipcc)
    // this lane's six pixels (two channels each) of chunk c; out-of-image pixels / the dead quad of a channel tail read as zeros
    // (live == false — past the last chunk: every lane gets an out-of-range offset, the loads still issue and return zeros without
    //  touching memory; the K loop then has no branch around its loads and hipcc counts the outstanding ones exactly — with branches it
    //  merged the states pessimistically and drained everything, vmcnt(0), once per chunk)
#define W4_LOAD_A(c, live)                                                                                                       \
    {                                                                                                                            \
        const int so__ = (c) * (W4_CK * 4);                                                                                     \
        const bool dead__ = !(live) | (c_tail & ((c) == nc - 1) & (tg != 0));                                                    \
        _Pragma("unroll") for (int col = 0; col < 6; ++col)                                                                     \
            ra[col] = __builtin_bit_cast(f32x2, (u32x2)__builtin_amdgcn_raw_buffer_load_b64(                                      \
                rsrc_a, (int)((!dead__ && ((a_ok >> col) & 1u)) ? a_base + (unsigned)(col * a_pix) : ZSG_OOB), so__, 0));          \
    }
#define W4_LOAD_B(c, pl, live) rb[pl] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rsrc_b, (int)((live) ? b_voff : ZSG_OOB), (c) * b_chunk + (wi * 6 + (pl)) * b_pos, 0))
    // row transform d B of this lane's patch row tq (two channels), from registers into stage `st`
#define W4_ST_ROW(c, v) *(f32x2*)(st__ + (tq * 6 + (c)) * SA + t_slot) = (v);
#define W4_ROWS(stage)                                                                                                           \
    {                                                                                                                            \
        float* st__ = As + (stage) * 36 * SA;                                                                                    \
        W4_BT_EMIT(ra, W4_ST_ROW)                                                                                                \
    }
    // column transform B^T (.) in place (this lane: patch column tq)
#define W4_ST_COL(i, v) *(f32x2*)(st__ + ((i) * 6 + tq) * SA + t_slot) = (v);
#define W4_COLUMNS(stage)                                                                                                        \
    {                                                                                                                            \
        float* st__ = As + (stage) * 36 * SA;                                                                                    \
        f32x2 d__[6];                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < 6; ++i) d__[i] = *(const f32x2*)(st__ + (i * 6 + tq) * SA + t_slot);               \
        W4_BT_EMIT(d__, W4_ST_COL)                                                                                               \
    }
#define W4_MFMA(stage, pl, refill_chunk, refill)                                                                                 \
    {                                                                                                                            \
        const f32x4 fa__ = *(const f32x4*)(As + ((stage) * 36 + wi * 6 + (pl)) * SA + frag_a);                                   \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) acc[pl] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa__[e], rb[pl][e], acc[pl], 0, 0, 0); \
        W4_LOAD_B(refill_chunk, pl, refill);                                                                                     \
    }

    f32x4 rb[6];
    f32x2 ra[6];
    f32x16 acc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    const int frag_a = li * 8 + 4 * (lh ^ ((li >> 3) & 1));
    // Three A stages rotate: in interval k the waves multiply chunk k (stage k % 3), finish chunk k+1 (column transform in place,
    // stage (k+1) % 3: its rows were written in interval k-1) and start chunk k+2 (row transform into stage (k+2) % 3) — ONE barrier per
    // chunk (the two-stage form needed two: 3.9 us per chunk with all global loads ablated, against 1.9 us of MFMA issue).
    if (nc > 0) {
        W4_LOAD_A(0, true);
#pragma unroll
        for (int pl = 0; pl < 6; ++pl) W4_LOAD_B(0, pl, true);
        W4_ROWS(0);
        W4_LOAD_A(1, nc > 1);
        __syncthreads();
        W4_COLUMNS(0);
        if (nc > 1) W4_ROWS(1);
        W4_LOAD_A(2, nc > 2);
    }
    __syncthreads();

    int cur = 0, nx1 = 1, nx2 = 2;
    for (int it = 0; it < nc; ++it) {
        const bool more = it + 1 < nc;             // wave-uniform
        W4_MFMA(cur, 0, it + 1, more);
        W4_MFMA(cur, 1, it + 1, more);
        W4_MFMA(cur, 2, it + 1, more);
        if (more) W4_COLUMNS(nx1);                 // chunk it+1: rows -> full transform (in place)
        W4_MFMA(cur, 3, it + 1, more);
        W4_MFMA(cur, 4, it + 1, more);
        W4_MFMA(cur, 5, it + 1, more);
        if (it + 2 < nc) W4_ROWS(nx2);             // chunk it+2 (requested one chunk ago)
        W4_LOAD_A(it + 3, it + 3 < nc);
        __syncthreads();
        const int t__ = cur;
        cur = nx1; nx1 = nx2; nx2 = t__;
    }
#undef W4_LOAD_A
#undef W4_LOAD_B
#undef W4_ROWS
#undef W4_COLUMNS
#undef W4_ST_ROW
#undef W4_ST_COL
#undef W4_MFMA

    // ---- output transform: z = M[i][.] A in registers, then Y[a][.] += A^T[a][i] z over the six rows through LDS ----------------------
    // A^T columns: i = 0: (1,0,0,0); 1: (1,1,1,1); 2: (1,-1,1,-1); 3: (1,1/2,1/4,1/8); 4: (1,-2,4,-8); 5: (0,0,0,1).  Rounds (fixed order:
    // deterministic): row 1 stores all sixteen pixels, rows 2, 3, 4 add to all, rows 0 and 5 add to their one output row (a = 0 / a = 3).
    float* ct = smem;                               // [TB*16][LDC]
#pragma unroll 1
    for (int rnd = 0; rnd < 5; ++rnd) {
        const bool mine = (rnd < 4) ? (wi == rnd + 1) : (wi == 0 || wi == 5);
        if (mine) {
            const float c1 = (wi == 3) ? 0.5f : ((wi == 4) ? -2.f : ((wi == 2) ? -1.f : 1.f));     // A^T[1][i]; A^T[2][i] = c1^2, A^T[3][i] = c1^3
            const float c2 = c1 * c1, c3 = c2 * c1;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float z[4];
                w4_at(acc[0][e], acc[1][e], acc[2][e], acc[3][e], acc[4][e], acc[5][e], z);
                const int tl = (e & 3) + 8 * (e >> 2) + 4 * lh;
                float* o = ct + (tl * 16) * LDC + wn * 32 + li;
                if (rnd == 0) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        o[(0 + b) * LDC] = z[b];
                        o[(4 + b) * LDC] = z[b];
                        o[(8 + b) * LDC] = z[b];
                        o[(12 + b) * LDC] = z[b];
                    }
                } else if (rnd < 4) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        o[(0 + b) * LDC] += z[b];
                        o[(4 + b) * LDC] += c1 * z[b];
                        o[(8 + b) * LDC] += c2 * z[b];
                        o[(12 + b) * LDC] += c3 * z[b];
                    }
                } else {
                    const int a = (wi == 0) ? 0 : 3;
#pragma unroll
                    for (int b = 0; b < 4; ++b) o[(a * 4 + b) * LDC] += z[b];
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: bias, residual / accumulate, relu, relu-mask ------------------------------------------------------------------
    if (p.vec) {
        constexpr int CG = BN / 4, RPP = NT / CG;
        const int cg = tid % CG, rr = tid / CG;
        const int n = n0 + 4 * cg;
        if (n < p.N) {
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) bv = *(const f32x4*)(p.bias + n);
#pragma unroll 4
            for (int row = rr; row < TB * 16; row += RPP) {
                const int tl = row >> 4, a = (row >> 2) & 3, b = row & 3;
                const int ro = rowinfo[2 * tl], fl = rowinfo[2 * tl + 1];
                if (ro < 0 || b >= (fl & 0xff) || a >= (fl >> 8)) continue;
                const size_t o = (size_t)(ro + (a * sg.W + b) * p.out_ld) + n;
                f32x4 v = *(const f32x4*)(ct + row * LDC + 4 * cg) + bv;
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
        return;
    }
    // scalar path: ragged channel counts (the 45-channel head output)
    for (int idx = tid; idx < TB * 16 * BN; idx += NT) {
        const int row = idx / BN, col = idx - row * BN;
        const int n = n0 + col;
        const int tl = row >> 4, a = (row >> 2) & 3, b = row & 3;
        const int ro = rowinfo[2 * tl], fl = rowinfo[2 * tl + 1];
        if (n >= p.N || ro < 0 || b >= (fl & 0xff) || a >= (fl >> 8)) continue;
        const size_t o = (size_t)(ro + (a * sg.W + b) * p.out_ld) + n;
        float v = ct[row * LDC + col];
        if (p.bias) v += p.bias[n];
        if (p.add_src) v += p.add_src[o];
        if (p.relu) v = fmaxf(v, 0.f);
        if (p.mask_src) v = (p.mask_src[o] > 0.f) ? v : 0.f;
        p.out[o] = v;
    }
}

// ---- weight transform U = G g G^T (6x6), all layers of a step in one launch --------------------------------------------------------
// Same job record as wino.hip's zsg_wino_weights (source rows [N][9 taps][src_tap_ld], flip = 1 rotates the filter by 180 degrees for the
// data gradient); dst: [chunks][36][Npad][8], position p = i*6 + j.
struct W4WJob {
    int64_t src, dst;            // absolute device addresses
    int32_t N, C, src_row_ld, src_tap_ld, flip, Npad, chunks, blk0;
};

__global__ __launch_bounds__(256) void wino4_weight_kernel(const W4WJob* jobs, int njobs) {
    int ji = 0;
    for (int s = 1; s < njobs; ++s)
        if ((int)blockIdx.x >= jobs[s].blk0) ji = s;
    const W4WJob jb = jobs[ji];
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
    // G (points 0, 1, -1, 1/2, -2, inf) = [[1,0,0],[1/3,1/3,1/3],[-1/3,1/3,-1/3],[-16/15,-8/15,-4/15],[1/15,-2/15,4/15],[0,0,1]]
    float tg[6][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const float g0 = gk[0][b], g1 = gk[1][b], g2 = gk[2][b];
        tg[0][b] = g0;
        tg[1][b] = (1.f / 3.f) * (g0 + g1 + g2);
        tg[2][b] = (-1.f / 3.f) * (g0 - g1 + g2);
        tg[3][b] = (-16.f / 15.f) * g0 + (-8.f / 15.f) * g1 + (-4.f / 15.f) * g2;
        tg[4][b] = (1.f / 15.f) * g0 + (-2.f / 15.f) * g1 + (4.f / 15.f) * g2;
        tg[5][b] = g2;
    }
    float* dst = (float*)jb.dst + ((size_t)chunk * 36 * jb.Npad + n) * 8 + cc;
    const size_t ps = (size_t)jb.Npad * 8;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const float g0 = tg[i][0], g1 = tg[i][1], g2 = tg[i][2];
        dst[(i * 6 + 0) * ps] = g0;
        dst[(i * 6 + 1) * ps] = (1.f / 3.f) * (g0 + g1 + g2);
        dst[(i * 6 + 2) * ps] = (-1.f / 3.f) * (g0 - g1 + g2);
        dst[(i * 6 + 3) * ps] = (-16.f / 15.f) * g0 + (-8.f / 15.f) * g1 + (-4.f / 15.f) * g2;
        dst[(i * 6 + 4) * ps] = (1.f / 15.f) * g0 + (-2.f / 15.f) * g1 + (4.f / 15.f) * g2;
        dst[(i * 6 + 5) * ps] = g2;
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------

extern "C" int64_t zsg_wino4_u_elems(int32_t C, int32_t N) {
    const int64_t chunks = (C + W4_CK - 1) / W4_CK, npad = (N + 63) / 64 * 64;
    return chunks * 36 * npad * 8;
}

extern "C" int zsg_wino4_weights(const void* jobs_dev, int32_t njobs, int32_t total_blocks, void* stream) {
    ZSG_REQUIRE(jobs_dev && njobs > 0 && total_blocks > 0, "wino4_weights: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("wino4_weight_kernel", st, 0, 0);
    ZSG_LAUNCH(wino4_weight_kernel, dim3(total_blocks), dim3(256), 0, st, (const W4WJob*)jobs_dev, njobs);
    ZSG_CHECK_LAUNCH("wino4_weights");
    return 0;
}

extern "C" int zsg_conv_wino4(const zsg_conv_desc* d, const float* src, const float* U, float* out, const float* bias, const float* add_src,
                              const float* mask_src, void* stream) {
    ZSG_REQUIRE(d && src && U && out, "conv_wino4: null argument");
    ZSG_REQUIRE(d->nseg >= 1 && d->nseg <= ZSG_MAX_SEG, "conv_wino4: nseg=%d", d->nseg);
    ZSG_REQUIRE(d->C > 0 && (d->C % 4) == 0 && (d->src_ld % 4) == 0, "conv_wino4: C=%d src_ld=%d must be multiples of 4", d->C, d->src_ld);
    ZSG_REQUIRE(d->wR == 3 && d->wS == 3 && !d->merge_x, "conv_wino4: 3x3 filters only");
    ZSG_REQUIRE(((d->tile_hint >> 16) & 0xff) <= 1, "conv_wino4: no split-K variant");
    W4Params p;
    memset(&p, 0, sizeof(p));
    p.src = src; p.U = U; p.out = out; p.bias = bias; p.add_src = add_src; p.mask_src = mask_src;
    p.C = d->C; p.N = d->N; p.Npad = (d->N + 63) / 64 * 64; p.src_ld = d->src_ld; p.out_ld = d->out_ld; p.relu = d->relu;
    p.nseg = d->nseg; p.chunks = (d->C + W4_CK - 1) / W4_CK;
    p.alg_bytes = zsg_conv_alg_bytes(d, add_src != nullptr);
    int blocks = 0;
    double fl = 0;
    bool v = (d->out_ld % 4) == 0 && (d->N % 4) == 0;
    for (int s = 0; s < d->nseg; ++s) {
        const zsg_seg& a = d->seg[s];
        ZSG_REQUIRE(a.ty.n == 3 && a.tx.n == 3 && a.sy == 1 && a.sx == 1 && a.osy == 1 && a.osx == 1 && a.opy == 0 && a.opx == 0 &&
                        a.ty.d0 == -a.ty.dstep && a.tx.d0 == -a.tx.dstep && (a.ty.dstep == 1 || a.ty.dstep == -1) && a.tx.dstep == a.ty.dstep,
                    "conv_wino4: seg %d is not a 3x3 / stride 1 / pad 1 convolution", s);
        ZSG_REQUIRE(a.rows_y == a.src_H && a.rows_x == a.src_W && a.out_W == a.rows_x, "conv_wino4: seg %d: output grid must equal the input grid", s);
        const int64_t tiles = (int64_t)d->B * ((a.src_H + 3) / 4) * ((a.src_W + 3) / 4);
        ZSG_REQUIRE(tiles > 0 && tiles < (1ll << 28), "conv_wino4: seg %d tiles=%lld", s, (long long)tiles);
        ZSG_REQUIRE(a.src_off + (int64_t)d->B * a.src_bstride < (1ll << 29) && a.out_off + (int64_t)d->B * a.out_bstride < (1ll << 29),
                    "conv_wino4: tensor exceeds 2^29 elements (2 GB window)");
        ZSG_REQUIRE((a.src_off % 4) == 0 && (a.src_bstride % 4) == 0, "conv_wino4: seg %d source not 16-byte aligned", s);
        W4SegDev& o = p.seg[s];
        o.tiles_y = (a.src_H + 3) / 4; o.tiles_x = (a.src_W + 3) / 4; o.tiles = (int)tiles; o.blk0 = blocks;
        o.H = a.src_H; o.W = a.src_W;
        o.src_off = (int)a.src_off; o.src_bstride = (int)a.src_bstride; o.out_off = (int)a.out_off; o.out_bstride = (int)a.out_bstride;
        blocks += cdiv(tiles, W4_TB);
        fl += 2.0 * d->B * a.src_H * a.src_W * d->N * 9.0 * d->C;
        v = v && (a.out_off % 4) == 0 && (a.out_bstride % 4) == 0;
    }
    ZSG_REQUIRE((int64_t)p.chunks * 36 * p.Npad * 8 < (1ll << 29), "conv_wino4: transformed weights exceed 2^29 elements");
    p.m_blocks = blocks;
    p.n_blocks = cdiv(d->N, W4_BN);
    const uintptr_t al = (uintptr_t)out | (uintptr_t)bias | (uintptr_t)add_src | (uintptr_t)mask_src;
    p.vec = (v && (al & 15) == 0) ? 1 : 0;
    constexpr size_t stage = (size_t)3 * 36 * W4_SA * sizeof(float);
    constexpr size_t epi = (size_t)W4_TB * 16 * W4_LDC * sizeof(float);
    constexpr size_t lds = (stage > epi ? stage : epi) + W4_TB * 2 * sizeof(int);
    static_assert(lds <= 160 * 1024, "stage / epilogue tile exceed a CU's LDS");
    static bool attr_done[ZSG_MAX_DEV] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    ZSG_REQUIRE(dev >= 0 && dev < ZSG_MAX_DEV, "conv_wino4: device %d", dev);
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)wino4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) ZSG_FAIL(-3, "wino4: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done[dev] = true;
    }
    hipStream_t st = (hipStream_t)stream;
    ZSG_PROF("wino4_kernel", st, fl, p.alg_bytes);
    ZSG_LAUNCH(wino4_kernel, dim3(p.m_blocks * p.n_blocks), dim3(W4_NT), lds, st, p);
    ZSG_CHECK_LAUNCH("conv_wino4");
    return 0;
}
