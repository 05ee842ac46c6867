// wgrad.hip — convolution weight gradient on fp32 MFMA (split-K over pixels, fp32 atomic accumulation), gfx950.
//
// GEMM view:  dW[n][col] += sum_rows dY[row][n] * Src[gather(row, tap(col))][c(col)]
//   n    = output channel (rows of dW)                     -> M of the GEMM
//   col  = (tap, c) flattened tap-major                    -> N of the GEMM  (weight row layout, OHWI)
//   rows = (segment, b, y, x) pixels of dY                 -> K of the GEMM, BK = 16 pixels per tile
// Both operands are stored pixel-major with the GEMM M/N index contiguous ("MN-contiguous"), so LDS tiles are
// [k][m] and a lane reads TM (TN) consecutive m (n) values of its k row with one ds_read_b32/b64: MFMA sub-tile t
// covers rows m = TM*i + t — a row permutation that the epilogue undoes.  K tiles never straddle a segment, so the
// pixel -> (b,y,x) decode uses wave-uniform geometry.  Split-K: every block stores its partial 128x128 tile with plain
// coalesced stores into a workspace slab [split][n][col]; a second (HBM-bound, tiny) kernel sums the slabs in a fixed
// order and writes dW — deterministic, and ~3x cheaper than fp32 atomics, which cost more than the K loop they
// follow on the 1x1 layers (measured: 34 -> 88 TFLOP/s on 1024->256 @19^2 with the atomics removed).
#include <stdlib.h>

#include "wgrad_common.h"

// Fixed summation order (deterministic).  A block of 256 threads covers 256/KL float4 elements with KL "split lanes" each
// (lane l sums slabs l, l+KL, ...), then an LDS tree over the lanes, so many-split launches (small weights, huge pixel counts)
// are not one serial latency chain per element.
template <int KL>
__device__ __forceinline__ void wgrad_reduce_body(const WgReduceJob& p, int block, f32x4* sm_) {
    constexpr int EL = 256 / KL;
    f32x4 (*sm)[EL] = (f32x4 (*)[EL])sm_;
    const int q4 = p.ncols / 4;
    const int64_t total = (int64_t)p.N * q4;
    const size_t slab = (size_t)p.N * p.ncols;
    const int el = threadIdx.x % EL, kl = threadIdx.x / EL;
    const int64_t i = (int64_t)block * EL + el;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int n = 0, q = 0;
    if (i < total) {
        n = (int)(i / q4);
        q = (int)(i % q4) * 4;
        const float* src = p.ws + (size_t)n * p.ncols + q;
#pragma unroll 4
        for (int k = kl; k < p.splits; k += KL) s += *(const f32x4*)(src + k * slab);
    }
    sm[kl][el] = s;
    __syncthreads();
    for (int o = KL / 2; o > 0; o >>= 1) {
        if (kl < o) sm[kl][el] += sm[kl + o][el];
        __syncthreads();
    }
    if (kl == 0 && i < total) {
        s = sm[0][el];
        const int tapi = q / p.C;
        const int c = q - tapi * p.C;
        const int jy = tapi / p.txn, jx = tapi - jy * p.txn;
        const int wr = p.ty_w0 + jy * p.ty_wstep, ws_ = p.tx_w0 + jx * p.tx_wstep;
        float* o = p.dw + (size_t)n * p.wt_ld + (wr * p.wS + ws_) * p.wC + p.wc0 + c;
        if (p.accumulate) s += *(const f32x4*)o;
        *(f32x4*)o = s;
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgReduceJob p) {
    __shared__ f32x4 sm[256];
    if (p.kl == 16) wgrad_reduce_body<16>(p, blockIdx.x, sm);
    else wgrad_reduce_body<4>(p, blockIdx.x, sm);
}

void wg_reduce_job_fill(WgReduceJob& j, const zsg_conv_desc* d, const float* ws, float* dw, int accumulate, int splits) {
    memset(&j, 0, sizeof(j));
    j.ws = ws; j.dw = dw; j.accumulate = accumulate ? 1 : 0; j.splits = splits;
    j.N = d->N; j.C = d->C; j.wS = d->wS; j.wC = d->wC; j.wc0 = d->wc0; j.wt_ld = d->wt_ld;
    j.txn = d->seg[0].tx.n;
    j.ncols = d->seg[0].ty.n * d->seg[0].tx.n * d->C;
    j.ty_w0 = d->seg[0].ty.w0; j.ty_wstep = d->seg[0].ty.wstep; j.tx_w0 = d->seg[0].tx.w0; j.tx_wstep = d->seg[0].tx.wstep;
    j.kl = wg_reduce_kl(j.N, j.ncols, splits);
}

int wg_reduce_launch(const WgReduceJob& j, hipStream_t st) {
    ZSG_PROF("wgrad_reduce_kernel", st, 0, (double)(j.splits + 1) * j.N * j.ncols * 4);
    ZSG_LAUNCH(wgrad_reduce_kernel, dim3(wg_reduce_blocks(j.N, j.ncols, j.kl)), dim3(256), 0, st, j);
    return 0;
}

// Block tile (32*TM*WM) x (32*TN*WN) computed by WM x WN waves, each TM x TN MFMA tiles of 32x32.
// AVEC: dY rows are 16-byte addressable (out_ld % 4 == 0; a ragged channel count just reads the row's own padding).
// WIDE: an image stride (src_bstride / out_bstride) does not fit mul24's 24 signed bits (an activation of >= 2^23 elements per image:
// the stem / layer1 maps of inputs beyond ~724x724): the batch-index x image-stride products use the 32-bit multiply.  Only the
// 64x64 tile is instantiated that way (the host maps every tile hint onto it): a correct fallback, not a tuned path.
// DENSE: a 1x1 / stride-1 / unpadded convolution over batch-dense tensors (every bottleneck conv1 / conv3, the FPN laterals): pixel
// row r of a level is row r of both operands, so a K tile is 16 (32) CONSECUTIVE rows.  The loader then has no per-lane address
// arithmetic at all: every lane's byte offset inside a tile is a kernel-lifetime constant, the tile's position travels in the
// buffer descriptor's base (scalar adds) and its num_records bound cuts the level's last, ragged tile (and a dead prefetch) to
// zeros — 0 VALU per K tile instead of ~25 per load next to fp32 MFMAs that nothing co-issues with (tools/ubench/mfma_coissue.hip).
template <int TM, int TN, int WG_BK, int WM, int WN, bool AVEC = true, bool WIDE = false, bool DENSE = false>
__global__ __launch_bounds__(64 * WM * WN, 2) void wgrad_kernel(const WgParams p) {
    constexpr int NT = 64 * WM * WN;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int LDA = BM + WG_PAD, LDB = BN + WG_PAD;
    constexpr int GA = BM / 4, PA = NT / GA, NA = WG_BK / PA;       // A staging: groups/row, rows/pass, passes
    constexpr int GB = BN / 4, PB = NT / GB, NB = WG_BK / PB;
    static_assert(NA >= 1 && NB >= 1, "K tile too short for this thread count");
    extern __shared__ __attribute__((aligned(16))) float wg_smem[];
    typedef float (*TileA)[WG_BK][LDA];
    typedef float (*TileB)[WG_BK][LDB];
    TileA As = (TileA)wg_smem;                                    // [2][WG_BK][LDA]
    TileB Bs = (TileB)(wg_smem + 2 * WG_BK * LDA);                // [2][WG_BK][LDB]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const int nmn = p.m_tiles * p.n_tiles;
    const int split = blockIdx.x / nmn;
    const int mn = xcd_remap(blockIdx.x % nmn, nmn);
    const int mt = mn / p.n_tiles, nt = mn % p.n_tiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int kt_begin = split * p.kt_chunk;
    const int kt_end = min(p.kt_total, kt_begin + p.kt_chunk);

    // ---- fixed per-thread column state -----------------------------------------------------------------------
    const int ga = tid % GA, ka = tid / GA;
    const int gb = tid % GB, kb = tid / GB;
    const int na = m0 + 4 * ga;                    // first dY channel of this thread's 16-byte group
    const bool a_colok = na < p.N;
    const int q = n0 + 4 * gb;                     // first logical weight column of this thread's group
    const bool b_colok = q < p.ncols;
    int b_dy, b_dx, b_c;
    {
        const int qq = b_colok ? q : 0;
        const int tapi = qq / p.C;
        b_c = qq - tapi * p.C;
        const int jy = tapi / p.txn, jx = tapi - jy * p.txn;
        b_dy = p.ty.d0 + jy * p.ty.dstep;
        b_dx = p.tx.d0 + jx * p.tx.dstep;
    }
    const rsrc_t rs_a = make_rsrc(p.dy);
    const rsrc_t rs_b = make_rsrc(p.src);

    int si = 0;
#pragma unroll
    for (int s = 1; s < ZSG_MAX_SEG; ++s)
        if (s < p.nseg && kt_begin >= p.seg[s].kt0) si = s;
    WgSegDev sg = p.seg[si];
    int kt_next = kt_begin;

    // register stages: the global loads run NS K tiles ahead
    constexpr int NS = 2;            // (4 stages measured together with igemm.hip's: no gain, see there)
    f32x4 ra[NS][NA], rb[NS][NB];
    unsigned a_vo[NA], b_vo[NB];     // DENSE: this lane's byte offsets inside a K tile
    if (DENSE) {
#pragma unroll
        for (int j = 0; j < NA; ++j) a_vo[j] = a_colok ? 4u * (unsigned)((ka + PA * j) * p.out_ld + na) : ZSG_OOB;
#pragma unroll
        for (int j = 0; j < NB; ++j) b_vo[j] = b_colok ? 4u * (unsigned)((kb + PB * j) * p.src_ld + b_c) : ZSG_OOB;
    }
    // live == false (past this block's last K tile): every lane gets an out-of-range offset — the loads still issue and
    // return zeros without touching memory, so the K loop has no branch around them and the compiler counts the
    // outstanding loads exactly (with a branch it waited for ALL of them, vmcnt(0), before parking the previous tile).
    auto load_tile = [&](f32x4 (&ra)[NA], f32x4 (&rb)[NB], bool live) {
        if (si + 1 < p.nseg && kt_next >= p.seg[si + 1].kt0) {     // wave-uniform segment switch
            ++si;
            sg = p.seg[si];
        }
        const int rbase = (kt_next - sg.kt0) * WG_BK;
        if (DENSE) {
            const unsigned left = live ? (unsigned)(sg.rows - rbase) : 0u;          // rows of the level from this tile on (wave-uniform)
            const rsrc_t ta = make_rsrc_n(p.dy + sg.out_off + rbase * p.out_ld, 4u * left * (unsigned)p.out_ld);
            const rsrc_t tb = make_rsrc_n(p.src + sg.src_off + rbase * p.src_ld, 4u * left * (unsigned)p.src_ld);
#pragma unroll
            for (int j = 0; j < NA; ++j) ra[j] = buf_load4(ta, a_vo[j]);
#pragma unroll
            for (int j = 0; j < NB; ++j) rb[j] = buf_load4(tb, b_vo[j]);
            ++kt_next;
            return;
        }
        const int per = sg.rows_y * sg.rows_x;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int r = rbase + ka + PA * j;
            const bool rok = live & (r < sg.rows);
            const int rr = rok ? r : 0;
            int b = fdiv(rr, per, sg.inv_per);
            int rem = rr - mul24(b, per);
            int y = fdiv(rem, sg.rows_x, sg.inv_rx);
            int x = rem - mul24(y, sg.rows_x);
            const unsigned off = 4u * (unsigned)(sg.out_off + (WIDE ? b * sg.out_bstride : mul24(b, sg.out_bstride)) +
                                                  mul24(mul24(mul24(y, sg.osy) + sg.opy, sg.out_W) + (mul24(x, sg.osx) + sg.opx), p.out_ld) + na);
            if (AVEC) {
                ra[j] = buf_load4(rs_a, (rok & a_colok) ? off : ZSG_OOB);
            } else {                                                  // rows not 16-byte addressable (out_ld % 4 != 0)
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = buf_load1(rs_a, (rok & (na + e < p.N)) ? off + 4u * e : ZSG_OOB);
                ra[j] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int r = rbase + kb + PB * j;
            const bool rok = live & (r < sg.rows);
            const int rr = rok ? r : 0;
            int b = fdiv(rr, per, sg.inv_per);
            int rem = rr - mul24(b, per);
            int y = fdiv(rem, sg.rows_x, sg.inv_rx);
            int x = rem - mul24(y, sg.rows_x);
            const int yy = mul24(y, sg.sy) + b_dy, xx = mul24(x, sg.sx) + b_dx;
            const bool ok = rok & b_colok & ((unsigned)yy < (unsigned)sg.src_H) & ((unsigned)xx < (unsigned)sg.src_W);
            const unsigned off = 4u * (unsigned)(sg.src_off + (WIDE ? b * sg.src_bstride : mul24(b, sg.src_bstride)) + mul24(mul24(yy, sg.src_W) + xx, p.src_ld) + b_c);
            rb[j] = buf_load4(rs_b, ok ? off : ZSG_OOB);
        }
        ++kt_next;
    };
    auto store_tile = [&](int buf, const f32x4 (&ra)[NA], const f32x4 (&rb)[NB]) {
#pragma unroll
        for (int j = 0; j < NA; ++j) *(f32x4*)&As[buf][ka + PA * j][4 * ga] = ra[j];
#pragma unroll
        for (int j = 0; j < NB; ++j) *(f32x4*)&Bs[buf][kb + PB * j][4 * gb] = rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int n_kt = kt_end - kt_begin;
    if (n_kt > 0) {
        load_tile(ra[0], rb[0], true);
        store_tile(0, ra[0], rb[0]);
#pragma unroll
        for (int st = 0; st < NS - 1; ++st) load_tile(ra[st], rb[st], n_kt > st + 1);
    }
    __syncthreads();

    // lane (li, lh): MFMA sub-tile t covers rows m = wm*32*TM + 32*t + li (and columns likewise): contiguous runs of 32,
    // so the epilogue's stores are 128-byte coalesced.
    // MFMA sub-tile t of a wave covers rows m = TM*li + t (columns likewise): a lane's TM (TN) operands are CONSECUTIVE in the
    // [k][m] LDS tile, so one LDS instruction (ds_read2_b32 / b64) fetches both from adjacent banks.
    const int am = wm * (32 * TM) + TM * li;
    const int bn = wn * (32 * TN) + TN * li;
    auto k_step = [&](int it, f32x4 (&cur_a)[NA], f32x4 (&cur_b)[NB], f32x4 (&nxt_a)[NA], f32x4 (&nxt_b)[NB]) {
        const int buf = it & 1;
        load_tile(nxt_a, nxt_b, it + NS < n_kt);
        __builtin_amdgcn_sched_barrier(0);           // keep the global loads AHEAD of the MFMA phase (the scheduler sinks them otherwise)
#pragma unroll
        for (int kk = 0; kk < WG_BK / 2; ++kk) {
            const int k = 2 * kk + lh;
            float fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = As[buf][k][am + i];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = Bs[buf][k][bn + j];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        store_tile(buf ^ 1, cur_a, cur_b);           // (after the last tile: zeros into the idle buffer)
        __syncthreads();
    };
    for (int it = 0; it < n_kt; it += NS) {
#pragma unroll
        for (int st = 0; st < NS; ++st)
            if (it + st < n_kt) k_step(it + st, ra[st], rb[st], ra[(st + NS - 1) % NS], rb[(st + NS - 1) % NS]);
    }
    if (kt_begin >= kt_end) return;

    // ---- epilogue: D[i][j] -> (n = output channel, q = logical weight column) -------------------------------------
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int qc = n0 + wn * (32 * TN) + TN * li + j;
        const bool cok = qc < p.ncols;
        size_t coff;
        if (p.ws) {
            coff = (size_t)split * p.N * p.ncols + qc;
        } else {
            const int qq = cok ? qc : 0;
            const int tapi = qq / p.C;
            const int c = qq - tapi * p.C;
            const int jy = tapi / p.txn, jx = tapi - jy * p.txn;
            const int wr = p.ty.w0 + jy * p.ty.wstep, ws_ = p.tx.w0 + jx * p.tx.wstep;
            coff = (size_t)((wr * p.wS + ws_) * p.wC + p.wc0 + c);
        }
        const int ld = p.ws ? p.ncols : p.wt_ld;
        float* dst = p.ws ? p.ws : p.dw;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = m0 + wm * (32 * TM) + TM * ((e & 3) + 8 * (e >> 2) + 4 * lh) + i;
                if (cok && n < p.N) {
                    float* o = dst + (size_t)n * ld + coff;
                    *o = (!p.ws && p.accumulate) ? *o + acc[i][j][e] : acc[i][j][e];
                }
            }
        }
    }
}

extern "C" size_t zsg_conv_wgrad_workspace_bytes(const zsg_conv_desc* d) {
    if (!d) return 0;
    int splits = (d->tile_hint >> 16) & 0xff;
    if (splits <= 0) splits = 64;                       // upper bound of the heuristic
    const size_t ncols = (size_t)d->seg[0].ty.n * d->seg[0].tx.n * d->C;
    return (size_t)splits * d->N * ncols * sizeof(float);
}

// A/B switch for the DENSE loader (ZSG_WG_DENSE=0: every launch takes the decoding loader), read once
static const int g_zsg_wg_no_dense = [] { const char* e = getenv("ZSG_WG_DENSE"); return (e && e[0] == '0') ? 1 : 0; }();

static int conv_wgrad_impl(const zsg_conv_desc* d, const float* src, const float* dy, float* dw, int32_t accumulate, void* ws,
                           size_t ws_bytes, void* stream) {
    ZSG_REQUIRE(d && src && dy && dw, "conv_wgrad: null argument");
    ZSG_REQUIRE(d->nseg >= 1 && d->nseg <= ZSG_MAX_SEG, "conv_wgrad: nseg=%d", d->nseg);
    ZSG_REQUIRE(d->C > 0 && (d->C % 4) == 0 && (d->src_ld % 4) == 0 && (d->wC % 4) == 0 && (d->wc0 % 4) == 0,
                "conv_wgrad: C=%d src_ld=%d wC=%d wc0=%d must be multiples of 4", d->C, d->src_ld, d->wC, d->wc0);
    WgParams p;
    memset(&p, 0, sizeof(p));
    p.src = src; p.dy = dy; p.dw = dw; p.accumulate = accumulate ? 1 : 0;
    p.C = d->C; p.N = d->N; p.src_ld = d->src_ld; p.out_ld = d->out_ld; p.wS = d->wS; p.wC = d->wC; p.wc0 = d->wc0;
    p.wt_ld = d->wt_ld; p.nseg = d->nseg;
    p.ty = d->seg[0].ty; p.tx = d->seg[0].tx;
    p.txn = p.tx.n;
    p.ncols = p.ty.n * p.tx.n * d->C;
    int kt = 0;
    double rows_all = 0;
    bool wide = false;
    const bool avec = (d->out_ld & 3) == 0;                       // else: the single scalar-load variant (64x64 tile, BK 16)
    int BKsel = (avec && ((d->tile_hint >> 25) & 1) && ((d->tile_hint >> 8) & 0xff) != 255) ? 32 : 16;       // tile_hint bit 25: 32-pixel K tiles
    for (int s = 0; s < d->nseg; ++s)
        if (d->seg[s].src_bstride >= (1 << 23) || d->seg[s].out_bstride >= (1 << 23)) BKsel = 16;       // (the wide fallback: 64x64 tile, 16-pixel K tiles)
    p.bk = BKsel;
    for (int s = 0; s < d->nseg; ++s) {
        const zsg_seg& a = d->seg[s];
        ZSG_REQUIRE(memcmp(&a.ty, &p.ty, sizeof(zsg_taps)) == 0 && memcmp(&a.tx, &p.tx, sizeof(zsg_taps)) == 0,
                    "conv_wgrad: segments must share one tap structure (pass the forward descriptor)");
        const int64_t rows = (int64_t)d->B * a.rows_y * a.rows_x;
        ZSG_REQUIRE(rows > 0 && rows < (1ll << 24), "conv_wgrad: seg %d rows=%lld (must be < 2^24)", s, (long long)rows);
        ZSG_REQUIRE(a.src_off + (int64_t)d->B * a.src_bstride < (1ll << 29) && a.out_off + (int64_t)d->B * a.out_bstride < (1ll << 29),
                    "conv_wgrad: tensor exceeds 2^29 elements (2 GB window)");
        ZSG_REQUIRE((a.src_off % 4) == 0 && (a.src_bstride % 4) == 0, "conv_wgrad: seg %d source not 16-byte aligned", s);
        // the loader multiplies with 24-bit operands (mul24, wgrad_common.h): row pitches and per-image pixel counts must fit; an image
        // STRIDE beyond 2^23 elements (inputs larger than ~724x724 at the stem / layer1) selects the 32-bit-multiply variant below
        ZSG_REQUIRE(d->src_ld < (1 << 23) && d->out_ld < (1 << 23) && (int64_t)a.src_H * a.src_W < (1 << 23) &&
                        (int64_t)(a.rows_y * a.osy + a.opy + 1) * a.out_W < (1 << 23),
                    "conv_wgrad: seg %d: a row pitch / per-image pixel count exceeds 2^23", s);
        if (a.src_bstride >= (1 << 23) || a.out_bstride >= (1 << 23)) wide = true;
        WgSegDev& o = p.seg[s];
        o.rows_y = a.rows_y; o.rows_x = a.rows_x; o.rows = (int)rows; o.kt0 = kt;
        o.src_H = a.src_H; o.src_W = a.src_W; o.sy = a.sy; o.sx = a.sx;
        o.out_W = a.out_W; o.osy = a.osy; o.osx = a.osx; o.opy = a.opy; o.opx = a.opx;
        o.src_off = (int)a.src_off; o.src_bstride = (int)a.src_bstride;
        o.out_off = (int)a.out_off; o.out_bstride = (int)a.out_bstride;
        o.inv_per = 1.0f / (float)(a.rows_y * a.rows_x);
        o.inv_rx = 1.0f / (float)a.rows_x;
        kt += cdiv(rows, BKsel);
        rows_all += (double)rows;
    }
    p.kt_total = kt;
    // 1x1 / stride 1 / no padding over batch-dense tensors: pixel row r of a level is row r of both operands (the DENSE loader)
    bool dense = avec && !wide && p.ty.n == 1 && p.tx.n == 1 && p.ty.d0 == 0 && p.tx.d0 == 0 && !g_zsg_wg_no_dense;
    for (int s = 0; s < d->nseg && dense; ++s) {
        const zsg_seg& a = d->seg[s];
        dense = a.sy == 1 && a.sx == 1 && a.osy == 1 && a.osx == 1 && a.opy == 0 && a.opx == 0 && a.rows_y == a.src_H && a.rows_x == a.src_W &&
                a.out_W == a.rows_x && a.src_bstride == (int64_t)a.src_H * a.src_W * d->src_ld &&
                a.out_bstride == (int64_t)a.rows_y * a.rows_x * d->out_ld;
    }
    // tile_hint = BM | (BN << 8) | (splits << 16) (BM over output channels, BN over weight columns); 0 = heuristic
    int TM = (d->N > 64) ? 2 : 1;
    int TN = (p.ncols > 64) ? 2 : 1;
    int want_splits = 0, w8 = 0;
    if (d->tile_hint) {
        TM = ((d->tile_hint & 0xff) >= 128) ? 2 : 1;
        TN = (((d->tile_hint >> 8) & 0xff) >= 128) ? 2 : 1;
        // BN field 255 = a 256-column tile (TM = 1 only): the 64-output-channel layers whose weight rows are <= 256 columns wide (the
        // 7x7x4 stem: 196) get ALL columns from one block, so dY — the large operand there — is read once instead of once per column tile
        if (((d->tile_hint >> 8) & 0xff) == 255 && TM == 1) TN = 4;
        want_splits = (d->tile_hint >> 16) & 0xff;
        w8 = ((d->tile_hint >> 24) & 1) && TM == 2 && TN == 2;      // 8-wave workgroup: 128x128 tile only
    }
    if (!avec || wide) { TM = 1; TN = 1; w8 = 0; }
    p.m_tiles = cdiv(d->N, 64 * TM);
    p.n_tiles = cdiv(p.ncols, 64 * TN);
    const int nmn = p.m_tiles * p.n_tiles;
    int splits = want_splits > 0 ? want_splits : (2 * ZSG_NUM_CU + nmn - 1) / nmn;
    if (splits > 64 && want_splits <= 0) splits = 64;
    if (splits > kt / 2) splits = kt / 2;
    if (splits < 1) splits = 1;
    p.kt_chunk = cdiv(kt, splits);
    p.splits = cdiv(kt, p.kt_chunk);
    if (p.splits > 1) {
        const size_t need = (size_t)p.splits * d->N * p.ncols * sizeof(float);
        if (!ws || ws_bytes < need) ZSG_FAIL(-2, "conv_wgrad: workspace too small (%zu < %zu bytes)", ws_bytes, need);
        p.ws = (float*)ws;
    }
    hipStream_t st = (hipStream_t)stream;
    int dev = 0;
    (void)hipGetDevice(&dev);
    ZSG_REQUIRE(dev >= 0 && dev < ZSG_MAX_DEV, "conv_wgrad: device %d", dev);
    const double wg_flops = 2.0 * rows_all * d->N * p.ncols;
    const double wg_bytes = zsg_conv_alg_bytes(d, accumulate != 0);      // (dw takes the filter's place, dy the output's: same count)
    dim3 grid(nmn * p.splits);
#define WG_LAUNCH(TM_, TN_, BK_, WM_, WN_)                                                     \
    do {                                                                                       \
        if (dense) WG_LAUNCH_D(TM_, TN_, BK_, WM_, WN_, true, false, true, ", true, false, true"); \
        else WG_LAUNCH_D(TM_, TN_, BK_, WM_, WN_, true, false, false, "");                       \
    } while (0)
#define WG_LAUNCH_A(TM_, TN_, BK_, WM_, WN_, AV_) WG_LAUNCH_W(TM_, TN_, BK_, WM_, WN_, AV_, false)
#define WG_LAUNCH_W(TM_, TN_, BK_, WM_, WN_, AV_, WD_) WG_LAUNCH_D(TM_, TN_, BK_, WM_, WN_, AV_, WD_, false, "")
#define WG_LAUNCH_D(TM_, TN_, BK_, WM_, WN_, AV_, WD_, DN_, SFX_)                                                              \
    do {                                                                                                                   \
        const size_t lds = (size_t)2 * BK_ * ((32 * TM_ * WM_ + WG_PAD) + (32 * TN_ * WN_ + WG_PAD)) * sizeof(float);       \
        static bool attr_done[ZSG_MAX_DEV] = {};                                                                           \
        if (!attr_done[dev]) {                                                                                             \
            hipError_t e = hipFuncSetAttribute((const void*)wgrad_kernel<TM_, TN_, BK_, WM_, WN_, AV_, WD_, DN_>,              \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
            if (e != hipSuccess) ZSG_FAIL(-3, "wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));                      \
            attr_done[dev] = true;                                                                                         \
        }                                                                                                                  \
        ZSG_PROF("wgrad_kernel<" #TM_ ", " #TN_ ", " #BK_ ", " #WM_ ", " #WN_ SFX_ ">", st, wg_flops, wg_bytes);                \
        ZSG_LAUNCH((wgrad_kernel<TM_, TN_, BK_, WM_, WN_, AV_, WD_, DN_>), grid, dim3(64 * WM_ * WN_), lds, st, p);     \
    } while (0)
    if (wide) {                                       // an image stride >= 2^23 elements: 32-bit batch-offset multiplies
        if (avec) WG_LAUNCH_W(1, 1, 16, 2, 2, true, true);
        else WG_LAUNCH_W(1, 1, 16, 2, 2, false, true);
    } else if (!avec) {                               // dY rows not 16-byte addressable: the one scalar-load variant
        WG_LAUNCH_A(1, 1, 16, 2, 2, false);
    } else if (w8) {
        if (BKsel == 32) WG_LAUNCH(2, 1, 32, 2, 4);
        else WG_LAUNCH(2, 1, 16, 2, 4);
    } else if (BKsel == 32) {
        if (TM == 2 && TN == 2) WG_LAUNCH(2, 2, 32, 2, 2);
        else if (TM == 2 && TN == 1) WG_LAUNCH(2, 1, 32, 2, 2);
        else if (TM == 1 && TN == 2) WG_LAUNCH(1, 2, 32, 2, 2);
        else WG_LAUNCH(1, 1, 32, 2, 2);
    } else {
        if (TN == 4) WG_LAUNCH(1, 4, 16, 2, 2);
        else if (TM == 2 && TN == 2) WG_LAUNCH(2, 2, 16, 2, 2);
        else if (TM == 2 && TN == 1) WG_LAUNCH(2, 1, 16, 2, 2);
        else if (TM == 1 && TN == 2) WG_LAUNCH(1, 2, 16, 2, 2);
        else WG_LAUNCH(1, 1, 16, 2, 2);
    }
#undef WG_LAUNCH
#undef WG_LAUNCH_A
#undef WG_LAUNCH_W
#undef WG_LAUNCH_D
    if (p.splits > 1) {
        WgReduceJob j;
        wg_reduce_job_fill(j, d, p.ws, dw, p.accumulate, p.splits);
        wg_reduce_launch(j, st);
    }
    ZSG_CHECK_LAUNCH("conv_wgrad");
    return 0;
}

extern "C" int zsg_conv_wgrad(const zsg_conv_desc* d, const float* src, const float* dy, float* dw, int32_t accumulate, void* ws,
                              size_t ws_bytes, void* stream) {
    return conv_wgrad_impl(d, src, dy, dw, accumulate, ws, ws_bytes, stream);
}
