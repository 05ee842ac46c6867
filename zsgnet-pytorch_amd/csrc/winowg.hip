// winowg.hip — Winograd F(3x3, 2x2) weight gradient of the 3x3 / stride 1 / pad 1 convolutions on fp32 MFMA, gfx950.
//
// dW[n][c][u][v] = sum_{b,y,x} dY[b][y][x][n] * X[b][y+u-1][x+v-1][c]  is, per 2x2 tile of dY and the 4x4 input patch
// around it, a correlation of a 4x4 signal with a 2x2 "filter" producing 3x3 outputs — the transposed form of the forward
// F(2x2,3x3) algorithm (same 4x4 transformed domain, 16 multiplies per tile instead of 36):
//
//   dW = G^T [ sum_tiles (A dy A^T) .* (B^T d B) ] G        A = [[1,0],[1,1],[1,-1],[0,-1]],  G, B^T as in wino.hip
//
// GEMM view: 16 independent GEMMs (positions p = j*4 + i) with M = output channels n, N = input channels c and
// K = tiles, 8 tiles per LDS stage; both operands are transformed ON THE FLY while staging (lane (tile, 4-channel
// group, row q): the dY transform is in-lane, the input transform gets its column step from the quad neighbours with
// DPP, as in wino.hip; row 3 of both transforms is produced negated, which cancels in the product).  Operands are
// "MN-contiguous" ([k][m] LDS tiles, ds_read_b32 fragments, like wgrad.hip).  A 512-thread workgroup owns a 64 x 64
// (n, c) block of all 16 positions for its slice of the tiles (split-K over tiles, deterministic slab reduction with
// wgrad_reduce_kernel); wave (ph, wm, wn) holds the 8 positions with i in {2ph, 2ph+1} of one 32x32 sub-block; the
// output transform is linear, so each position half transforms its own share and the two are added through LDS.
#include "wgrad_common.h"

#define WW_KT 8                       // tiles per stage
#define WW_SP (WW_KT * 64 + 8)        // floats between positions (+8: conflict-free quad writes)
#define WW_BIAS (1 << 24)             // bytes: > 4 * (W + 1) * src_ld for every supported geometry (checked by the host)

struct WwSegDev {
    int tiles_y, tiles_x, tiles, st0;
    int H, W;
    int src_off, src_bstride, dy_off, dy_bstride;
    float inv_per, inv_tx;
};

struct WwParams {
    const float* src;
    const float* dy;
    float* dw;
    float* ws;
    int accumulate;
    int C, N, src_ld, dy_ld, wC, wc0, wt_ld, nseg;
    int m_tiles, n_tiles, splits, st_total, st_chunk, xmap;
    // Job batch (round 6, zsg_conv_wgrad_wino_batched): njobs convolutions of ONE geometry (a stage's identical bottlenecks) in one launch —
    // njobs x the (n, c) blocks at the same K range, i.e. a fraction of the split-K slabs and a longer stage loop per block.  Job j reads
    // srcj[j] / dyj[j], writes dwj[j] (or the j-th group of `splits` slabs in ws); njobs == 0: the plain launch (src / dy / dw above).
    int njobs;
    const float* srcj[ZSG_WG_MAX_JOBS];
    const float* dyj[ZSG_WG_MAX_JOBS];
    float* dwj[ZSG_WG_MAX_JOBS];
    WwSegDev seg[ZSG_MAX_SEG];
};

__device__ __forceinline__ float buf_load1_s(rsrc_t r, unsigned byte_off, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, soff, 0));
}
__device__ __forceinline__ float ww_quad_other(float v) {      // quad_perm [2,2,1,1]
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x5A, 0xf, 0xf, true));
}

__global__ __launch_bounds__(512) void wino_wgrad_kernel(const WwParams p) {
    constexpr int SP = WW_SP;
    extern __shared__ __attribute__((aligned(16))) float ww_smem[];
    float* Ps = ww_smem;                       // [2][16][SP]   transformed dY   [pos][tile][n]
    float* Vs = ww_smem + 2 * 16 * SP;         // [2][16][SP]   transformed input [pos][tile][c]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int ph = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    const int nmn = p.m_tiles * p.n_tiles;
    // xmap: an XCD owns WHOLE K slices (all (n, c) blocks of a slice on one L2: the slice's rows of dY / X are fetched into one L2
    // instead of eight); else the (n, c) blocks of every slice are spread over the XCDs
    int lb = p.xmap ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    // job batch: the blocks of a job are consecutive (job-major), so that a job's slices share their operands' cache lines in time
    const int per_job = nmn * p.splits;
    const int job = p.njobs ? lb / per_job : 0;
    lb -= job * per_job;
    // (a select chain on the wave-uniform job index: indexing the kernel-argument arrays dynamically makes hipcc copy them to scratch)
    const float* src_j = p.src;
    const float* dy_j = p.dy;
    float* dw_j = p.dw;
#pragma unroll
    for (int j = 0; j < ZSG_WG_MAX_JOBS; ++j)
        if (p.njobs && job == j) {
            src_j = p.srcj[j];
            dy_j = p.dyj[j];
            dw_j = p.dwj[j];
        }
    const int split = lb / nmn;
    const int mn = p.xmap ? lb % nmn : xcd_remap(lb % nmn, nmn);
    const int mt = mn / p.n_tiles, nt = mn % p.n_tiles;
    const int m0 = mt * 64, n0 = nt * 64;
    const int st_begin = split * p.st_chunk;
    const int st_end = min(p.st_total, st_begin + p.st_chunk);
    const int n_st = st_end - st_begin;

    // ---- loader: this wave stages tile `wave` of every stage; lane = (4-channel group g, patch row q) ----------------
    // Nothing co-issues with a SIMD's fp32 MFMA stream on gfx950 and every VALU instruction costs ~6 cycles on top of it
    // (tools/ubench/mfma_coissue.hip), so the loader keeps everything that is wave-uniform in SGPRs: the tile cursor (b, ty, tx)
    // advances by 8 tiles per stage with scalar adds / wraps (no per-stage divisions), the tile's base offsets travel in the buffer
    // instructions' soffset, and the per-lane parts of the addresses (pixel of the tile / patch, channel group; out of range where
    // this lane's row of the transform does not need the pixel) are loop constants that change only with the pyramid level.
    const int q = lane & 3, g = lane >> 2;
    const int a_col = m0 + 4 * g, b_col = n0 + 4 * g;
    const bool a_colok = a_col < p.N, b_colok = b_col < p.C;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const rsrc_t rs_a = make_rsrc(dy_j);
    // input patches reach one row and one column above / left of pixel (2ty, 2tx): the descriptor's base sits WW_BIAS bytes below the
    // tensor so that neither the scalar nor the per-lane part of an offset is ever negative (the range check looks at the per-lane part)
    const rsrc_t rs_b = make_rsrc((const char*)src_j - WW_BIAS);
    int si = 0;
#pragma unroll
    for (int s = 1; s < ZSG_MAX_SEG; ++s)
        if (s < p.nseg && st_begin >= p.seg[s].st0) si = s;
    WwSegDev sg = p.seg[si];
    int st_next = st_begin;
    int cur_t, cur_b, cur_ty, cur_tx;                 // this wave's tile of the next stage (SGPRs)
    unsigned dyv[4], xv[4];                           // per-lane byte offsets inside the tile / patch (level constants)
    auto seg_enter = [&](int t0) {                    // t0: wave-uniform tile index inside the level
        const int per = sg.tiles_y * sg.tiles_x;
        const int b = t0 / per, rem = t0 - b * per, ty = rem / sg.tiles_x;
        cur_t = t0;
        cur_b = __builtin_amdgcn_readfirstlane(b);
        cur_ty = __builtin_amdgcn_readfirstlane(ty);
        cur_tx = __builtin_amdgcn_readfirstlane(rem - ty * sg.tiles_x);
        // dY pixel (a, bb) of the 2x2 tile: row q of A dy needs dy row 0 for q < 3 and dy row 1 for q > 0
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const bool need = a_colok & (a == 0 ? q != 3 : q != 0);
                dyv[a * 2 + bb] = need ? 4u * (unsigned)((a * sg.W + bb) * p.dy_ld + a_col) : ZSG_OOB;
            }
        // input pixel (q, col) of the 4x4 patch, relative to pixel (2ty - 1, 2tx - 1) (the scalar part carries the - (W + 1) pixels)
#pragma unroll
        for (int col = 0; col < 4; ++col) xv[col] = b_colok ? 4u * (unsigned)((q * sg.W + col) * p.src_ld + b_col) : ZSG_OOB;
    };
    seg_enter((st_begin - sg.st0) * WW_KT + wave_u);

    f32x4 rd[4], rx[4];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto load_stage = [&](bool live) {
        if (si + 1 < p.nseg && st_next >= p.seg[si + 1].st0) {     // wave-uniform level switch
            ++si;
            sg = p.seg[si];
            seg_enter(wave_u);
        }
        const bool tok = live & (cur_t < sg.tiles);
        const int y0 = 2 * cur_ty, x0 = 2 * cur_tx;
        const int so_d = 4 * (sg.dy_off + cur_b * sg.dy_bstride + (y0 * sg.W + x0) * p.dy_ld);
        const int so_x = 4 * (sg.src_off + cur_b * sg.src_bstride + (y0 * sg.W + x0 - sg.W - 1) * p.src_ld) + WW_BIAS;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const bool ok = tok & (y0 + a < sg.H) & (x0 + bb < sg.W);          // wave-uniform
                rd[a * 2 + bb] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)(ok ? dyv[a * 2 + bb] : ZSG_OOB), so_d, 0));
            }
        const bool rok = tok & ((unsigned)(y0 - 1 + q) < (unsigned)sg.H);
#pragma unroll
        for (int col = 0; col < 4; ++col) {
            const bool ok = rok & ((unsigned)(x0 - 1 + col) < (unsigned)sg.W);
            rx[col] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_b, (int)(ok ? xv[col] : ZSG_OOB), so_x, 0));
        }
        ++st_next;
        cur_t += WW_KT;
        cur_tx += WW_KT;
        while (cur_tx >= sg.tiles_x) {
            cur_tx -= sg.tiles_x;
            ++cur_ty;
        }
        while (cur_ty >= sg.tiles_y) {
            cur_ty -= sg.tiles_y;
            ++cur_b;
        }
    };
    // dY transform A dy A^T, row q in-lane: rows (dy0, dy0 + dy1, dy0 - dy1, +dy1 [negated]) = dy0 + beta*dy1 with the dy row a
    // lane does not need loaded as zeros (seg_enter)
    const float beta = (q == 2) ? -1.f : 1.f;
    const float sgn = (q == 1) ? 1.f : -1.f;     // input transform: V[q] = own + sgn * other (row 3 negated), see wino.hip
    const int lds_w = q * SP + wave * 64 + 4 * g;
    auto store_stage = [&](int buf) {
        float* ps = Ps + buf * 16 * SP + lds_w;
        float* vs = Vs + buf * 16 * SP + lds_w;
        f32x4 r0, r1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            r0[e] = fmaf(beta, rd[2][e], rd[0][e]);
            r1[e] = fmaf(beta, rd[3][e], rd[1][e]);
        }
        *(f32x4*)(ps) = r0;                      // j = 0
        *(f32x4*)(ps + 4 * SP) = r0 + r1;        // j = 1
        *(f32x4*)(ps + 8 * SP) = r0 - r1;        // j = 2
        *(f32x4*)(ps + 12 * SP) = -r1;           // j = 3
        f32x4 t[4];
        t[0] = rx[0] - rx[2];
        t[1] = rx[1] + rx[2];
        t[2] = rx[2] - rx[1];
        t[3] = rx[1] - rx[3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(ww_quad_other(t[j][e]), sgn, t[j][e]);
            *(f32x4*)(vs + j * 4 * SP) = v;
        }
    };

    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    if (n_st > 0) {
        load_stage(true);
        store_stage(0);
    }
    __syncthreads();

    const int frag_a = 2 * ph * SP + lh * 64 + wm * 32 + li;
    const int frag_b = 2 * ph * SP + lh * 64 + wn * 32 + li;
    auto mfma_pos = [&](const float* a, const float* b, int pl) {       // position p = j*4 + 2*ph + il, pl = j*2 + il
        const int po = ((pl >> 1) * 4 + (pl & 1)) * SP;
#pragma unroll
        for (int ks = 0; ks < WW_KT / 2; ++ks)
            acc[pl] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[po + ks * 128], b[po + ks * 128], acc[pl], 0, 0, 0);
    };
    for (int it = 0; it < n_st; ++it) {
        const float* a = Ps + (it & 1) * 16 * SP + frag_a;
        const float* b = Vs + (it & 1) * 16 * SP + frag_b;
        load_stage(it + 1 < n_st);                 // (past the end: out-of-range offsets, zeros)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pl = 0; pl < 6; ++pl) mfma_pos(a, b, pl);
        __builtin_amdgcn_sched_barrier(0);
        mfma_pos(a, b, 6);
        mfma_pos(a, b, 7);
        store_stage((it + 1) & 1);
        __syncthreads();
    }
    if (n_st <= 0) return;

    // ---- output transform dW = G^T M G, this wave's rows i of M; G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]] ----------------
    // z[il][v] = sum_j M[i][j] G[j][v];  partial dW[u][v] = sum_il G[2ph+il][u] z[il][v]
    // Both position halves finish HALF of the elements each: half ph hands the other one the 8 accumulator rows it does not finish
    // (3 taps x 8 elements per lane and round) through LDS, all reads of a round are issued together, and the stores are branch-free
    // raw-buffer stores (out-of-range offset where the row / column is past N / C).  The per-element form this replaces (LDS read ->
    // wait -> conditional store, 144 times in half of the waves) cost 10 us per launch: 18 % of the 13-stage launches.
    float* xch = ww_smem;                           // [2 rounds in flight][2 directions][4 sub-blocks][24][64 lanes]
    const int sub = wave & 3;
    float* dst = p.ws ? p.ws + (size_t)(job * p.splits + split) * p.N * (9 * p.C) : dw_j;
    const int ld = p.ws ? 9 * p.C : p.wt_ld;
    const int tap_ld = p.ws ? p.C : p.wC;
    const int c = n0 + wn * 32 + li;
    const bool cok = c < p.C;
    const int col0 = p.ws ? c : (p.wc0 + c);
    const rsrc_t rs_o = make_rsrc(dst);
    const int nrow0 = m0 + wm * 32 + 4 * lh + 16 * ph;                  // this lane's first finished row: e = 8 ph + (e & 7)
    unsigned vo[8];                                                       // byte offsets of its 8 rows at tap 0 (level constants)
#pragma unroll
    for (int e8 = 0; e8 < 8; ++e8) {
        const int n = nrow0 + (e8 & 3) + 8 * (e8 >> 2);
        vo[e8] = (cok && n < p.N) ? 4u * (unsigned)(n * ld + col0) : ZSG_OOB;
    }
    const bool rmw = !p.ws && p.accumulate;                              // wave-uniform
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        float part[3][16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float z0[3], z1[3];
            {
                const float m0_ = acc[0][e], m1 = acc[2][e], m2 = acc[4][e], m3 = acc[6][e];
                const float h = 0.5f * (m1 + m2);
                z0[0] = m0_ + h; z0[1] = 0.5f * (m1 - m2); z0[2] = h + m3;
            }
            {
                const float m0_ = acc[1][e], m1 = acc[3][e], m2 = acc[5][e], m3 = acc[7][e];
                const float h = 0.5f * (m1 + m2);
                z1[0] = m0_ + h; z1[1] = 0.5f * (m1 - m2); z1[2] = h + m3;
            }
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                float y;
                if (ph == 0) y = (u == 0) ? z0[v] + 0.5f * z1[v] : 0.5f * z1[v];                       // rows i = 0, 1
                else y = (u == 0) ? 0.5f * z0[v] : ((u == 1) ? -0.5f * z0[v] : 0.5f * z0[v] + z1[v]);  // rows i = 2, 3
                part[v][e] = y;
            }
        }
        // hand over the half this wave does not finish: ph 0 gives e = 8..15, ph 1 gives e = 0..7
        float* give = xch + ((((u & 1) * 2 + ph) * 4 + sub) * 24) * 64 + lane;
        const float* take = xch + ((((u & 1) * 2 + (ph ^ 1)) * 4 + sub) * 24) * 64 + lane;
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) give[(v * 8 + e8) * 64] = ph ? part[v][e8] : part[v][8 + e8];
        __syncthreads();     // (one barrier per round: round u + 2 rewrites this region only after every wave has passed round u + 1's barrier, i.e. read it)
        float got[3][8];
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) got[v][e8] = take[(v * 8 + e8) * 64];
        float old[3][8];
        if (rmw) {
#pragma unroll
            for (int v = 0; v < 3; ++v)
#pragma unroll
                for (int e8 = 0; e8 < 8; ++e8) old[v][e8] = buf_load1_s(rs_o, vo[e8], 4 * (u * 3 + v) * tap_ld);
        }
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) {
                // the same sum as ever: (rows i = 0, 1 half) + (rows i = 2, 3 half)
                float val = ph ? got[v][e8] + part[v][8 + e8] : part[v][e8] + got[v][e8];
                if (rmw) val = old[v][e8] + val;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rs_o, (int)vo[e8], 4 * (u * 3 + v) * tap_ld, 0);
            }
    }
}

// split-K factor of a launch: tile_hint's, else enough K slices for one block per CU — ONE rule for the launch and for the
// workspace query (so a caller that sizes its workspace from the query never gets "workspace too small")
static int ww_pick_splits(const zsg_conv_desc* d, int stages, int njobs = 1) {
    const int nmn = cdiv(d->N, 64) * cdiv(d->C, 64) * njobs;
    int splits = (d->tile_hint >> 16) & 0xff;
    if (splits <= 0) splits = (ZSG_NUM_CU + nmn - 1) / nmn;
    if (splits > stages / 2) splits = stages / 2;
    if (splits < 1) splits = 1;
    return splits;
}

extern "C" size_t zsg_conv_wgrad_wino_workspace_bytes(const zsg_conv_desc* d) {
    if (!d || d->nseg < 1 || d->nseg > ZSG_MAX_SEG) return 0;
    int st = 0;
    for (int s = 0; s < d->nseg; ++s) st += cdiv((int64_t)d->B * ((d->seg[s].src_H + 1) / 2) * ((d->seg[s].src_W + 1) / 2), WW_KT);
    const int splits = ww_pick_splits(d, st);
    const int chunk = cdiv(st, splits);
    const int eff = cdiv(st, chunk);
    return eff > 1 ? (size_t)eff * d->N * 9 * d->C * sizeof(float) : 0;
}

// Same contract as zsg_conv_wgrad (forward descriptor, dy in the "out" geometry, accumulate flag, split-K workspace,
// deterministic slab reduction); 3x3 / stride 1 / pad 1 only.  tile_hint: split_k << 16 (0: heuristic), bit 24: block order "whole K slices per XCD".
static int conv_wgrad_wino_impl(const zsg_conv_desc* d, const float* src, const float* dy, float* dw, int32_t accumulate, void* ws,
                                size_t ws_bytes, void* stream, int njobs = 0, const float* const* srcj = nullptr, const float* const* dyj = nullptr,
                                float* const* dwj = nullptr) {
    ZSG_REQUIRE(d && src && dy && dw, "conv_wgrad_wino: null argument");
    ZSG_REQUIRE(njobs >= 0 && njobs <= ZSG_WG_MAX_JOBS, "conv_wgrad_wino_batched: %d jobs (at most %d)", njobs, ZSG_WG_MAX_JOBS);
    ZSG_REQUIRE(d->nseg >= 1 && d->nseg <= ZSG_MAX_SEG, "conv_wgrad_wino: nseg=%d", d->nseg);
    ZSG_REQUIRE(d->C > 0 && (d->C % 4) == 0 && (d->src_ld % 4) == 0 && (d->wC % 4) == 0 && (d->wc0 % 4) == 0 && (d->out_ld % 4) == 0,
                "conv_wgrad_wino: C=%d src_ld=%d wC=%d wc0=%d out_ld=%d must be multiples of 4", d->C, d->src_ld, d->wC, d->wc0, d->out_ld);
    ZSG_REQUIRE(d->wR == 3 && d->wS == 3 && !d->merge_x, "conv_wgrad_wino: 3x3 filters only");
    WwParams p;
    memset(&p, 0, sizeof(p));
    p.src = src; p.dy = dy; p.dw = dw; p.accumulate = accumulate ? 1 : 0;
    p.njobs = njobs;
    for (int j = 0; j < njobs; ++j) {
        ZSG_REQUIRE(srcj[j] && dyj[j] && dwj[j], "conv_wgrad_wino_batched: job %d has a null operand", j);
        p.srcj[j] = srcj[j]; p.dyj[j] = dyj[j]; p.dwj[j] = dwj[j];
    }
    const int jn = njobs ? njobs : 1;
    p.C = d->C; p.N = d->N; p.src_ld = d->src_ld; p.dy_ld = d->out_ld; p.wC = d->wC; p.wc0 = d->wc0; p.wt_ld = d->wt_ld; p.nseg = d->nseg;
    int st = 0;
    double rows_all = 0;
    for (int s = 0; s < d->nseg; ++s) {
        const zsg_seg& a = d->seg[s];
        ZSG_REQUIRE(a.ty.n == 3 && a.tx.n == 3 && a.sy == 1 && a.sx == 1 && a.osy == 1 && a.osx == 1 && a.opy == 0 && a.opx == 0 &&
                        a.ty.d0 == -1 && a.tx.d0 == -1 && a.ty.dstep == 1 && a.tx.dstep == 1 && a.ty.w0 == 0 && a.tx.w0 == 0 &&
                        a.ty.wstep == 1 && a.tx.wstep == 1,
                    "conv_wgrad_wino: seg %d is not a 3x3 / stride 1 / pad 1 forward descriptor", s);
        ZSG_REQUIRE(a.rows_y == a.src_H && a.rows_x == a.src_W && a.out_W == a.rows_x, "conv_wgrad_wino: seg %d: dy grid must equal the input grid", s);
        const int64_t tiles = (int64_t)d->B * ((a.src_H + 1) / 2) * ((a.src_W + 1) / 2);
        ZSG_REQUIRE(tiles > 0 && tiles < (1ll << 24), "conv_wgrad_wino: seg %d tiles=%lld (must be < 2^24)", s, (long long)tiles);
        ZSG_REQUIRE(a.src_off + (int64_t)d->B * a.src_bstride < (1ll << 29) && a.out_off + (int64_t)d->B * a.out_bstride < (1ll << 29),
                    "conv_wgrad_wino: tensor exceeds 2^29 elements (2 GB window)");
        ZSG_REQUIRE((a.src_off % 4) == 0 && (a.src_bstride % 4) == 0 && (a.out_off % 4) == 0 && (a.out_bstride % 4) == 0,
                    "conv_wgrad_wino: seg %d operands not 16-byte aligned", s);
        ZSG_REQUIRE(4ll * (a.src_W + 1) * d->src_ld < WW_BIAS, "conv_wgrad_wino: seg %d: row pitch too large", s);
        WwSegDev& o = p.seg[s];
        o.tiles_y = (a.src_H + 1) / 2; o.tiles_x = (a.src_W + 1) / 2; o.tiles = (int)tiles; o.st0 = st;
        o.H = a.src_H; o.W = a.src_W;
        o.src_off = (int)a.src_off; o.src_bstride = (int)a.src_bstride; o.dy_off = (int)a.out_off; o.dy_bstride = (int)a.out_bstride;
        o.inv_per = 1.0f / (float)(o.tiles_y * o.tiles_x);
        o.inv_tx = 1.0f / (float)o.tiles_x;
        st += cdiv(tiles, WW_KT);
        rows_all += (double)d->B * a.src_H * a.src_W;
    }
    p.st_total = st;
    p.m_tiles = cdiv(d->N, 64);
    p.n_tiles = cdiv(d->C, 64);
    const int nmn = p.m_tiles * p.n_tiles;
    const int splits = ww_pick_splits(d, st, jn);
    p.st_chunk = cdiv(st, splits);
    p.splits = cdiv(st, p.st_chunk);
    if (p.splits > 1) {
        const size_t need = (size_t)jn * p.splits * d->N * 9 * d->C * sizeof(float);
        if (!ws || ws_bytes < need) ZSG_FAIL(-2, "conv_wgrad_wino: workspace too small (%zu < %zu bytes)", ws_bytes, need);
        p.ws = (float*)ws;
    }
    p.xmap = (d->tile_hint >> 24) & 1;      // tile_hint bit 24: whole K slices per XCD (a tuner candidate: the large launches gain 1-2 %, the 13-stage ones lose 5 %)
    hipStream_t stq = (hipStream_t)stream;
    const size_t lds = (size_t)4 * 16 * WW_SP * sizeof(float);
    static bool attr_done[ZSG_MAX_DEV] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    ZSG_REQUIRE(dev >= 0 && dev < ZSG_MAX_DEV, "conv_wgrad_wino: device %d", dev);
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)wino_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) ZSG_FAIL(-3, "wgrad_wino: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done[dev] = true;
    }
    {
        ZSG_PROF("wino_wgrad_kernel", stq, jn * 2.0 * rows_all * d->N * 9.0 * d->C, jn * zsg_conv_alg_bytes(d, accumulate != 0));
        ZSG_LAUNCH(wino_wgrad_kernel, dim3(jn * nmn * p.splits), dim3(512), lds, stq, p);
    }
    if (p.splits > 1) {     // fixed-order slab sum -> dw (shared with the direct kernel), one per job
        for (int j = 0; j < jn; ++j) {
            WgReduceJob r;
            wg_reduce_job_fill(r, d, p.ws + (size_t)j * p.splits * d->N * 9 * d->C, njobs ? dwj[j] : dw, p.accumulate, p.splits);
            wg_reduce_launch(r, stq);
        }
    }
    ZSG_CHECK_LAUNCH("conv_wgrad_wino");
    return 0;
}

extern "C" int zsg_conv_wgrad_wino(const zsg_conv_desc* d, const float* src, const float* dy, float* dw, int32_t accumulate, void* ws,
                                   size_t ws_bytes, void* stream) {
    return conv_wgrad_wino_impl(d, src, dy, dw, accumulate, ws, ws_bytes, stream);
}

// njobs convolutions of ONE geometry (descriptor d) in one launch: job j = (src[j], dy[j]) -> dw[j]; accumulate and the workspace as above
// (the workspace holds njobs x split-K slabs: zsg_conv_wgrad_wino_workspace_bytes(d) x njobs is always enough).  The identical
// bottlenecks of a ResNet stage (fpn_resnet.py:86-100, layerN.1 .. layerN.k conv2): their weight gradients are leaves of the backward
// graph, so the lowering may hold them back until the last one's operands exist and release them together.
extern "C" int zsg_conv_wgrad_wino_batched(const zsg_conv_desc* d, int32_t njobs, const float* const* src, const float* const* dy, float* const* dw,
                                           int32_t accumulate, void* ws, size_t ws_bytes, void* stream) {
    ZSG_REQUIRE(njobs >= 1 && src && dy && dw, "conv_wgrad_wino_batched: bad argument");
    return conv_wgrad_wino_impl(d, src[0], dy[0], dw[0], accumulate, ws, ws_bytes, stream, njobs, src, dy, dw);
}
