// mx.hip — filter-resident streaming kernel for the network's FIRST convolution (the RGB input: C = 4 after padding, every tap
// row one contiguous run of <= 8 pixels x 4 channels — the descriptor's merge_x form): ResNet's 7x7 / stride-2 stem
// (fpn_resnet.py:80-83 `conv1`, torchvision resnet conv1 behind mdl.py:149-156) and SSD-VGG's 3x3 conv1_1 (ssd_vgg.py:54-66), 64
// output channels, fp32 MFMA, gfx950.  Reached through zsg_conv_igemm with tile_hint BM = 32 on a merge_x descriptor; the host
// autotuner times it next to the implicit-GEMM tiles.
//
// Why a kernel of its own.  As 64- / 128-row tiles of the implicit GEMM the stem (M = B x 150 x 150 = 360 000 pixels at the bench
// shape, N = 64, K = 7 x 28) takes 162 us for 57 us of MFMA work: a tile lives for 7 K steps, every one of them gathers its
// operand through LDS behind a barrier, 5 625 tiles each pay a prologue, an LDS-transposed epilogue and a BatchNorm partial row
// (the finalize launch behind it then reduces 5 625 rows: 20 us).  Here (the scheme of pw.hip, adapted to a strided window):
//   * a workgroup is PERSISTENT (one per CU); the whole filter [R][64][8 taps x 4] (64.5 KB for R = 7) is parked in LDS once;
//   * its eight waves are AUTONOMOUS: a wave walks "units" of 32 consecutive output pixels x all 64 output channels and no barrier
//     ties the waves together after the prologue;
//   * the PIXEL operand never touches LDS: lane (pixel i, half h) of a wave loads, per tap row, the four taps kx = 2 kq + h
//     (kq = 0..3) of ITS pixel's window as four 16-byte buffer loads (one pixel = 4 channels = 16 bytes) straight into the
//     registers the MFMA reads — v_mfma_f32_32x32x2_f32 takes k = 0 from lanes 0..31 and k = 1 from lanes 32..63, and the K order
//     of a dot product is free as long as both operands use the same one (here: k slot = 4 kx + channel; MFMA (kq, e) multiplies
//     slots 8 kq + e and 8 kq + 4 + e).  Zero padding, the tap row's 8th (absent) tap and pixels past the end are out-of-range
//     buffer offsets = hardware zeros, no branches.  Neighbouring windows overlap (stride 2 of 7 taps): every input byte is
//     requested ~12 times, from L1 / L2 (23 MB image batch), which the texture path has room for (282 MB per launch);
//   * all R tap rows of the NEXT unit are in flight while the current one is multiplied: row r's registers are refilled right
//     behind row r's MFMAs (a whole unit = ~6 us of prefetch distance, counted vmcnt waits);
//   * the MFMA operands are swapped as in pw.hip (filter rows first): lane (i, h)'s accumulator quad q holds, for PIXEL i, the
//     channels 8 q + 4 h .. + 3, so the transposition through the wave-private LDS buffer moves 16-byte units and every store
//     instruction writes 4 pixel rows x 256 contiguous bytes;
//   * ONE BatchNorm partial row per workgroup (a wave accumulates over all its units; lanes, then waves, in a fixed order:
//     deterministic) — 256 rows instead of 5 625 for the finalize launch.
// Units are dealt round-robin over (workgroup, wave slot): a SIMD's two waves (w, w + 4) get the same number of units +- 1.
// The K sum of an output runs over tap rows, then k slots, in the same order as the implicit GEMM's merge_x tile.
#include "common.h"

ZSG_DEFINE_PRIO_FLAG()

#define MX_TB 68          // floats per row of a wave's tile buffer: 64 + 4 = 17 x 16 B (odd: conflict-free b128 rows)
#define MX_LDW 36         // floats per filter row in LDS: 32 k slots + 4 (9 x 16 B, odd)
#define MX_WAVES 8
#define MX_N 64
#define MX_WINDOW 0x40000000u      // buffer window of the source: valid byte offsets < 2^30; the two out-of-range markers below add up
#define MX_OOB 0x40000000u         // to 2^31 at most, and either one alone lifts a valid offset out of the window

struct MxParams {
    const float* src;
    const float* wt;
    float* out;
    const float* bias;
    float* stats;             // [workgroups][2][64] partial rows (BatchNorm statistics) or nullptr
    int M;                    // output pixels (B x rows_y x rows_x)
    int per_img, rows_x;      // rows_y x rows_x, rows_x
    int src_H, src_W, sy, sx, d0y, d0x, dy;      // window geometry (dy: source rows per tap row = dilation)
    int TX;                   // taps per row (<= 8)
    int src_off, src_bstride; // elements
    int out_off, out_ld;      // elements
    int wt_ld;
    int relu;
    int units;                // ceil(M / 32)
};

__device__ __forceinline__ void mx_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void mx_store4(rsrc_t r, unsigned byte_off, f32x4 v) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)byte_off, 0, 0);
}

// the window addressing of one unit for this lane: byte offset of tap row 0's first pixel row (or out of range) and of the lane's
// four taps within a pixel row (or out of range)
struct MxAddr {
    unsigned xoff[4];         // (the first KQ are used)
    int ys0;                  // source row of tap row 0 (pixels past the end: far out of range)
    unsigned img;             // byte offset of the image
};

// R: tap rows; KQ: pairs of taps per row that exist (ceil(taps / 2): 4 for the 7-tap stem, 2 for a 3-tap row)
template <int R, int KQ>
__global__ __launch_bounds__(64 * MX_WAVES) void mx_kernel(const MxParams p) {
    ZSG_SET_MAIN_PRIO();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* Ws = smem;                                              // [R][64][MX_LDW]
    float* Tb = smem + R * MX_N * MX_LDW + wave * (32 * MX_TB);    // this wave's [32][MX_TB]
    const int li = lane & 31, lh = lane >> 5;                      // MFMA fragment coordinates: pixel / filter row, k half
    const int cg = lane & 15, rr = lane >> 4;                      // epilogue coordinates: 16-byte column group, row class
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, MX_WINDOW, 0x00020000);
    const int G = (int)gridDim.x, units = p.units, M = p.M;
    const int srcH = p.src_H, srcW = p.src_W, TX = p.TX;

    auto address = [&](int u) {
        MxAddr a;
        const int m = u * 32 + li;
        const bool ok = (u < units) & (m < M);
        const int mm = ok ? m : 0;
        const int b = mm / p.per_img;
        const int rem = mm - b * p.per_img;
        const int y = rem / p.rows_x;
        const int x = rem - y * p.rows_x;
        a.ys0 = ok ? y * p.sy + p.d0y : -(1 << 28);
        a.img = 4u * (unsigned)(p.src_off + b * p.src_bstride);
        const int xs = x * p.sx + p.d0x;
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
            const int kx = 2 * kq + lh, px = xs + kx;
            a.xoff[kq] = ((kx < TX) & ((unsigned)px < (unsigned)srcW)) ? 16u * (unsigned)px : MX_OOB;
        }
        return a;
    };
    f32x4 A[R][KQ];
#define MX_REQUEST(a, r)                                                                                         \
    {                                                                                                            \
        const int py__ = (a).ys0 + (r) * p.dy;                                                                   \
        const unsigned rb__ = ((unsigned)py__ < (unsigned)srcH) ? (a).img + 16u * (unsigned)(py__ * srcW) : MX_OOB; \
        _Pragma("unroll") for (int kq = 0; kq < KQ; ++kq) A[r][kq] = buf_load4(rs, rb__ + (a).xoff[kq]);          \
    }
    int u = (int)blockIdx.x + G * wave;
    {
        const MxAddr a0 = address(u);
#pragma unroll
        for (int r = 0; r < R; ++r) MX_REQUEST(a0, r);
    }
    // bias columns of the epilogue (requested before the filter: its waits cover them)
    const f32x4 cv0 = buf_load4(make_rsrc(p.bias ? p.bias : p.src), p.bias ? 16u * (unsigned)cg : ZSG_OOB);

    {   // the filter, once per workgroup: k slot 4 kx + c of tap row r <- wt[n][(r * TX + kx) * 4 + c]; absent taps are ZEROS (the
        // pixel operand's zeros must not meet uninitialised LDS)
        const rsrc_t rw = make_rsrc(p.wt);
        for (int idx = tid; idx < R * MX_N * 8; idx += 64 * MX_WAVES) {
            const int kx = idx & 7, n = (idx >> 3) & (MX_N - 1), r = idx >> 9;
            const f32x4 t = buf_load4(rw, kx < TX ? 4u * (unsigned)(n * p.wt_ld + (r * TX + kx) * 4) : ZSG_OOB);
            *(f32x4*)(Ws + (r * MX_N + n) * MX_LDW + 4 * kx) = t;
        }
    }
    __syncthreads();

    const rsrc_t rs_out = make_rsrc(p.out);
    const bool has_stats = p.stats != nullptr;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    for (; u < units; u += G * MX_WAVES) {
        const MxAddr an = address(u + G * MX_WAVES);           // (past the last unit: every offset out of range, no traffic)
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float* b = Ws + (r * MX_N + li) * MX_LDW + 4 * lh;
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq) {
                const f32x4 f0 = *(const f32x4*)(b + kq * 8), f1 = *(const f32x4*)(b + 32 * MX_LDW + kq * 8);
                const f32x4 fa = A[r][kq];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f0[e], fa[e], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f1[e], fa[e], acc[1], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            MX_REQUEST(an, r);                                 // row r of the next unit moves into the registers just consumed
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: transpose through the wave's buffer, 64 channels in one pass; branch-free raw buffer stores ----------------
        const int m0 = u * 32;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(f32x4*)(Tb + li * MX_TB + j * 32 + 8 * q + 4 * lh) = (f32x4){acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]};
        mx_wave_sync();
        f32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = *(const f32x4*)(Tb + (rr + 4 * i) * MX_TB + 4 * cg);
        mx_wave_sync();
        if (has_stats) {          // (dead rows hold exact zeros: no row test)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                s1 += v[i];
                s2 += v[i] * v[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[i] += cv0;
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[i][e] = fmaxf(v[i][e], 0.f);
            }
            const int m = m0 + rr + 4 * i;
            mx_store4(rs_out, m < M ? 4u * (unsigned)(p.out_off + m * p.out_ld + 4 * cg) : ZSG_OOB, v[i]);
        }
    }
#undef MX_REQUEST

    // ---- one partial row per workgroup: the four row classes of a column group, then the eight waves, in a fixed order -----------
    if (has_stats) {
        __syncthreads();                    // every wave has left the streaming loop: the filter is no longer needed
        float* red = smem;                  // [MX_WAVES][2][64]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s1[e] += __shfl_xor(s1[e], 16, 64);
            s1[e] += __shfl_xor(s1[e], 32, 64);
            s2[e] += __shfl_xor(s2[e], 16, 64);
            s2[e] += __shfl_xor(s2[e], 32, 64);
        }
        if (rr == 0) {
            *(f32x4*)(red + (wave * 2 + 0) * MX_N + 4 * cg) = s1;
            *(f32x4*)(red + (wave * 2 + 1) * MX_N + 4 * cg) = s2;
        }
        __syncthreads();
        if (tid < MX_N) {
            float a = 0.f, b = 0.f;
            for (int w = 0; w < MX_WAVES; ++w) {
                a += red[(w * 2 + 0) * MX_N + tid];
                b += red[(w * 2 + 1) * MX_N + tid];
            }
            float* o = p.stats + (size_t)blockIdx.x * 2 * MX_N;
            o[tid] = a;
            o[MX_N + tid] = b;
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------

// geometry the kernel covers: ONE merge_x segment (C = 4, <= 8 unit-step x taps), 3 or 7 tap rows, 64 output channels stored as
// dense pixel rows, a source of less than 2^30 bytes
bool zsg_conv_mx_ok(const zsg_conv_desc* d, const char** why) {
    static const char* msg;
    const char*& w = why ? *why : msg;
    if (!d->merge_x || d->nseg != 1) { w = "one merge_x segment"; return false; }
    const zsg_seg& s = d->seg[0];
    if (d->C != 4 || d->src_ld != 4 || d->wC != 4 || d->wc0 != 0 || s.tx.n < 1 || s.tx.n > 8 || s.tx.dstep != 1 || s.tx.wstep != 1 || s.tx.w0 != 0 ||
        d->wS != s.tx.n) { w = "C = 4 and 1..8 contiguous x taps"; return false; }
    if (!((s.ty.n == 7 && s.tx.n >= 7) || (s.ty.n == 3 && s.tx.n <= 4)) || s.ty.w0 != 0 || s.ty.wstep != 1 || s.ty.dstep < 1) { w = "a 7x7 / 7x8 or 3x(1..4) window"; return false; }
    if (d->N != MX_N) { w = "64 output channels"; return false; }
    if (s.osy != 1 || s.osx != 1 || s.opy != 0 || s.opx != 0 || s.out_W != s.rows_x || s.out_bstride != (int64_t)s.rows_y * s.rows_x * d->out_ld ||
        (d->out_ld % 4) || (s.out_off % 4) || (d->wt_ld % 4) || (s.src_off % 4) || (s.src_bstride % 4)) { w = "dense, 16-byte aligned output rows"; return false; }
    const int64_t rows = (int64_t)d->B * s.rows_y * s.rows_x;
    if (rows <= 0 || rows >= (1ll << 30) || s.out_off + rows * d->out_ld >= (1ll << 29)) { w = "output below 2^29 elements"; return false; }
    if (4 * (s.src_off + (int64_t)d->B * s.src_bstride) >= (int64_t)MX_WINDOW || (int64_t)s.src_H * s.src_W * 16 >= (int64_t)MX_WINDOW) { w = "a source below 2^30 bytes"; return false; }
    return true;
}
// workgroups of a launch = rows of BatchNorm partials it writes (at most one per CU)
int zsg_conv_mx_groups(const zsg_conv_desc* d) {
    const int64_t rows = (int64_t)d->B * d->seg[0].rows_y * d->seg[0].rows_x;
    const int g = cdiv(cdiv(rows, 32), MX_WAVES);
    return g < ZSG_NUM_CU ? g : ZSG_NUM_CU;
}

template <int R, int KQ>
static int mx_launch(const MxParams& p, int grid, hipStream_t st, double flops, double bytes, const char* kname) {
    const size_t lds = ((size_t)R * MX_N * MX_LDW + (size_t)MX_WAVES * 32 * MX_TB) * sizeof(float);
    static bool attr_done[ZSG_MAX_DEV] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    ZSG_REQUIRE(dev >= 0 && dev < ZSG_MAX_DEV, "conv_mx: device %d", dev);
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)mx_kernel<R, KQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) ZSG_FAIL(-3, "conv_mx: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done[dev] = true;
    }
    ZSG_PROF(kname, st, flops, bytes);
    ZSG_LAUNCH((mx_kernel<R, KQ>), dim3(grid), dim3(64 * MX_WAVES), lds, st, p);
    ZSG_CHECK_LAUNCH("conv_mx");
    return 0;
}

// called by conv_igemm_impl (igemm.hip) for tile_hint BM == 32 on a merge_x descriptor
int zsg_conv_mx_launch(const zsg_conv_desc* d, const float* src, const float* wt, float* out, const float* bias, const float* add_src,
                       const float* mask_src, float* bn_partials, hipStream_t st) {
    const char* why = "";
    ZSG_REQUIRE(zsg_conv_mx_ok(d, &why), "conv_igemm: the streaming first-layer kernel (tile_hint BM = 32, merge_x) needs %s", why);
    ZSG_REQUIRE(!add_src && !mask_src, "conv_igemm: the streaming first-layer kernel has no add / mask operands");
    const uintptr_t al = (uintptr_t)src | (uintptr_t)wt | (uintptr_t)out | (uintptr_t)bias | (uintptr_t)bn_partials;
    ZSG_REQUIRE((al & 15) == 0, "conv_igemm: the streaming first-layer kernel needs 16-byte aligned operands");
    if (bn_partials) ZSG_REQUIRE(!bias && !d->relu, "conv_igemm: BN-statistics fusion needs a plain (bias-free) convolution");
    const zsg_seg& s = d->seg[0];
    MxParams p;
    memset(&p, 0, sizeof(p));
    p.src = src; p.wt = wt; p.out = out; p.bias = bias; p.stats = bn_partials;
    p.M = (int)((int64_t)d->B * s.rows_y * s.rows_x);
    p.per_img = s.rows_y * s.rows_x; p.rows_x = s.rows_x;
    p.src_H = s.src_H; p.src_W = s.src_W; p.sy = s.sy; p.sx = s.sx; p.d0y = s.ty.d0; p.d0x = s.tx.d0; p.dy = s.ty.dstep;
    p.TX = s.tx.n;
    p.src_off = (int)s.src_off; p.src_bstride = (int)s.src_bstride;
    p.out_off = (int)s.out_off; p.out_ld = d->out_ld; p.wt_ld = d->wt_ld; p.relu = d->relu;
    p.units = cdiv(p.M, 32);
    const int grid = zsg_conv_mx_groups(d);
    const double flops = 2.0 * p.M * (double)d->N * s.ty.n * s.tx.n * d->C;
    const double bytes = zsg_conv_alg_bytes(d, false);
    if (s.ty.n == 7 && s.tx.n > 6) return mx_launch<7, 4>(p, grid, st, flops, bytes, "mx_kernel<7, 4>");
    if (s.ty.n == 3 && s.tx.n <= 4) return mx_launch<3, 2>(p, grid, st, flops, bytes, "mx_kernel<3, 2>");
    ZSG_FAIL(-1, "conv_igemm: the streaming first-layer kernel is built for 7x7 and 3x3 windows (%dx%d)", s.ty.n, s.tx.n);
}
