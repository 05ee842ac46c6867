/*
 * zsg.h — C ABI of libzsg.so: the MI355X (gfx950) HIP kernels behind the ZSGNet training step.
 *
 * Drop-in boundary.  The reference (TheShadow29/zsgnet-pytorch) is pure Python on PyTorch and has
 * NO native interface of its own: every device kernel it runs is an implicit nn / F / torch call.  Each entry
 * point below therefore cites the reference call site whose implicit PyTorch/cuDNN kernel(s) it replaces.  The host
 * side (the Python modules of zsgnet-pytorch_amd) mirrors the reference's nn.Module / loss / evaluator signatures and binds these
 * symbols with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.  All tensors are fp32 unless stated, device memory owned by the
 *     caller (PyTorch): the library never allocates, frees or retains device memory.
 *   - activations are NHWC ("pixel-major"): element (b,y,x,c) at  off + b*bstride + (y*W + x)*ld + c.
 *     weights are OHWI: element (co,r,s,ci) at ((co*R + r)*S + s)*C + ci   (== torch channels_last of an OIHW tensor).
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), re-entrant, and never synchronises.
 *   - return 0 on success, <0 on error: -1 bad argument/shape, -2 workspace too small, -3 HIP error, -4 RCCL error.
 *     zsg_last_error() returns a thread-local message.  Nothing throws across the ABI.
 */
#ifndef ZSG_H
#define ZSG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZSG_VERSION 100
#define ZSG_MAX_SEG 8

int zsg_version(void);
/* sha256 stamp (16 hex digits) of the kernel sources this library was BUILT from (csrc/stamp.py): the host ties the shipped tuning
 * table and the committed rocprof summaries to the library that is actually loaded (ZSG_LIB_PATH may point at another build), not to the
 * source files lying next to it.  No reference counterpart (the reference has no tuned native kernels). */
const char* zsg_source_stamp(void);
const char* zsg_last_error(void);
/* on != 0: the column-sum kernels (bias gradients, the head's border sums) use one block per output element group
 * instead of combining block partials with fp32 atomics, so every kernel of the library sums in a fixed order
 * (convolution split-K with atomics is only ever requested through tile_hint: the host tuner does not offer it in this
 * mode).  Process-global; the Python binding sets it from ZSG_DETERMINISTIC=1. */
int zsg_set_deterministic(int32_t on);

/* ---------------------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution family (fp32 MFMA v_mfma_f32_32x32x2_f32).
 * Replaces nn.Conv2d forward and its autograd backward at: encoder mdl.py:149-156 (-> fpn_resnet.py:80-100),
 * FPN fpn_resnet.py:157-172, head mdl.py:379-380, SSD ssd_vgg.py:72-95; LSTM input projection (mdl.py:227).
 *
 * One launch covers up to ZSG_MAX_SEG "segments" that share weights / channel counts but have their own geometry
 * (pyramid levels of the shared head; stride-parity classes of a strided dgrad).  A segment enumerates output rows
 * (b, y, x) over rows_y x rows_x per image; row (b,y,x) and tap (jy,jx) gather the source pixel
 *      (y*sy + ty.d0 + jy*ty.dstep,  x*sx + tx.d0 + jx*tx.dstep)      (zero outside [0,src_H) x [0,src_W))
 * and multiply it with weight tap (ty.w0 + jy*ty.wstep, tx.w0 + jx*tx.wstep).  The row is stored at output pixel
 *      (y*osy + opy, x*osx + opx) of an out_W-wide image.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n;      /* taps along this axis                      */
    int32_t w0;     /* first weight tap index                    */
    int32_t wstep;  /* weight tap index step                     */
    int32_t d0;     /* source offset of the first tap (pixels)   */
    int32_t dstep;  /* source offset step per tap                */
} zsg_taps;

typedef struct {
    int32_t rows_y, rows_x;          /* row grid per image                                   */
    int32_t src_H, src_W;            /* bounds of the gathered tensor                        */
    int32_t sy, sx;                  /* row -> source multiply                               */
    int32_t out_W;                   /* output image width (pixels)                          */
    int32_t osy, osx, opy, opx;      /* row -> output pixel                                  */
    int32_t reserved;
    int64_t src_off, src_bstride;    /* elements                                             */
    int64_t out_off, out_bstride;    /* elements                                             */
    zsg_taps ty, tx;
} zsg_seg;

typedef struct {
    int32_t B;            /* images                                                                         */
    int32_t C;            /* reduction channels per tap (multiple of 4)                                     */
    int32_t N;            /* output channels                                                                */
    int32_t src_ld;       /* source pixel stride (elements, multiple of 4)                                  */
    int32_t out_ld;       /* output pixel stride (elements)                                                 */
    int32_t wR, wS;       /* weight tap grid; weight row n starts at n*wt_ld, tap (r,s) at (r*wS+s)*wC       */
    int32_t wC;           /* channels per weight tap (>= C; C < wC selects a channel sub-range with wc0)     */
    int32_t wc0;          /* first weight channel used                                                      */
    int32_t wt_ld;        /* elements between consecutive weight rows                                       */
    int32_t relu;         /* epilogue: max(.,0)                                                             */
    int32_t merge_x;      /* 1: C==4 and all x-taps of a row are one contiguous run (stem / RGB input)       */
    int32_t nseg;
    int32_t tile_hint;    /* 0 heuristic, else BM | (BN << 8) | (split_k << 16) | variant bits; chosen by the host autotuner.
                           * bit 24: 8-wave workgroup (igemm, wgrad) / four position groups (wino); bit 25: 32-pixel K tiles
                           * (wgrad); bit 27 (igemm): 64-deep K tiles; bits 28-29 (igemm): stream-K with 1..3 workgroups per CU —
                           * 256 x that many workgroups share the (tile, K step) units evenly, cut tiles are completed in a fixed
                           * order (deterministic; needs zsg_set_stream_workspace, fewer tiles than workgroups, one segment,
                           * split_k <= 1)                                                                                    */
    int32_t epi_flags;    /* bit 0, BatchNorm-backward epilogues only (zsg_conv_*_bnb, *_bnb_tail): STORE the ReLU-masked gradient
                           * g = (acc [+ add_src]) * relu-bit instead of the unmasked sum — the stored dout is then at once the
                           * residual branch's gradient (autograd's ReLU backward of `out = relu(bn3(x) + residual)`,
                           * fpn_resnet.py:96-100), and the BatchNorm's apply pass needs neither the mask nor a second output     */
    zsg_seg seg[ZSG_MAX_SEG];
} zsg_conv_desc;

/* out = epilogue( sum_taps src * wt ) ;  epilogue: + bias[n] ; + add_src[same index as out] ; relu ;
 * * (mask_src[same index] > 0).   bias / add_src / mask_src may be NULL.  add_src may alias out (accumulate).
 * split_k > 1 (single dense segment, no ReLU): K slices are combined with fp32 atomics (output zeroed by the call). */
int zsg_conv_igemm(const zsg_conv_desc* d, const float* src, const float* wt, float* out, const float* bias,
                   const float* add_src, const float* mask_src, float* bn_partials, void* stream);
/* bn_partials (optional, may be NULL): [m_tiles][2][N] per-tile column (sum, sum of squares) of the output, written
 * by the epilogue so the following train-mode BatchNorm needs no extra pass over the activation
 * (m_tiles = sum over segments of ceil(rows / BM), BM = tile_hint & 0xff); finalize with zsg_bn_stats_from_partials. */

/* Weight gradient.  dw[n][(r*wS+s)*wC + wc0 + c] (+)= sum_rows dy[row][n] * src[gather(row, r, s)][c]
 * The descriptor is the FORWARD descriptor of the convolution (src = forward input, "out" geometry = dy); all
 * segments (pyramid levels of the shared head) are reduced in the one launch.  The pixel dimension is split over
 * tile_hint's split_k blocks whose partial tiles go to the workspace and are summed in a fixed order (deterministic).
 * Limits (error -1 otherwise): every tensor within a 2^29-element window, fewer than 2^24 pixel rows per segment, row pitches and
 * per-image pixel counts below 2^23; an IMAGE STRIDE of 2^23 elements or more (stem / layer1 maps of inputs beyond ~724x724) is
 * served by a 32-bit-multiply variant of the 64x64 tile (correct, not tuned).  1x1 / stride-1 / unpadded convolutions over
 * batch-dense tensors (src_bstride = H*W*src_ld, out_bstride = rows*out_ld) take the address-arithmetic-free DENSE loader. */
size_t zsg_conv_wgrad_workspace_bytes(const zsg_conv_desc* d);
int zsg_conv_wgrad(const zsg_conv_desc* d, const float* src, const float* dy, float* dw, int32_t accumulate, void* ws,
                   size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Winograd F(2x2,3x3) convolution on fp32 MFMA: the 3x3 / stride 1 / pad 1 convolutions (forward and data gradient)
 * of the same call sites as zsg_conv_igemm — fpn_resnet.py:73-74,92-94 (Bottleneck.conv2), :141-152 (P*_2),
 * mdl.py:211-219 (the shared head) — at 4 instead of 9 multiply-adds per (pixel, cin, cout), still fp32 throughout
 * (what cuDNN's fp32 path runs for the reference; tolerances in tests/test_gpu_ops.py).  Same descriptor, epilogue
 * terms and bn_partials contract as zsg_conv_igemm; `U` is the transformed filter image made by zsg_wino_weights.
 * tile_hint = tiles_per_block | (BN << 8) | (split_k << 16), both tile sizes in {32, 64}; bn_partials rows =
 * sum over segments of ceil(B * ceil(H/2) * ceil(W/2) / tiles_per_block).
 * ------------------------------------------------------------------------------------------------------------- */
int zsg_conv_wino(const zsg_conv_desc* d, const float* src, const float* U, float* out, const float* bias,
                  const float* add_src, const float* mask_src, float* bn_partials, void* stream);

/* Data gradient that COMPLETES dout of a train-mode BatchNorm (out = dgrad [+ add_src]; autograd's conv backward followed by
 * native_batch_norm_backward of fpn_resnet.py:86-97's conv-bn-relu chains): the epilogue also reduces that BatchNorm's
 * backward sums per output tile — partials[m_tile][0][n] = sum g, partials[m_tile][1][n] = sum g * (x - mean) * invstd with
 * g = out * relu-bit (bn_relu_mask as written by zsg_bn_apply, or NULL) — so that zsg_bn_backward_from_partials needs no
 * pass of its own over dout and x.  bn_x / bn_relu_mask are indexed with the convolution's OUTPUT element offsets (dense
 * [rows][N], the layout of dout).  Rows of partials: as bn_partials of zsg_conv_igemm / zsg_conv_wino for the same
 * tile_hint.  Requires split_k <= 1, every output element covered by the launch, N % 4 == 0. */
int zsg_conv_igemm_bnb(const zsg_conv_desc* d, const float* src, const float* wt, float* out, const float* add_src,
                       const float* bn_x, const float* bn_mean, const float* bn_invstd, const uint8_t* bn_relu_mask,
                       float* partials, void* stream);
/* tile_hint BM = 32 (BN field = unit width 32 / 64 / 128) selects the filter-resident streaming kernel for the 1x1 / stride-1
 * convolutions whose filter fits a CU's LDS (fpn_resnet.py:66-72: the bottleneck conv1 / conv3 / projection shortcut of the
 * first stage, forward and data gradient): one persistent workgroup per CU, every wave walks units of 32 pixels x BN channels
 * on its own.  It writes ONE partial row per workgroup; every other tile writes one per BM rows per segment.
 * zsg_conv_igemm_partial_rows: rows of bn_partials / partials a zsg_conv_igemm / zsg_conv_igemm_bnb launch with this
 * descriptor (and its tile_hint) writes; -1 when the hint is 0 (heuristic) or the streaming kernel does not cover the geometry. */
int32_t zsg_conv_igemm_partial_rows(const zsg_conv_desc* d);

/* ---- BatchNorm statistics finalised INSIDE the producing convolution (round 5; csrc/bn_tail.h) -----------------------------------
 * The tiles of one column block (BN output channels) take a ticket after writing their partial row; the tile that draws the last
 * ticket reduces the column block's rows in a fixed order in fp64 (deterministic) and publishes the result, so no finalize launch
 * (zsg_bn_stats_from_partials / the finalize half of zsg_bn_backward_from_partials) and no re-reduction in the apply pass remain on
 * the conv -> BatchNorm -> conv chain (fpn_resnet.py:80-100 forward; its autograd backward).
 * zsg_conv_bn_tail_tickets: the number of 32-bit ticket words the launch selected by d->tile_hint needs (its column blocks), or -1
 * when that launch cannot finalise in-kernel (no hint, the streaming 1x1 kernel, split_k > 1, merge_x, or more than 128 partial rows
 * per column block) — the caller then keeps the separate finalize.  is_wino != 0: the hint is zsg_conv_wino's.
 * tickets must be ZERO at entry and are zero again when the launch ends (the caller allocates them zeroed once; one set per call
 * site, never shared between launches that may be in flight together). */
int32_t zsg_conv_bn_tail_tickets(const zsg_conv_desc* d, int32_t is_wino);
/* The 1x1 / stride-1 convolution that consumes a train-mode BatchNorm + residual + ReLU — the next bottleneck's conv1 behind bn3
 * (fpn_resnet.py:86-100: `out = relu(bn3(conv3(..)) + residual)` ... `conv1(out)`) — applies that BatchNorm in its operand loader
 * (round 5): `x` is the RAW conv3 output [rows][C]; the loader stages relu((x - pre_mean) * pre_invstd * pre_gamma + pre_beta +
 * pre_residual), zsg_bn_apply's own arithmetic, and the first column tile's workgroups also write that activation to pre_y and its
 * packed ReLU bits to pre_relu_mask (NULL: no bits) — the separate zsg_bn_apply launch and the second read of the activation
 * disappear.  64x64 / 128x64 tiles with 32-deep K tiles only (tile_hint), C % 32 == 0, dense source.  bn_partials (optional): this
 * convolution's own fused statistics; tickets != NULL: finalised in-kernel exactly as zsg_conv_igemm_bnstat does. */
int zsg_conv_igemm_bnpre(const zsg_conv_desc* d, const float* x, const float* wt, float* out, float* bn_partials, uint32_t* tickets,
                         float* mean, float* invstd, float* running_mean, float* running_var, float momentum, float eps,
                         const float* pre_mean, const float* pre_invstd, const float* pre_gamma, const float* pre_beta,
                         const float* pre_residual, float* pre_y, uint8_t* pre_relu_mask, void* stream);
/* zsg_conv_igemm / zsg_conv_wino with bn_partials (plain, bias-free convolution feeding a train-mode BatchNorm, fpn_resnet.py:86-97)
 * + in-kernel finalize: mean / invstd (and running statistics, momentum as torch.nn.BatchNorm2d; NULL to skip) are valid when the
 * launch ends; zsg_bn_apply follows directly. */
int zsg_conv_igemm_bnstat(const zsg_conv_desc* d, const float* src, const float* wt, float* out, float* partials, uint32_t* tickets,
                          float* mean, float* invstd, float* running_mean, float* running_var, float momentum, float eps, void* stream);
int zsg_conv_wino_bnstat(const zsg_conv_desc* d, const float* src, const float* U, float* out, float* partials, uint32_t* tickets,
                         float* mean, float* invstd, float* running_mean, float* running_var, float momentum, float eps, void* stream);
/* zsg_conv_igemm_bnb / zsg_conv_wino_bnb + in-kernel finalize of the BatchNorm BACKWARD sums: coef[0][c] = sum g / n,
 * coef[1][c] = sum g * xhat / n (coef: 2*N floats), d(gamma) = sum g * xhat and d(beta) = sum g written (accumulate == 0) or added;
 * zsg_bn_bwd_apply is all that remains of the BatchNorm's backward. */
int zsg_conv_igemm_bnb_tail(const zsg_conv_desc* d, const float* src, const float* wt, float* out, const float* add_src,
                            const float* bn_x, const float* bn_mean, const float* bn_invstd, const uint8_t* bn_relu_mask,
                            float* partials, uint32_t* tickets, float* coef, float* dgamma, float* dbeta, int32_t accumulate, void* stream);
int zsg_conv_wino_bnb_tail(const zsg_conv_desc* d, const float* src, const float* U, float* out, const float* add_src,
                           const float* bn_x, const float* bn_mean, const float* bn_invstd, const uint8_t* bn_relu_mask,
                           float* partials, uint32_t* tickets, float* coef, float* dgamma, float* dbeta, int32_t accumulate, void* stream);
int zsg_conv_wino_bnb(const zsg_conv_desc* d, const float* src, const float* U, float* out, const float* add_src,
                      const float* bn_x, const float* bn_mean, const float* bn_invstd, const uint8_t* bn_relu_mask,
                      float* partials, void* stream);
/* elements of the transformed image of a C -> N 3x3 filter: [ceil(C/8)][16][roundup(N,64)][8] */
int64_t zsg_wino_u_elems(int32_t C, int32_t N);
/* U = G g G^T for every job in ONE launch.  jobs: device array of { int64 src, dst (absolute device addresses);
 * int32 N, C, src_row_ld, src_tap_ld, flip, Npad, chunks, blk0 }: source element (n, tap, c) at n*src_row_ld +
 * tap*src_tap_ld + c (an OHWI weight or a channel window of it; or a zsg_transpose_w image with flip = 1: the filter
 * rotated by 180 degrees, for the data gradient); blk0 = running sum of ceil(chunks*Npad*8 / 256). */
int zsg_wino_weights(const void* jobs, int32_t njobs, int32_t total_blocks, void* stream);

/* Winograd F(3x3,2x2) weight gradient of a 3x3 / stride 1 / pad 1 convolution: same contract as zsg_conv_wgrad (forward
 * descriptor, accumulate flag, split-K workspace [splits][N][9*C] with the deterministic slab reduction), 16 instead of
 * 36 multiply-adds per 2x2 output tile.  tile_hint: split_k << 16 (0: heuristic), bit 24: block order "whole K slices per XCD". */
size_t zsg_conv_wgrad_wino_workspace_bytes(const zsg_conv_desc* d);
int zsg_conv_wgrad_wino(const zsg_conv_desc* d, const float* src, const float* dy, float* dw, int32_t accumulate, void* ws,
                        size_t ws_bytes, void* stream);
/* Round 6: njobs (<= 8) convolutions of ONE geometry in one launch — job j: (src[j], dy[j]) -> dw[j]; the pointer arrays are HOST arrays
 * (read at the call).  The identical bottlenecks of a ResNet stage (/root/reference/code/fpn_resnet.py:86-100: layerN.1 .. layerN.k conv2;
 * autograd's weight gradient of each) are leaves of the backward graph: released together they are njobs x the (n, c) blocks at a
 * fraction of the split-K slabs.  Workspace: njobs x zsg_conv_wgrad_wino_workspace_bytes(d) is always enough; tile_hint as above
 * (the split-K factor is per job). */
int zsg_conv_wgrad_wino_batched(const zsg_conv_desc* d, int32_t njobs, const float* const* src, const float* const* dy, float* const* dw,
                                int32_t accumulate, void* ws, size_t ws_bytes, void* stream);

/* dst[c][t][n] = src[n][t][c]  (OHWI -> IHWO, the dgrad weight image); T = R*S; dst rows are dst_ld >= N wide
 * (columns N..dst_ld-1 are zeroed: the 45-channel head output is handled as a 48-channel GEMM operand). */
int zsg_transpose_w(const float* src, float* dst, int32_t N, int32_t T, int32_t C, int32_t dst_ld, void* stream);
/* All dgrad weight images of a step in ONE launch.  jobs: device array of
 * {int64 src_off, dst_off; int32 N, T, C, dst_ld, tile0, tiles_c, tiles_n, pad} (offsets in elements from the bases;
 * tiles_c = ceil(C/64), tiles_n = ceil(dst_ld/64), tile0 = running sum of T*tiles_c*tiles_n; round 5: 64 x 64 tiles moved with
 * 16-byte accesses — C, dst_ld and both offsets multiples of 4, bases 16-byte aligned). */
int zsg_transpose_w_batched(const float* src_base, float* dst_base, const void* jobs, int32_t njobs, int32_t total_tiles,
                            void* stream);
/* dst[r][0:dst_ld] = [ src[r*src_ld + 0:C] | 0 ... ] */
int zsg_pad_rows(const float* src, int64_t rows, int32_t C, int32_t src_ld, float* dst, int32_t dst_ld, void* stream);

/* out[g][c] (+)= sum_{r<rows} x[g*gstride + r*ld + c0 + c],  c < C   (bias gradients; per-image language-vector grads) */
int zsg_colsum(const float* x, int32_t groups, int64_t gstride, int32_t rows, int32_t ld, int32_t c0, int32_t C,
               float* out, int32_t accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * BatchNorm2d (train-mode batch statistics, momentum/eps as nn.BatchNorm2d) fused with ReLU / residual add.
 * Replaces nn.BatchNorm2d + ReLU(inplace) + `out += residual` of fpn_resnet.py:80-100 (53 layers, utils.py:395).
 * x: [rows][C] contiguous NHWC (ld == C), C % 4 == 0.
 * ------------------------------------------------------------------------------------------------------------- */
size_t zsg_bn_workspace_bytes(int64_t rows, int32_t C);
/* mean/invstd out; running_mean/var updated in place (unbiased var), NULL to skip. */
int zsg_bn_stats(const float* x, int64_t rows, int32_t C, float* mean, float* invstd, float* running_mean,
                 float* running_var, float momentum, float eps, void* ws, size_t ws_bytes, void* stream);
int zsg_bn_stats_from_partials(const float* partials, int32_t chunks, int64_t rows, int32_t C, float* mean, float* invstd,
                               float* running_mean, float* running_var, float momentum, float eps, void* stream);
/* Stem: nn.BatchNorm2d -> nn.ReLU -> nn.MaxPool2d(3, 2, 1) (mdl.py:149-152 on fpn_resnet.py's conv1 / bn1) in ONE pass over the
 * stem activation x [B][H][W][C] (the network's largest tensor): out [B][Ho][Wo][C] = maxpool(relu(bn(x))), idx = window position
 * of the first maximum (uint8, as zsg_maxpool_fwd).  The backward takes d(out): per-channel sums over the pooled gradient (the only
 * non-zero entries of the BatchNorm's grad_output), then dx per input pixel; dgamma / dbeta as zsg_bn_backward.
 * ws >= zsg_bn_workspace_bytes(B * Ho * Wo, C). */
int zsg_bn_relu_maxpool_fwd(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, const float* mean, const float* invstd,
                            const float* gamma, const float* beta, int32_t k, int32_t s, int32_t p, int32_t Ho, int32_t Wo, float* out,
                            uint8_t* idx, void* stream);
int zsg_bn_relu_maxpool_bwd(const float* dout, const uint8_t* idx, const float* x, int32_t B, int32_t H, int32_t W, int32_t C,
                            const float* mean, const float* invstd, const float* gamma, const float* beta, int32_t k, int32_t s, int32_t p,
                            int32_t Ho, int32_t Wo, float* dx, float* dgamma, float* dbeta, int32_t accumulate, void* ws, size_t ws_bytes,
                            void* stream);
/* BatchNorm apply straight from the convolution epilogue's partial rows when there are at most zsg_bn_inline_max_chunks()
 * of them (small maps: layer3 / layer4 / pyramid sizes): every block reduces the rows for its own channels (fp64, fixed
 * order), block 0 publishes mean / invstd / the running statistics — no separate finalize launch between the convolution
 * and the normalisation.  (The backward's partial rows are never that few at useful occupancy: a 64-chunk partial pass
 * measured 3x slower than the finalize launch it would save.) */
int32_t zsg_bn_inline_max_chunks(void);
int zsg_bn_apply_from_partials(const float* x, int64_t rows, int32_t C, const float* partials, int32_t chunks, const float* gamma,
                               const float* beta, const float* residual, int32_t relu, float* out, uint8_t* relu_mask,
                               float* mean, float* invstd, float* running_mean, float* running_var, float momentum, float eps,
                               void* stream);
/* eval mode, folded: every (conv, BatchNorm) pair of the job list gets W*s and (beta - mean*s), s = gamma/sqrt(var+eps)
 * per output channel, written to `arena` in ONE launch; the plan then runs conv(+bias, +residual, ReLU) without any
 * BatchNorm launch.  jobs: device array of { int64 w_off, dst_off, gamma_off, beta_off, bias_off; int32 row0, N, row_len,
 * bn_index } (element offsets into flat / arena; rows are OHWI output channels, row_len % 4 == 0). */
int zsg_bn_fold(const float* flat, const float* running_mean, const float* running_var, float eps, const void* jobs,
                int32_t njobs, int32_t total_rows, float* arena, void* stream);
/* eval mode: mean = running_mean, invstd = rsqrt(running_var + eps) */
int zsg_bn_eval_stats(const float* running_mean, const float* running_var, int32_t C, float eps, float* mean,
                      float* invstd, void* stream);
/* out = [relu]( (x-mean)*invstd*gamma + beta [+ residual] ).  relu_mask (optional, rows*C/4 bytes): bit e of byte i =
 * (element 4i+e > 0) — what zsg_bn_backward needs of the output, at 1/16 of its HBM traffic. */
int zsg_bn_apply(const float* x, int64_t rows, int32_t C, const float* mean, const float* invstd, const float* gamma,
                 const float* beta, const float* residual, int32_t relu, float* out, uint8_t* relu_mask, void* stream);
/* g = dout * (out > 0), the mask taken from relu_mask if given, else from relu_out if given, else g = dout;  dgamma = sum g*xhat ; dbeta = sum g ;
 * dx = gamma*invstd*(g - dbeta/n - xhat*dgamma/n) ; optional g_out = g (gradient of the residual branch).
 * dgamma/dbeta are ACCUMULATED (+=) when accumulate != 0, else overwritten. */
int zsg_bn_backward(const float* dout, const float* relu_out, const uint8_t* relu_mask, const float* x, int64_t rows, int32_t C,
                    const float* mean, const float* invstd, const float* gamma, float* dx, float* g_out,
                    float* dgamma, float* dbeta, int32_t accumulate, void* ws, size_t ws_bytes, void* stream);
/* The apply pass alone: dx = gamma*invstd*(g - coef[0] - xhat*coef[1]), g = dout * relu-bit, optional g_out = g; coef (2*C floats)
 * as finalised by zsg_conv_igemm_bnb_tail / zsg_conv_wino_bnb_tail (native_batch_norm_backward's input-gradient formula). */
int zsg_bn_bwd_apply(const float* dout, const uint8_t* relu_mask, const float* x, int64_t rows, int32_t C, const float* mean,
                     const float* invstd, const float* gamma, const float* coef, float* dx, float* g_out, void* stream);
/* The same from the partial rows [chunks][2][C] of zsg_conv_igemm_bnb / zsg_conv_wino_bnb (finalize + apply: one pass over
 * dout and x instead of two).  ws: >= 2*C floats. */
int zsg_bn_backward_from_partials(const float* dout, const uint8_t* relu_mask, const float* x, int64_t rows, int32_t C,
                                  const float* mean, const float* invstd, const float* gamma, float* dx, float* g_out,
                                  float* dgamma, float* dbeta, int32_t accumulate, const float* partials, int32_t chunks,
                                  void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Pooling / resampling / elementwise (NHWC, C % 4 == 0).
 * ------------------------------------------------------------------------------------------------------------- */
/* nn.MaxPool2d(k, s, p, ceil_mode) — mdl.py:152 (3,2,1), ssd_vgg.py:122,124,132.  idx: uint8 window position. */
int zsg_maxpool_fwd(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, int32_t s, int32_t p,
                    int32_t Ho, int32_t Wo, float* out, uint8_t* idx, void* stream);
int zsg_maxpool_bwd(const float* dout, const uint8_t* idx, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k,
                    int32_t s, int32_t p, int32_t Ho, int32_t Wo, float* dx, void* stream);
/* out = a + nearest_upsample(p -> Hd x Wd)   (F.interpolate(size=) + add, fpn_resnet.py:161-162,166-167) */
int zsg_upsample_add_fwd(const float* a, const float* p, int32_t B, int32_t Hs, int32_t Ws, int32_t Hd, int32_t Wd,
                         int32_t C, float* out, void* stream);
/* dp (+)= nearest-upsample adjoint of dout */
int zsg_upsample_add_bwd(const float* dout, int32_t B, int32_t Hs, int32_t Ws, int32_t Hd, int32_t Wd, int32_t C,
                         float* dp, int32_t accumulate, void* stream);
/* out = max(x,0) (fpn_resnet.py:172 F.relu(p6)) ; dx = dout * (x > 0) [+ dx] */
int zsg_relu_fwd(const float* x, int64_t n, float* out, void* stream);
int zsg_relu_bwd(const float* dout, const float* x, int64_t n, float* dx, int32_t accumulate, void* stream);
/* adaptive_avg_pool2d(.,1) — fpn_resnet.py:177 ; x [B][HW][C] -> out [B][C] ; and its adjoint */
int zsg_avgpool_fwd(const float* x, int32_t B, int32_t HW, int32_t C, float* out, void* stream);
int zsg_avgpool_bwd(const float* dout, int32_t B, int32_t HW, int32_t C, float* dx, int32_t accumulate, void* stream);
/* channel L2 normalisation x / ||x||_2 (no eps) — ssd_vgg.py:80, mdl.py:118-130 ; rows x C */
int zsg_l2norm_fwd(const float* x, int64_t rows, int32_t C, float* out, float* norm, void* stream);
int zsg_l2norm_bwd(const float* dout, const float* out, const float* norm, int64_t rows, int32_t C, float* dx,
                   void* stream);
/* image NCHW [B][3][H][W] -> NHWC4 [B][H][W][4] (4th channel zero) */
int zsg_nchw_to_nhwc4(const float* img, int32_t B, int32_t C, int32_t H, int32_t W, float* out, void* stream);
/* uint8 [pixels][3] (HWC, as PIL decodes) -> float [pixels][4] = (r,g,b)/255, 0 — `pil2tensor(img).float().div_(255)`
 * (dat_loader.py:26-33, 134) fused with the stem layout; IEEE division: equal to the host conversion bit for bit. */
int zsg_u8hwc_to_nhwc4(const uint8_t* img, int64_t pixels, float* out, void* stream);
/* `img.resize((Wo, Ho))` of the reference loader (dat_loader.py:121: PIL.Image.resize, default filter = bicubic) for one uint8
 * [H][W][C] image, byte-identical to Pillow: its two fixed-point passes (horizontal into tmp [H][Wo][C], then vertical) with the
 * per-axis tap tables the HOST computes from the two axis lengths (bounds[o] = {first tap, tap count}, coef[o][ksize] 22-bit
 * fixed-point weights; zsgnet_pytorch_amd.dat_loader.resize_tables).  A NULL table pair = that axis keeps its length. */
int zsg_resize_u8(const uint8_t* src, int32_t H, int32_t W, int32_t C, const int32_t* x_bounds, const int32_t* x_coef, int32_t x_ksize,
                  const int32_t* y_bounds, const int32_t* y_coef, int32_t y_ksize, int32_t Ho, int32_t Wo, uint8_t* tmp, uint8_t* out,
                  void* stream);
/* The same for a whole batch in TWO launches (round 5): jobs = device array of
 * {int64 src, tmp, out; int64 x_bounds, x_coef, y_bounds, y_coef; int32 h, w, x_ksize, y_ksize, blk0_x, blk0_y, pad, pad} — absolute
 * device addresses of the raw image [h][w][C], its scratch [h][Wo][C] and its result [Ho][Wo][C], the tap tables of its horizontal and
 * vertical pass (an axis that keeps its length carries the identity table: one tap, coefficient 2^22), the first 1024-output block of
 * the job in each launch (blocks_x / blocks_y = their totals).  Byte-identical to zsg_resize_u8 (dat_loader.py:98-146, :121). */
int zsg_resize_u8_batched(const void* jobs_dev, int32_t njobs, int32_t C, int32_t Ho, int32_t Wo, int32_t blocks_x, int32_t blocks_y,
                          void* stream);
/* Head input BackBone.concat_we (mdl.py:69-104) + head conv0 (mdl.py:216, 514 -> 256, 3x3 pad 1) without ever materialising
 * the concatenated tensor, and without its spatially-constant input channels: the language vector
 * is constant over the image and the grid channels do not depend on the batch index, so only the 256 feature channels
 * go through the implicit GEMM (half the MACs of the reference's dense conv); their contribution is an additive map
 *   out[b][p][n] = G[p][n] + sum_{tap valid at p} V[b][n*9 + tap],   V = W[:, :, lang] * we[b]  (tiny GEMM),
 * and the backward needs the validity-masked sums  S1[b][n*9+tap] += sum_{p: tap valid at p} dy[b][p][n]  (S2 = the
 * [n*9+tap][b] transpose) plus the batch sum of dy for the grid weights.  The masked sums come by inclusion-exclusion
 * from nine plain per-image sums Q (all pixels, first/last row, first/last column, four corners) that
 * zsg_head_border_sums accumulates level by level (caller zeroes Q) and zsg_head_border_finalize turns into S1, S2 and,
 * optionally, the bias gradient sum_b Q[0][b][:].  dy / out: one pyramid level [B][h*w][N]. */
int zsg_head_lang_map(const float* V, const float* G, int32_t B, int32_t h, int32_t w, int32_t N, float* out, void* stream);
/* Per-forward input staging in one launch (round 5): what `batch[k].to(device)` of the trainer (utils.py:403-405), the zero padding of
 * the query bucket, lstm_init_hidden's host draws (mdl.py:279-294: hc_src = h0 | c0, hc_n floats, may be PINNED HOST memory) and the
 * BatchNorm layers' num_batches_tracked += 1 (n_nbt int64 counters; 0 in eval mode) do at the head of ZSGNet.forward.
 * qvec [B][T][E] device fp32 -> qbuf [B][Tplan][E] (zeros behind T); qlens [B] (int64, or fp32 when qlens_f32) -> qlens_dst float [B]. */
int zsg_stage_inputs(const float* qvec, int32_t B, int32_t T, int32_t E, int32_t Tplan, float* qbuf, const void* qlens, int32_t qlens_f32, float* qlens_dst,
                     const float* hc_src, int32_t hc_n, float* hc_dst, int64_t* nbt, int32_t n_nbt, void* stream);
/* the same map for ALL pyramid levels in one launch (round 5): `out` [B][h_i*w_i][N] per level, levels packed level-major (level i
 * starts at B * N * sum_{j<i} h_j w_j); G (or NULL) packed the same way with B = 1; hw = {h_0, w_0, h_1, w_1, ...} (host memory,
 * nlev <= ZSG_MAX_SEG pairs).  A block sums V's taps once per border class instead of once per output element. */
int zsg_head_lang_map_packed(const float* V, const float* G, int32_t B, int32_t nlev, const int32_t* hw, int32_t N, float* out, void* stream);
int zsg_head_border_sums(const float* dy, int32_t B, int32_t h, int32_t w, int32_t N, float* Q /* [9][B][N], += */, void* stream);
int zsg_head_border_finalize(const float* Q, int32_t B, int32_t N, float* S1, float* S2, float* bias_grad /* [N], += ; or NULL */,
                             void* stream);
int zsg_batch_sum(const float* x, int32_t B, int64_t stride, float* out, void* stream);

/* Separate attention / box heads (use_same_atb = False, mdl.py:220-225, 383-389): their [rows][groups*k] outputs are
 * interleaved into the [B, A, 5] tensor the loss / evaluator read (dir 0), and the incoming gradient is split the same
 * way (dir 1):  strided[(r*groups + a)*group_stride + offset + e] <-> compact[(r*groups + a)*k + e]. */
int zsg_interleave(float* compact, int64_t rows, int32_t groups, int32_t k, float* strided, int32_t group_stride, int32_t offset,
                   int32_t dir, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * BiLSTM query encoder — nn.LSTM(300,128,bidirectional) on a PackedSequence + last-token gather, mdl.py:296-336.
 * gin: input projections x_t W_ih^T + b_ih  [B][T][4H] (made with zsg_conv_igemm);  one launch per direction.
 * The sorted-position rule (h0/c0 [B][H] of this direction are indexed by the rank of the sample in a stable
 * descending sort of the lengths, mdl.py:309) is evaluated on device.
 * forward direction: lens = qlens (float, as the collater makes them, dat_loader.py:193);
 * reverse direction: one cell step on x[len-1]: pass T=1, lens=NULL (all ones).
 * Saves for backward: gates [B][T][4H] (post-activation i,f,g,o), cst [B][T][H] (c_t), hprev [B][T][H] (h_{t-1}).
 * out: h at the last valid step, written to we[b*we_ld + we_off : +H].
 * ------------------------------------------------------------------------------------------------------------- */
int zsg_lstm_gather_last(const float* qvec, const float* qlens, int32_t B, int32_t T, int32_t E, float* out,
                         void* stream);
int zsg_lstm_fwd(const float* gin, const float* w_hh, const float* b_hh, const float* h0, const float* c0,
                 const float* qlens_rank, const float* lens, int32_t B, int32_t T, int32_t H, float* gates,
                 float* cst, float* hprev, float* we, int32_t we_ld, int32_t we_off, void* stream);
/* dgates [B][T][4H] (zero beyond the sample's length) from dwe; weight grads then come from zsg_conv_wgrad/colsum */
int zsg_lstm_bwd(const float* dwe, int32_t we_ld, int32_t we_off, const float* w_hh, const float* gates,
                 const float* cst, const float* c0, const float* qlens_rank, const float* lens, int32_t B, int32_t T,
                 int32_t H, float* dgates, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Anchor matching + focal / smooth-L1 loss, forward and backward in one call — ZSGLoss.forward, loss.py:43-143
 * (IoU_values anchors.py:90-116, simple_match_anchors :153-165, bbox_to_reg_params :168-179).
 * out5: network output [B][A][5] = (dy,dx,dh,dw,att);  annot [B][4] y1x1y2x2;  anchors [A][4] fp32 tlbr.
 * losses[3] = (loss, cls_ls, box_ls);  grad5 [B][A][5] = d loss / d out5 (already includes lamb_reg, 1/B, 1/#pos).
 * match_idx [B] int32 = arg-max-IoU anchor (lowest index wins; bit-exact IoU: no FMA contraction, IEEE divide).
 * flags: bit0 use_focal, bit1 use_multi, bit2 use_softmax.  NaN branch (loss.py:128-133) is taken on device.
 * Limit: B <= 512 samples per call (one LDS record per sample in the merge kernel); larger batches are rejected with -1
 * (the reference's per-GPU batches are 16-32, BASELINE configs; split a larger batch over calls and sum the losses).
 * ------------------------------------------------------------------------------------------------------------- */
size_t zsg_loss_workspace_bytes(int32_t B, int32_t A);
int zsg_loss_fwd_bwd(const float* out5, const float* annot, const float* anchors, int32_t B, int32_t A, float alpha,
                     float gamma, float lamb_reg, float match_thr, int32_t flags, float grad_scale, float* losses,
                     float* grad5, int32_t* match_idx, int32_t* npos, void* ws, size_t ws_bytes, void* stream);

/* Evaluator.forward, evaluator.py:48-117 (reg_params_to_bbox anchors.py:182-197): arg-max score anchor -> decode ->
 * IoU >= thr.  metrics[2] = (Acc, MaxPos); pred_boxes [B][4] pixels x1y1x2y2; pred_scores [B]; pred_idx [B] int32. */
size_t zsg_eval_workspace_bytes(int32_t B);
int zsg_eval(const float* out5, const float* annot, const float* anchors, const float* img_size, int32_t B, int32_t A,
             float acc_thr, float* metrics, float* pred_boxes, float* pred_scores, int32_t* pred_idx, int32_t* best_idx,
             float* ws /* zsg_eval_workspace_bytes(B): per-sample, per-anchor-range arg-max records */, void* stream);
/* IoU table [B][A] (tests / diagnostics) */
int zsg_iou(const float* boxes, const float* anchors, int32_t B, int32_t A, float* iou, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Fused Adam over a flat parameter buffer — torch.optim.Adam(betas=(0.9,0.99)) at main_dist.py:50 / utils.py:413.
 * step_count: device int32[2], zero-initialised by the caller: [0] = steps taken, incremented by the kernel (graph-capturable);
 * [1] = the kernel's completion ticket (0 between launches).  grad_scale folds 1/world_size.
 * ------------------------------------------------------------------------------------------------------------- */
int zsg_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, float grad_scale, int32_t* step_count, void* stream);
/* The same step as several launches over disjoint ranges (pointers offset by the caller, 16-byte aligned): every launch computes with
 * t = counter + 1; exactly the last one passes publish = 1 — and it must be ordered behind the others (same stream, or an event
 * edge): it advances the counter they read.  Lets the update of the parameters whose gradients are complete overlap the tail of
 * the backward (the stem's weight gradient). */
int zsg_adam_step_range(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, float grad_scale, int32_t* step_count, int32_t publish, void* stream);

int zsg_memset_f32(float* p, int64_t n, float value, void* stream);

/* Wave priority of the kernels of the step's dependent chain (convolutions forward / data gradient, BatchNorm passes, the small
 * main-stream kernels): 3 (default) = they issue ahead of the weight-gradient kernels wherever a CU holds waves of both streams
 * (s_setprio 3 as their first instruction), 0 = off.  A run-time switch since round 6 (a build-time constant before) so that a multi-GPU
 * run can compare both beside RCCL's priority-0 kernels.  Applies to the CURRENT device, synchronises it; call between steps.  No
 * reference counterpart (PyTorch / cuDNN kernels carry no priorities; the reference's overlap is NCCL's own, main_dist.py:36-40). */
int zsg_set_main_priority(int32_t prio);
int zsg_get_main_priority(void);

/* Scratch for the launches of one stream (round 6).  The reference has no counterpart: cuDNN / cuBLAS take their split-K workspace from
 * PyTorch's caching allocator behind nn.Conv2d (mdl.py:149-156, fpn_resnet.py:86-100).  Here the caller owns it: `ws` (256-byte aligned,
 * more than 16 KB; device memory of the current device) serves every later libzsg launch on `stream` that needs scratch — today the
 * stream-K implicit GEMM (tile_hint bits 28-29: one partial accumulator tile per workgroup + hand-off flags).  Launches of one stream are
 * ordered, so they share the buffer; two streams that run such launches concurrently register one buffer each.  The call clears the
 * first 16 KB (the flags) on `stream`; launches leave them zero.  ws = NULL unregisters.  A launch that needs scratch on a stream
 * without a registered buffer fails with -1, with a buffer that is too small with -2. */
int zsg_set_stream_workspace(void* stream, void* ws, size_t bytes);

/* ---------------------------------------------------------------------------------------------------------------
 * Cross-stream ordering without marker packets (SURVEY 8b "Threading / streams": asynchronous launches on the passed stream, a
 * dedicated side stream + hipEvents; the reference gets its concurrency from PyTorch's autograd / DDP reducer streams,
 * main_dist.py:37-40, utils.py:407-414).  The weight gradients of the backward (and some leaves of the forward) run on a side
 * stream; releasing them used to cost the main stream one hipEventRecord (a marker packet: ~4.3 us of main-stream time each, ~35
 * per step).  Instead the caller arms an event for its thread, makes ONE libzsg call — every kernel that call launches carries the
 * event as its dispatch packet's completion signal, the last launch wins — disarms it (NULL) and lets the other stream wait:
 *     zsg_set_completion_event(ev); zsg_conv_igemm(..., main); zsg_set_completion_event(NULL); zsg_stream_wait_event(side, ev);
 * Events are plain hipEvent_t (timing disabled) owned by the caller between create and destroy; zsg_event_record is the marker
 * fallback for a release point that no libzsg launch precedes, or when the armed call launched nothing (zsg_set_completion_event
 * returned 0 on disarming).  Thread-local arming: re-entrant across threads. */
void* zsg_event_create(void);
int zsg_event_destroy(void* ev);
int zsg_set_completion_event(void* ev);     /* returns how many launches carried the event armed before this call (0: none did) */
int zsg_event_record(void* ev, void* stream);
int zsg_stream_wait_event(void* stream, void* ev);

/* ---------------------------------------------------------------------------------------------------------------
 * Gradient exchange over RCCL (xGMI) — the NCCL collectives torch DistributedDataParallel issues for the reference
 * (main_dist.py:36-40, utils.py:395-414): C1 bucketed gradient all-reduce during backward, C2 BatchNorm-buffer
 * broadcast per training forward, C3 parameter broadcast at wrap time.  One process per GPU, one communicator per
 * process; the collectives run on the communicator's own non-blocking HIP stream, fenced against the caller's compute
 * stream with events (no host synchronisation).  librccl is bound with dlopen on first use (the copy PyTorch loaded).
 *   rank 0: zsg_comm_unique_id(id) -> share the 128 bytes with every rank (TCPStore / torch.distributed) ->
 *   every rank, on its own device: zsg_comm_init(&c, id, nranks, rank).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct zsg_comm zsg_comm;
int zsg_comm_unique_id(void* id128 /* out: 128 bytes */);
int zsg_comm_init(zsg_comm** out, const void* id128, int32_t nranks, int32_t rank);
/* in-place SUM all-reduce of buf[0:count], ordered after everything already enqueued on compute_stream; returns at once */
int zsg_comm_allreduce_bucket(zsg_comm* c, float* buf, int64_t count, void* compute_stream);
/* buf[0:count] of `root` to all ranks; compute_stream waits for it on the device */
int zsg_comm_broadcast(zsg_comm* c, float* buf, int64_t count, int32_t root, void* compute_stream);
/* compute_stream waits (device-side) for every bucket enqueued so far — call before the optimizer step */
int zsg_comm_wait(zsg_comm* c, void* compute_stream);
int zsg_comm_destroy(zsg_comm* c);

/* ---------------------------------------------------------------------------------------------------------------
 * Per-launch timing (HIP events on the launch stream) used by bench.py's roofline leg.  Not used in timed steps.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct {
    char name[48];
    int64_t launches;
    double ms;        /* summed kernel time between the bracketing events */
    double flops;     /* algorithmic 2*MAC for conv kernels, else 0       */
    double bytes;     /* algorithmic bytes moved (HBM-bound kernels)      */
} zsg_prof_entry;
int zsg_prof_enable(int32_t on);
int zsg_prof_collect(zsg_prof_entry* out, int32_t max_entries); /* syncs the recorded events; returns #entries */

#ifdef __cplusplus
}
#endif
#endif /* ZSG_H */
