/*
 * dvg_hip.h — C ABI of libdvg_hip.so, the MI355X (gfx950) kernel library behind
 * the DVG frame-prediction hot path.
 *
 * The reference (shgaurav1/DVG) has no FFI / plugin boundary of its own: every
 * device op is an implicit torch.nn / gpytorch call.  Each entry
 * point below therefore names the reference *call site* it replaces.  Citations
 * are file:line into the reference tree.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 data unless stated otherwise;
 *   - activations are NHWC ("channels last"): x[n][y][x][c]; the Python side
 *     exposes them as (N,C,H,W) torch tensors with channels_last strides;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - every function returns 0 on success, a DVG_ERR_* code otherwise and never
 *     throws; dvg_last_error() gives a thread-local message;
 *   - all shape checks happen on the host BEFORE a kernel is launched: a call
 *     that fails a check launches nothing.
 *   - no function allocates, frees or synchronises: graph-capture safe;
 *   - threading: ONE host thread per process drives the library (the reference's own
 *     contract, SURVEY.md 8(b) "Threading"; data parallelism is one process per GPU).  The
 *     first launch of each conv kernel instantiation raises its dynamic-LDS limit through an
 *     unsynchronised function-local flag (conv_igemm2.hip `attr_set`), and the debug hooks
 *     at the end of this file are plain globals.
 */
#ifndef DVG_HIP_H
#define DVG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVG_OK 0
#define DVG_ERR_SHAPE 1   /* unsupported / inconsistent shape            */
#define DVG_ERR_NULL 2    /* required pointer is NULL                    */
#define DVG_ERR_HIP 3     /* a HIP runtime call / kernel launch failed   */
#define DVG_ERR_ALIGN 4   /* pointer not 16-byte aligned                 */

/* activation codes for the fused epilogues */
#define DVG_ACT_NONE 0
#define DVG_ACT_LRELU 1   /* LeakyReLU(slope)   vgg_64.py:11, dcgan_64.py:10 */
#define DVG_ACT_TANH 2    /* vgg_64.py:47, dcgan_64.py:45,78, lstm.py:55     */
#define DVG_ACT_SIGMOID 3 /* vgg_64.py:91                                     */

int dvg_abi_version(void);
const char* dvg_last_error(void);
/* 0 when `stream` is not being captured into a hipGraph, else a nonzero id unique to that capture (r04, additive).  For
 * callers that cache buffers whose INITIAL CONTENTS matter (the split-K counter tail below): a buffer first touched during a
 * capture is only initialised when that graph replays, so such a cache must be keyed by the capture.                     */
long dvg_stream_capture_id(void* stream);

/* ------------------------------------------------------------------ *
 * Weight re-layout (one launch per parameter, cached by the caller).
 * ------------------------------------------------------------------ */

/* Conv2d weight (Cout,Cin,KH,KW) [vgg_64.py:8, dcgan_64.py:8] ->
 * packed [KH*KW][Cout][Cin] (Cin contiguous = implicit-GEMM K order).     */
int dvg_pack_conv_weight(const float* w_oihw, float* w_packed, int cout, int cin,
                         int kh, int kw, void* stream);

/* ConvTranspose2d weight (Cin,Cout,KH,KW) [vgg_64.py:88, dcgan_64.py:20,76]
 * -> packed [KH*KW][Cout][Cin] with the kernel spatially FLIPPED, i.e. the
 * weight of the equivalent direct correlation.                             */
int dvg_pack_convT_weight(const float* w_iohw, float* w_packed, int cin, int cout,
                          int kh, int kw, void* stream);

/* Inverse of the two packs (used by the backward pass to scatter dW back). */
int dvg_unpack_conv_weight(const float* w_packed, float* w_oihw, int cout, int cin,
                           int kh, int kw, void* stream);
int dvg_unpack_convT_weight(const float* w_packed, float* w_iohw, int cin, int cout,
                            int kh, int kw, void* stream);

/* ------------------------------------------------------------------ *
 * Encoder / decoder blocks
 * ------------------------------------------------------------------ */

/* vgg_layer = Conv2d(nin,nout,3,1,1)+BatchNorm2d+LeakyReLU(0.2)  (vgg_64.py:5-15)
 * as ONE fp32-MFMA implicit GEMM (M = N*H*W pixels, N = Cout, K = 9*Cin):
 *
 *   u        = conv3x3(in) + 0            (bias is folded into `shift`)
 *   y        = act(u * scale[c] + shift[c])
 *   y_pool   = maxpool2x2(y)              (optional; nn.MaxPool2d(2,2) vgg_64.py:49)
 *
 * `in` is either `x` (N,H,W,C1) or, when `upsample_x` != 0, the channel concat
 *   cat([nearest_up2(x), skip], C)   (vgg_64.py:93,98-105)
 * with x given at (N,H/2,W/2,C1) and skip at (N,H,W,C2); the concat and the
 * up-sampling are done by the tile loader and never materialised.
 * With skip == NULL, C2 must be 0.  When upsample_x == 0 and skip != NULL the
 * input is cat([x, skip]) at full resolution.
 *
 * scale/shift: [Cout] (eval-mode BN folded with the conv bias); either may be
 * NULL (=1 / =0).
 *
 * stats (optional, train-mode BN): float[rows][2][Cout] with
 * rows = dvg_conv_stats_rows_v2(...); every workgroup writes the
 * per-channel sum(u') and sum(u'^2) of its own pixel tile, where
 * u' = u*scale+shift BEFORE the activation (callers pass scale=NULL,
 * shift=bias, act=NONE to obtain the raw conv output and its statistics).
 * Deterministic (no atomics); dvg_bn_finalize reduces the rows.
 *
 * Requirements: C1 % 32 == 0, C2 % 32 == 0, Cout % 64 == 0, H % 8 == 0,
 * W % 8 == 0, all pointers 16-byte aligned.                                 */
#define DVG_MODE_CONV3 0
#define DVG_MODE_CONV4S2 1
#define DVG_MODE_CONVT4S2 2
/* rows of the `stats` partial buffer of the first-layer kernels (ks = 3 or 4); -1 if the
 * shape is unsupported.  (The implicit-GEMM convs: dvg_conv_stats_rows_v2 below.)          */
int dvg_conv_first_stats_rows(int ks, int N, int H, int W);

/* The three implicit-GEMM convs (conv_igemm2.hip; the "_v2" suffix is historical: the first
 * schedule and its un-suffixed entry points were retired in ABI 3).  Weights are packed by
 * dvg_pack_conv_weight_k16 (transposed != 0: ConvTranspose2d weight (Cin,Cout,KH,KW), flipped) into
 * Cin/16 * KH*KW * Cout rows of dvg_packed_row_floats() floats, [Cin/16][Cout/64][tap slot][64][row]
 * (a 4x4 transposed pack orders the slots by OUTPUT parity: it feeds dvg_convT4x4s2_bn_act_v2; a 4x4
 * plain pack orders them by INPUT parity (r05): it feeds dvg_conv4x4s2_bn_act_v2, which runs one stage per
 * parity and needs N * H * W * Cin < 2^31; a 3x3 pack feeds dvg_conv3x3_bn_act_v2), and `stats` has
 * dvg_conv_stats_rows_v2(...) rows.  C1, C2 multiples of 16; Cout multiple of 64.
 *
 * ABI 7 - arithmetic of the implicit-GEMM kernels.  dvg_mfma_mode() == 1 (the default build): every fp32
 * operand is split EXACTLY into three bf16 terms (8 + 8 + 8 significant bits; activations when a tile is
 * staged in LDS, weights when they are packed: a packed row is 3 x 16 bf16 = 24 floats) and a K = 16 slab
 * of the product is six v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the dropped cross terms are
 * below 2^-24 |a||b|, under the rounding of one fp32 product, and the measured error against fp64 is
 * that of the f32 MFMA or below.  dvg_mfma_mode() == 0 (library built with -DDVG_BF16X3=0): the native
 * v_mfma_f32_32x32x2_f32, packed rows of 16 floats.  Callers size packed buffers with
 * dvg_packed_row_floats() and are otherwise unaffected.                                          */
int dvg_mfma_mode(void);
int dvg_packed_row_floats(void);
/* ABI 9 - which BUILD of the library is loaded: a static string "abi=9 bf16x3=1 x3_terms=6 ablate=0 first_selects=0
 * timing_experiments=0 variant= src=<12 hex digits>".  `src` = sha256 over the library's sources at build time (`make -C
 * dvg_amd/csrc srcid` prints the tree's); x3_terms / ablate / first_selects / timing_experiments != 6 / 0 / 0 / 0 mark a
 * TIMING-ONLY build whose results are wrong by construction (conv_igemm2.hip), `variant` an A/B build (`make variant`).
 * The reference has no counterpart (one torch build per process); the host side (bench.py, dvg_amd/_lib.build_info) uses it
 * to refuse a headline measurement from anything but the product build.                                                 */
const char* dvg_build_info(void);
/* r06 - tile policy of the implicit-GEMM launches: 0 (default) = LATENCY, the tiles that make one launch alone on the chip fastest
 * (one chain of launches: a training iteration, GPtrigger_gen, a single sample); 1 = ENERGY, for callers that keep several
 * independent chains in flight (the samples of generate_frames.py:143-177's loop): the board then sits at its power cap and
 * larger register tiles - fewer LDS / L2 bytes per MFMA - give more frames per joule (vgg_64 rollouts in flight +4.6 %).  Within
 * one policy every form of a computation (eager, captured, in flight) is bit-identical; ACROSS policies results agree to fp32
 * rounding only (a rollout's frames to < 5e-6 max-norm, tests/test_gpu_headline.py): the tile shape fixes how a layer's K sum is
 * cut into partial sums.  Per-tile statistics rows differ in number - dvg_conv_splitk_v2 / dvg_conv_stats_rows_v2 answer for
 * the policy in force.  Process-global host state: set it before the
 * launches (or the hipGraph capture) it shall apply to.  The reference has no counterpart (cuDNN picks its own algorithms).   */
void dvg_set_tile_policy(int energy);
int dvg_tile_policy(void);
int dvg_pack_conv_weight_k16(const float* w, float* w_packed, int cout, int cin, int kh, int kw,
                             int transposed, void* stream);
/* Split-K: when a layer would launch < 384 workgroups (deep, narrow layers; small per-GPU batches) and the caller
 * supplies `workspace` (>= dvg_conv_splitk_v2(...) * N*Ho*Wo*Cout floats), K is split over workgroups, raw partial
 * tiles go to the workspace and a finish kernel applies scale/shift/act (+pool, +statistics).  workspace may be
 * NULL (no split).  `stats` then has dvg_conv_stats_rows_v2(..., pool, with_workspace) rows.
 * `addend` (may be NULL): raw pre-scale partial sums in y's NHWC shape, y = act((conv + addend) * scale + shift).
 * It carries the skip half of a decoder block's first conv, cat([up(d), skip]) (vgg_64.py:98-105, dcgan_64.py:84-86),
 * when the skip tensor is loop-invariant over the steps of a rollout (generate_frames.py:154-157): the caller
 * computes conv(skip, W[:, C1:]) once with scale = shift = NULL, act = NONE and then runs only the x half per step.
 * Statistics (train-mode BatchNorm) are taken over conv + addend; excludes the pooled output.                 */
int dvg_conv_splitk_v2(int mode, int N, int H, int W, int Cin, int Cout);
int dvg_conv_stats_rows_v2(int mode, int N, int H, int W, int Cin, int Cout, int pool, int with_workspace);
int dvg_conv3x3_bn_act_v2(const float* x, const float* skip, const float* w_k16,
                          const float* scale, const float* shift, float* y, float* y_pool,
                          float* stats, int N, int H, int W, int C1, int C2, int Cout,
                          int upsample_x, int act, float slope, float* workspace,
                          long workspace_floats, const float* addend,
                          /* ABI 6, time-batched decoder calls (train.py:227-231): the images form groups of addend_block;
                           * group g = n / addend_block adds block addend_map[g] (device ints) of an addend made of
                           * addend_block-image blocks - the skip half shared by the three decoder calls of a time step
                           * and, once the skip is frozen, by every later step.  NULL: image n adds addend image n.     */
                          const int* addend_map, int addend_block, void* stream);
/* dcgan_conv = Conv2d(nin,nout,4,2,1)+BN+LReLU (dcgan_64.py:4-14): K = 16*Cin,
 * x NHWC (N,H,W,Cin) -> y NHWC (N,H/2,W/2,Cout).                                          */
/* vgg_64's first stage in eval mode, c1 = vgg_layer(1, 64) -> vgg_layer(64, Cout) (vgg_64.py:23-26, 49), as ONE launch: the
 * second layer's implicit GEMM computes its input tile from the frame patch under it, so the 64-channel activation between
 * the two layers (67 MB at B = 64) is never written or read.  frame (N,1,H,W) NCHW; w0 = the first layer's (64,1,3,3) weight
 * transposed to [9 taps][64 channels]; scale0 / shift0 = the first
 * layer's folded BatchNorm (64 each; LeakyReLU `slope`); w1_k16 / scale1 / shift1 / y / y_pool / act as in
 * dvg_conv3x3_bn_act_v2 with C1 = 64.  H % 8 == 0, W % 16 == 0, N * H/8 * W/16 * Cout/64 >= 512.                          */
int dvg_conv3x3_first_pair(const float* frame, const float* w0, const float* scale0, const float* shift0,
                           const float* w1_k16, const float* scale1, const float* shift1, float* y, float* y_pool,
                           int N, int H, int W, int Cout, int act, float slope,
                           int y_from /* ABI 8: y holds the images [y_from, N) only - see dvg_winograd_output_pool_input */,
                           void* stream);
int dvg_conv4x4s2_bn_act_v2(const float* x, const float* w_k16, const float* scale,
                            const float* shift, float* y, float* stats, int N, int H, int W,
                            int Cin, int Cout, int act, float slope, float* workspace,
                            long workspace_floats, void* stream);
/* dcgan_upconv = ConvTranspose2d(nin,nout,4,2,1)+BN+LReLU on cat([x, skip])
 * (dcgan_64.py:16-26,84-87) as four parity-class implicit GEMMs (K = 4*Cin).
 * x NHWC (N,H,W,C1), skip NHWC (N,H,W,C2) or NULL; y NHWC (N,2H,2W,Cout).                 */
int dvg_convT4x4s2_bn_act_v2(const float* x, const float* skip, const float* w_k16,
                             const float* scale, const float* shift, float* y, float* stats,
                             int N, int H, int W, int C1, int C2, int Cout, int act, float slope,
                             float* workspace, long workspace_floats, const float* addend,
                             const int* addend_map /* as dvg_conv3x3_bn_act_v2 */, int addend_block, void* stream);

/* Winograd F(m x m, 3x3), m = 2 or 4, form of vgg_layer for the deep eval-mode layers (vgg_64.py:5-15 at 16x16 / 8x8 maps
 * with 256-512 channels): fp32 data and transforms, 2.25x (m = 2) / 4x (m = 4) fewer multiply-adds, P = (m+2)^2 transform
 * positions.  y = act(scale * A^T[(G g G^T) .* (B^T d B)]A + shift):
 *   dvg_winograd_weight   U: P * Cin/16 * Cout packed rows of dvg_packed_row_floats() floats, [P][Cout/64][Cin/16][64][row],
 *                         from the Conv2d weight (Cout,Cin,3,3)                                       once per weight version
 *   dvg_winograd_input    V (P, T, C)  from x NHWC (N,H,W,C), T = N*(H/m)*(W/m) tiles, zero padding 1
 *   dvg_gemm_batched_k16  M (P, T, Cout) = V x U: P GEMMs; the tensors are passed as P "images" of (T/16) x 16 pixels
 *   dvg_winograd_output   y NHWC (N,H,W,Cout) (+ y_pool, MaxPool2d(2,2) vgg_64.py:49: pool windows never straddle tiles)
 * H, W multiples of m; C, Cout % 64 == 0; T % 128 == 0.
 * r06: dvg_winograd_weight / dvg_pack_conv_weight_k16 were rewritten (same layout, same values) after the m = 4 transform wrote
 * zero rows into U whenever ANOTHER PROCESS was busy on the device ("Sharing a device").                */
int dvg_winograd_weight(const float* w_oihw, float* u_k16, int cout, int cin, int m, void* stream);
/* upsample = 1 (ABI 6, m = 4): x is stored at (N, H/2, W/2, C) and read through nn.UpsamplingNearest2d(2) (vgg_64.py:93) -
 * the x half of a decoder block's first conv in Winograd form; the upsampled tensor never exists.                       */
int dvg_winograd_input(const float* x, float* v, int N, int H, int W, int C, int m, int upsample, void* stream);
int dvg_gemm_batched_k16(const float* x, const float* w_k16, float* y, int NB, int H, int W, int Cin, int Cout,
                         void* stream);
/* addend (ABI 6; optional, m = 4, no y_pool): raw partial sums (N,H,W,Cout) NHWC added before scale / shift / activation -
 * the hoisted skip half of a decoder block's first conv, as in dvg_conv3x3_bn_act_v2.                                  */
int dvg_winograd_output(const float* mm, const float* scale, const float* shift, float* y, float* y_pool, int N, int H,
                        int W, int C, int act, float slope, int m, const float* addend,
                        int y_from /* ABI 8; 0 unless y_pool is given: as dvg_winograd_output_pool_input */, void* stream);
/* dvg_winograd_output of layer L and dvg_winograd_input of layer L+1 in one pass (m = 4, H == W in {8, 16, 32}, C % 64 == 0) for
 * two consecutive eval-mode vgg_layers at one resolution whose intermediate activation has no other consumer (the inner
 * layers of a vgg block, vgg_64.py:24-43,70-87): mm (36, T, C) -> v_next (36, T, C); the activation is not written.     */
int dvg_winograd_output_input(const float* mm, const float* scale, const float* shift, float* v_next, int N, int H, int W,
                              int C, int act, float slope, const float* addend /* as dvg_winograd_output; may be NULL */,
                              void* stream);
/* Decoder stem -> first conv of the first decoder block (vgg_64.py:65-69 then :93,98-99): ConvTranspose2d(dim,C,4,1,0) on the
 * 1x1 latent + BN + LeakyReLU, `up`, and the Winograd input transform of the x half of upc2's concat conv in ONE launch - the
 * 4 x 4 x C map is never written.  vec (M, K), row stride ldv; w_kn as dvg_stem_gemm takes it ([KP][16 C], KP in {96, 128});
 * v_next (36, 4 M, C).  Bit-identical to dvg_stem_gemm followed by dvg_winograd_input(upsample = 1).  C % 16 == 0.  ABI 8.  */
int dvg_stem_up_winograd_input(const float* vec, int ldv, const float* w_kn, int KP, const float* scale, const float* shift,
                               float* v_next, int M, int C, int K, int act, float slope, void* stream);
/* Last layer of a decoder block -> first conv of the next block (vgg_64.py:93,98-105: `up` + the x half of the concat conv in
 * Winograd form): mm (36, T, C) of an 8 x 8 layer -> v_next (36, 4 T, C), the input transform of
 * UpsamplingNearest2d(2)(act(scale * A^T M A + shift)); neither the activation nor its upsampled form is written.  Bit-identical
 * to dvg_winograd_output followed by dvg_winograd_input(upsample = 1).  H == W == 8, C % 64 == 0.  ABI 8.               */
int dvg_winograd_output_up_input(const float* mm, const float* scale, const float* shift, float* v_next, int N, int H, int W,
                                 int C, int act, float slope, void* stream);
/* Last layer of an encoder stage (vgg_64.py:51-56, `mp` :49): M (36, T, C) -> the stage's skip tensor y = act(scale * A^T M A
 * + shift) (N,H,W,C) NHWC AND V' (36, T / 4, C), the F(4x4,3x3) input transform of maxpool2x2(y) for the first layer of the
 * next stage.  The pooled tensor itself is never written.  H == W in {16, 32}, C % 64 == 0.  Bit-identical to
 * dvg_winograd_output(+pool) followed by dvg_winograd_input.  ABI 6.
 * y_from (ABI 8): y holds the images [y_from, N) only; the skip tensor of the images before is not stored (y may be NULL when
 * y_from == N).  A rollout keeps the skip tensors of the last conditioning frame alone and discards those of every predicted
 * frame (generate_frames.py:154-157: `h, skip = h` only while i < n_past); v_next always covers all N images.            */
int dvg_winograd_output_pool_input(const float* m, const float* scale, const float* shift, float* y, float* v_next, int N,
                                   int H, int W, int C, int act, float slope, int y_from, void* stream);

/* First encoder layer: Conv2d(nc,Cout,3,1,1)+BN+LReLU with nc in {1..4}
 * (vgg_64.py:23 `vgg_layer(nc, 64)`).  HBM-bound direct convolution.
 * x is NCHW (N,nc,H,W) exactly as the caller's frame tensor (utils.py:90-91);
 * w is the ORIGINAL (Cout,nc,3,3) weight; y is NHWC (N,H,W,Cout).
 * Cout % 64 == 0.  stats as above (optional).                                */
int dvg_conv3x3_first(const float* x_nchw, const float* w_oihw, const float* scale,
                      const float* shift, float* y, float* stats, int N, int H,
                      int W, int nc, int Cout, int act, float slope, void* stream);

/* Last decoder layer: ConvTranspose2d(Cin,nc,3,1,1)+Sigmoid (vgg_64.py:88-92),
 * nc in {1..4}.  x NHWC (N,H,W,Cin), w the ORIGINAL (Cin,nc,3,3) weight,
 * bias [nc], y NCHW (N,nc,H,W).  Cin % 4 == 0, Cin <= 128.                    */
int dvg_convT3x3_last(const float* x, const float* w_iohw, const float* bias,
                      float* y_nchw, int N, int H, int W, int Cin, int nc, int act,
                      void* stream);

/* First dcgan layer Conv2d(nc,Cout,4,2,1)+BN+LReLU, nc in {1..4}
 * (dcgan_64.py:34).  x NCHW, w ORIGINAL (Cout,nc,4,4), y NHWC (N,H/2,W/2,Cout). */
int dvg_conv4x4s2_first(const float* x_nchw, const float* w_oihw, const float* scale,
                        const float* shift, float* y, float* stats, int N, int H,
                        int W, int nc, int Cout, int act, float slope, void* stream);

/* Last dcgan layer ConvTranspose2d(C1+C2,nc,4,2,1)+Tanh|Sigmoid on cat([x,skip])
 * (dcgan_64.py:75-79; dcgan_128.py:80-84).  w ORIGINAL (C1+C2,nc,4,4),
 * y NCHW (N,nc,2H,2W).                                                        */
int dvg_convT4x4s2_last(const float* x, const float* skip, const float* w_iohw,
                        const float* bias, float* y_nchw, int N, int H, int W,
                        int C1, int C2, int nc, int act, void* stream);

/* First step of the two-step last layer: d[px][t] = sum_c in[px][c] * w[t][c] for the P = N*H*W pixels of an
 * NHWC activation (C = 64 or 128 channels) and the T = ks*ks*nc <= 48 tap outputs of the decoder's final
 * ConvTranspose2d (vgg_64.py:90-93, dcgan_64.py:75-79; w is that weight as [(kh,kw,co)][ci]).  HBM-bound MFMA
 * kernel; dvg_convT_gather (below) sums the shifted taps.  P must be a multiple of 16. */
int dvg_pixel_proj(const float* in, const float* w, float* out, long P, int C, int T, void* stream);

/* Second step of the two-step last layers (inference path): y (N,nc,S*H,S*W) NCHW = act(bias + shifted sum of
 * the per-pixel projections d = x . W computed with dvg_gemm_nt_bias_act, d1/d2 [N*H*W][ks*ks*nc] (one per
 * concatenated input, d2 may be NULL), column order (kh, kw, co).  ks = 3: ConvTranspose2d(.,nc,3,1,1)
 * (vgg_64.py:88-92); ks = 4: ConvTranspose2d(.,nc,4,2,1) (dcgan_64.py:75-79).                          */
int dvg_convT_gather(const float* d1, const float* d2, const float* bias, float* y_nchw, int ks,
                     int N, int H, int W, int nc, int act,
                     const int* d2_map /* NULL, or: image n reads block d2_map[n / d2_block] of d2 (shared skip) */,
                     int d2_block, void* stream);

/* ---- GROUPS (ABI 6, time-batched training) ----------------------------------------------------------------------
 * train.py:213-232 is teacher-forced: the encoder calls of a closure (and, once the latent chain has run, its decoder
 * calls) do not depend on each other, so they run as ONE launch over G x B images.  Each reference call is its own
 * BatchNorm batch: the images of such a launch form G consecutive GROUPS of B images, statistics, normalisation and
 * the BatchNorm backward are per group, and every per-channel array of the entry points below becomes [G][C] (group-
 * major).  groups = 1 is the plain call.  Partial-row producers never let a row straddle two groups.
 * ------------------------------------------------------------------------------------------------------------------ */

/* Per-channel sum / sum-of-squares of a [groups * rows][C] (NHWC) tensor, written as
 * groups x dvg_channel_stats_rows(rows) partial rows [2][C] (deterministic slab sums; `rows` = rows PER GROUP).
 * Used for the small BN inputs (encoder head, decoder stem) and the Winograd-form training layers.        */
int dvg_channel_stats_rows(long rows);
int dvg_channel_stats(const float* u, float* stats_partial, long rows, int C, int groups, void* stream);

/* Train-mode BatchNorm2d finalisation (vgg_64.py:9; torch semantics: biased
 * variance for normalisation, unbiased for the running estimate, eps 1e-5):
 * reduces `nrows` partial rows [2][C] (double accumulation) over `count`
 * elements per channel and produces
 *   scale[c] = gamma[c] / sqrt(var_b[c] + eps),  shift[c] = beta[c] - mean[c]*scale[c]
 * and, when running_mean/var != NULL, updates them with `momentum`.
 * save_mean / save_invstd (optional) are kept for the backward pass.
 * groups > 1: `nrows` partial rows and `count` elements PER GROUP; scale / shift / save_* are [G][C]; group_var [G][C]
 * (optional) receives the unbiased variance; running_mean / running_var / num_batches_tracked must be NULL - the running
 * statistics are a recurrence over the groups in call order: dvg_bn_running_update.                                  */
int dvg_bn_finalize(const float* stats_partial, int nrows, const float* gamma,
                    const float* beta, float* scale, float* shift, float* running_mean,
                    float* running_var, float* save_mean, float* save_invstd,
                    int C, double count, float eps, float momentum,
                    int64_t* num_batches_tracked /* may be NULL; += nbt_inc */, int nbt_inc, int groups,
                    float* group_var, void* stream);
/* running <- (1 - m_g) running + m_g stat_g for g = 0 .. groups-1 in order, the update nn.BatchNorm2d applies per call
 * (vgg_64.py:9), from dvg_bn_finalize's save_mean and group_var; m_0 = mom_first, m_{G-1} = mom_last, mom_mid otherwise
 * (the first / last frame of a sequence is encoded once per closure, the others twice: train.py:158-162,184-188,217-221);
 * num_batches_tracked += nbt_inc.                                                                                   */
int dvg_bn_running_update(const float* mean, const float* unbiased_var, int groups, int C, float mom_first, float mom_mid,
                          float mom_last, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                          int nbt_inc, void* stream);

/* y = act(u*scale[c]+shift[c]) elementwise over an NHWC tensor of `npix`
 * pixels (train-mode BN apply + activation), optional fused 2x2 max-pool
 * output (H,W needed only then).  In-place (y == u) is allowed.
 * group_images: images per coefficient group (scale / shift [N / group_images][C]); 0 or N = one group.   */
int dvg_bn_act_apply(const float* u, const float* scale, const float* shift, float* y,
                     float* y_pool, int N, int H, int W, int C, int act, float slope,
                     int group_images, void* stream);

/* ------------------------------------------------------------------ *
 * Dense ends and the recurrent predictor
 * ------------------------------------------------------------------ */

/* out[m][n] = act( (sum_k a[m][k]*w[n][k]) * scale[n % period] + shift[n % period] )
 * Small-M GEMM used for: encoder head Conv2d(512,dim,4,1,0)+BN+Tanh on the 4x4
 * map (vgg_64.py:44-48; K = 8192), decoder stem ConvTranspose2d(dim,512,4,1,0)
 * +BN+LReLU (vgg_64.py:65-69; N = 8192, period = 512), nn.Linear of lstm.py:50,
 * 53-55.  a [M][K] (row stride lda), w [N][K], out [M][N] (row stride ldo).
 * scale/shift may be NULL.  `workspace` must hold M*N*splitk floats when
 * splitk > 1 (NULL allowed when splitk == 1).                                  */
int dvg_gemm_nt_bias_act(const float* a, const float* w, const float* scale,
                         const float* shift, float* out, float* workspace, int M,
                         int N, int K, int lda, int ldo, int period, int splitk,
                         int act, float slope, int accumulate /* out += result */, void* stream);

/* One nn.LSTMCell step (lstm.py:51,68-70; gate order i,f,g,o):
 *   g = W_ih x + b_ih + W_hh h + b_hh ; c' = sig(f)*c + sig(i)*tanh(g~) ;
 *   h' = sig(o)*tanh(c')
 * x,h,c,h_out,c_out: [B][H] contiguous; w_ih,w_hh: [4H][H]; b_ih,b_hh: [4H].
 * In-place state update (h_out == h, c_out == c) is NOT allowed (every
 * workgroup reads all of h).  gates_out (optional, [B][4H]) receives the
 * post-activation gates (i,f,g~,o) for the backward pass.  H % 64 == 0.       */
int dvg_lstm_cell(const float* x, const float* h, const float* c, const float* w_ih,
                  const float* w_hh, const float* b_ih, const float* b_hh,
                  float* h_out, float* c_out, float* gates_out, int B, int H,
                  void* stream);

/* Teacher-forced training (train.py:213-222: every step of a closure feeds the encoding of a GROUND-TRUTH frame to the
 * predictor, so the inputs of all S steps exist before the recurrence starts).  r04: the input halves of both cells, the
 * embedding and the output head run as ONE GEMM each over the S*B rows of a sequence (dvg_gemm_nt_bias_act) and only the
 * recurrent half stays per step:
 *   dvg_lstm_cell_pre: the nn.LSTMCell step of lstm.py:69 with `pre` [B][4H] = W_ih x + b_ih + b_hh given; h, c, h_out,
 *     c_out [B][H], w_hh [4H][H], gates_out (optional) as in dvg_lstm_cell.
 *   dvg_lstm_cell_bwd: one BPTT step of that cell in one launch (what `loss.backward()` of train.py:194,240 does per step
 *     and layer): dh = dh_a + dh_b (from above and from step t+1; either may be NULL), dc (may be NULL), the saved
 *     activated gates [B][4H], c_prev (NULL = the zero initial state of lstm.py:58-63) and c_new ->
 *     dG [B][4H] (gate pre-activation gradients), dc_prev [B][H] and dh_prev [B][H] = dG W_hh (NULL: not needed at the
 *     first step); w_hh_t = W_hh^T [H][4H].  H must be 256 (train.py:37 rnn_size; one gate per 256-wide K slice).     */
int dvg_lstm_cell_pre(const float* pre, const float* h, const float* c, const float* w_hh, float* h_out,
                      float* c_out, float* gates_out, int B, int H, void* stream);
int dvg_lstm_cell_bwd(const float* dh_a, const float* dh_b, const float* dc, const float* gates,
                      const float* c_prev, const float* c_new, const float* w_hh_t, float* dG, float* dc_prev,
                      float* dh_prev, int B, int H, void* stream);

/* The FIRST cell of a time step with the embedding folded in (lstm.py:50,66-70: `embed` is a plain nn.Linear, so
 * W_ih (W_e x + b_e) + b_ih + W_hh h + b_hh = (W_ih W_e) x + W_hh h + bias): x [B][Kx] (row stride ldx floats, 8-byte
 * aligned, Kx even and <= 128), w_x = W_ih W_e as [4H][Kxp] (Kxp % 4 == 0, zero padded), bias = W_ih b_e + b_ih + b_hh
 * [4H], both folded by the caller once per weight version.  Inference path (no gates output).                    */
int dvg_lstm_cell_x(const float* x, int ldx, int Kx, const float* h, const float* c, const float* w_x, int Kxp,
                    const float* w_hh, const float* bias, float* h_out, float* c_out, int B, int H, void* stream);

/* Decoder stem ConvTranspose2d(dim,512,4,1,0)+BN+LReLU on a 1x1 map (vgg_64.py:65-69, dcgan_64.py:62-67), eval mode:
 * out[m][n] = act((sum_k vec[m][k] * w_kn[k][n]) * scale[n % period] + shift[n % period]); w_kn is the GEMM weight
 * TRANSPOSED to [KP][N] (N = 16*512 in NHWC flatten order, N % 32 == 0; KP = 96 or 128 rows, rows K.. zero), K = dim. */
int dvg_stem_gemm(const float* vec, int ldv, const float* w_kn, int KP, const float* scale, const float* shift,
                  float* out, int ldo, int M, int N, int K, int period, int act, float slope, void* stream);

/* ------------------------------------------------------------------ *
 * Sparse variational GP trigger (gp_models.py:10-24 + gpytorch 0.3.x
 * WhitenedVariationalStrategy / GaussianLikelihood / MultivariateNormal;
 * equations of record in DESIGN.md §GP).  One workgroup per latent dim.
 * ------------------------------------------------------------------ */

/* Predictive distribution q(f(x)) for D independent 1-D GPs, M inducing points.
 *   h            [B][D]   latent codes; GP d sees column d (the reference's
 *                         h.transpose(0,1).view(D,B,1), train.py:225)
 *   z            [D][M]   inducing inputs
 *   var_mean     [D][M]   variational mean m
 *   chol_var     [D][M][M] variational Cholesky factor (lower part is used)
 *   mean_const   [D], outputscale [D], lengthscale [D]  (already soft-plus'ed)
 *   noise        [D] or NULL: likelihood noise added to the variance/covariance
 *                (GaussianLikelihood.__call__, generate_frames.py:131,170)
 * Outputs (any may be NULL):
 *   mean [D][B]; var [D][B] (marginal variance; train-mode clamp at 0 when
 *   `train_mode` != 0); sample [D][B] = mean + chol(Sigma)*eps with
 *   eps [D][B] supplied by the caller (MultivariateNormal.rsample);
 *   cov [D][B][B] full predictive covariance (eval mode only).
 *   kl [D] (train mode: KL(q(u)||p(u))).
 * train_mode is a flag word: bit 0 = train-mode prediction (diagonal variance with the clamp, KL); bit 1 = outputscale /
 * lengthscale / noise point at the RAW parameters (covar_module.raw_outputscale, base_kernel.raw_lengthscale,
 * noise_covar.raw_noise) and the kernel applies soft-plus itself (noise: + the 1e-4 floor of GaussianLikelihood).
 * Arithmetic (ABI 6): inputs and outputs are fp32; INSIDE the kernel the covariance assembly, chol(K_ZZ), the triangular
 * solves, the k(x,x) - A^T A cancellation, the KL terms, chol(Sigma) and the sample are computed in fp64 whenever the fp64
 * working set fits the 160 KiB of LDS (M = 40: every B <= 128; above B = 95 the covariance is kept as a packed lower
 * triangle on top of the dead K_ZZ factors), otherwise in fp32.
 * dvg_gp_precision(B, M, cov||sample) returns 64 or 32 accordingly (DVG_GP_FP32=1 in the environment forces 32 for A/B
 * runs); dvg_gp_lds_bytes the LDS bytes of the variant that will run.
 * Limits: M <= 64, B <= 128 and dvg_gp_lds_bytes(B, M, cov||sample) <= 160 KiB. */
int dvg_gp_precision(int B, int M, int need_cov);
size_t dvg_gp_lds_bytes(int B, int M, int need_cov);
int dvg_gp_predict(const float* h, const float* z, const float* var_mean,
                   const float* chol_var, const float* mean_const,
                   const float* outputscale, const float* lengthscale,
                   const float* noise, const float* eps, float* mean, float* var,
                   float* sample, float* cov, float* kl, int B, int D, int M,
                   int train_mode, float jitter,
                   int param_period /* ABI 8; 0 = D.  Workgroup d reads the parameters (z, var_mean, chol_var, mean_const,
                                       outputscale, lengthscale, noise) of latent dim d % param_period: the S time steps of a
                                       training closure side by side as D = S * g_dim dims on ONE parameter set */,
                   int step_group /* ABI 8; <= 1: one workgroup per column of h.  k > 1 (train-mode outputs only, k <= S):
                                     workgroup g * param_period + q takes the steps [g k, g k + k) of latent dim q as ONE
                                     problem of up to k * B points - K_ZZ, its factor and the KL term once per k steps.
                                     Same outputs, same layout; mean / var bit-identical to k = 1 */,
                   void* stream);
/* the step_group the host side uses for a time-batched training call of S steps x P latent dims x B points (1 = no grouping) */
int dvg_gp_step_group(int B, int S, int P, int M);

/* ------------------------------------------------------------------ *
 * Backward (training) entry points: what `loss.backward()` (train.py:170,194,240)
 * runs for the modules above.  Data gradients of the dense convs reuse the
 * forward implicit-GEMM kernels with re-packed weights:
 *   dgrad(Conv2d 3x3)        = dvg_conv3x3_bn_act_v2    with the transposed/flipped k16 pack of W
 *   dgrad(Conv2d 4x4 s2)     = dvg_convT4x4s2_bn_act_v2 with the transposed k16 pack of W
 *   dgrad(ConvTranspose 4x4) = dvg_conv4x4s2_bn_act_v2  with the plain k16 pack of W
 * ------------------------------------------------------------------ */

/* BatchNorm + activation (+ 2x2 max-pool) backward, pass 1:
 *   dp = (dy + scatter_maxpool(dyp)) * act'(y)          (written to `dp`, NHWC)
 *   partial[r] = { sum dp, sum dp*u } per channel, r < dvg_bn_act_bwd_rows(...)
 * y = forward output (post activation), u = conv output before BN.  dy or dyp
 * may be NULL (not both); dyp != NULL selects the pooled variant.
 * groups: N must be a multiple; partial holds groups x dvg_bn_act_bwd_rows(N / groups, ...) rows, group-major.   */
int dvg_bn_act_bwd_rows(int N, int H, int W, int pool);
int dvg_bn_act_bwd_reduce(const float* dy, const float* dyp, const float* y, const float* u,
                          float* dp, float* partial, int N, int H, int W, int C, int act,
                          float slope, int groups, void* stream);
/* pass 2: per-channel coefficients of du = A*dp + B*u + Cc (train: batch-statistics
 * BN backward; eval: plain affine), plus dgamma, dbeta, dbias (any may be NULL).
 * groups > 1: nrows / count per group; mean, invstd, coef*, dgamma, dbeta, dbias are [G][C] (the parameter gradients per
 * group: sum them with dvg_colsum; accumulate must be 0).                                                          */
int dvg_bn_bwd_finalize(const float* partial, int nrows, const float* gamma, const float* mean,
                        const float* invstd, float* coefA, float* coefB, float* coefC,
                        float* dgamma, float* dbeta, float* dbias, int C, double count,
                        int train, int accumulate /* dgamma / dbeta / dbias += */, int groups, void* stream);
/* pass 3: du = A[c]*dp + B[c]*u + Cc[c] over n elements (n %% C == 0); du may alias dp.  `sum` (optional second output,
 * same shape, not aliasing du): sum_mode 1: sum = du, 2: sum += du, 0: unused - d(addend) of the decoder calls of one
 * time step that share a skip half (train.py:227-231) is collected there instead of by separate additions.           */
int dvg_affine3_apply(const float* dp, const float* u, const float* A, const float* B,
                      const float* Cc, float* du, long n, int C, float* sum, int sum_mode,
                      int groups /* A, B, Cc are [groups][C]; group = run of n / groups elements */, void* stream);
/* dst[b] = sum over the groups g with map[g] == b, in ascending g (deterministic), of src[g]: src [groups][block_elems],
 * dst [blocks][block_elems], map = device ints.  The adjoint of the shared addend (addend_map above): d(addend block) =
 * the sum of d(pre-activation) over the decoder calls that read it.  block_elems % 4 == 0.  ABI 6.                    */
int dvg_group_sum(const float* src, const int* map, float* dst, int groups, int blocks, long block_elems, void* stream);
/* dpre = dy * act'(y), flat tensors (last layers, nn.Linear+Tanh).               */
int dvg_act_bwd(const float* dy, const float* y, float* dpre, long n, int act, float slope,
                void* stream);
/* nn.UpsamplingNearest2d(2) backward: dx (N,H,W,C) = 2x2 block sums of dxu (N,2H,2W,C). */
int dvg_upsample2x_bwd(const float* dxu, float* dx, int N, int H, int W, int C, void* stream);
/* out[c] = sum_r a[r][c]  (bias gradients)                                       */
int dvg_colsum(const float* a, float* out, int rows, int C, int accumulate /* out += */, void* stream);
/* out[i] = sum_s partial[s][i], n %% 4 == 0                                       */
int dvg_reduce_partials(const float* partial, float* out, int S, long n, void* stream);

/* Weight gradient of the dense convs as an fp32-MFMA GEMM over pixels:
 *   partial[s][tap][Cout][Cin], s < dvg_conv_wgrad_splits(...); reduce with
 *   dvg_reduce_partials, then dvg_unpack_conv(T)_weight gives the nn layout.
 * mode = DVG_MODE_*; x/skip/upsample_x describe the forward input exactly as in the
 * forward call (H,W = forward input grid); dout = gradient w.r.t. the conv output.
 * C1, C2, Cout multiples of 64.                                                  */
int dvg_conv_wgrad_splits(int mode, int N, int H, int W, int Cin, int Cout);
int dvg_conv_wgrad(int mode, const float* x, const float* skip, const float* dout,
                   float* partial, int N, int H, int W, int C1, int C2, int Cout,
                   int upsample_x, void* stream);
/* The same over `items` (1..8) uses of ONE layer with identical shapes - dW = sum_i dOut_i (x) In_i, e.g. the time steps
 * / decoder calls of train.py:213-232, which share their weights: x / skip / dout are HOST arrays of `items` device
 * pointers (skip == NULL when C2 == 0).  The GEMM K dimension grows `items`-fold, so the K-split partial slabs
 * (dvg_conv_wgrad_splits_multi(..., items) of them) are written and reduced once per `items` uses.                */
int dvg_conv_wgrad_splits_multi(int mode, int N, int H, int W, int Cin, int Cout, int items);
int dvg_conv_wgrad_multi(int mode, int items, const float* const* x, const float* const* skip,
                         const float* const* dout, float* partial, int N, int H, int W, int C1, int C2,
                         int Cout, int upsample_x, void* stream);

/* The weight gradient of a 3x3 / stride-1 Conv2d in Winograd F(4x4,3x3) form (vgg_layer, vgg_64.py:5-15, under
 * train.py:240 `loss.backward()`; maps up to 32x32, Cin and Cout multiples of 128): a quarter of the direct form's
 * multiply-adds.   dg = G^T [ sum_tiles (A dY A^T) .* (B^T d B) ] G:
 *   dvg_winograd_wgrad_operands  V (36, t_total, Cin) = B^T x B of the layer input x NHWC (N,H,W,Cin) and
 *                                dM (36, t_total, Cout) = A dY A^T of d(out) NHWC (N,H,W,Cout), written at tile offset
 *                                t_off: the N*(H/4)*(W/4) tiles of several uses of ONE layer are concatenated along the
 *                                tile axis (x or dy may be NULL to skip that operand); t_total % 64 == 0, rows no use
 *                                wrote must be zero
 *   dvg_winograd_wgrad_gemm      partial (S, 36, Cout, Cin): per position the product dM^T V over the tiles, K-split into
 *                                S = dvg_winograd_wgrad_splits(t_total, Cin, Cout) slabs (0 = unsupported shape)
 *   dvg_winograd_wgrad_reduce    packed (9, Cout, Cin) = G^T (sum_s partial[s]) G: ONE slab in dvg_conv_wgrad's layout,
 *                                which dvg_wgrad_finish places into the parameter's gradient                              */
int dvg_winograd_wgrad_operands(const float* x, const float* dy, float* v, float* dm, int N, int H, int W, int Cin,
                                int Cout, long t_total, long t_off, void* stream);
int dvg_winograd_wgrad_splits(long t_total, int Cin, int Cout);
int dvg_winograd_wgrad_gemm(const float* dm, const float* v, float* partial, long t_total, int Cin, int Cout, void* stream);
/* The same with V given per use: `v_items` is a HOST array of `items` (1..8) device pointers to (36, tiles_per_item, Cin)
 * buffers - the input transforms the forward pass computed for the layer (dvg_winograd_input, m = 4), kept instead of
 * recomputed; dM is still one concatenated (36, items * tiles_per_item, Cout) buffer.  tiles_per_item % 64 == 0.          */
int dvg_winograd_wgrad_gemm_items(const float* dm, const float* const* v_items, int items, long tiles_per_item,
                                  float* partial, int Cin, int Cout, void* stream);
int dvg_winograd_wgrad_reduce(const float* partial, int S, float* packed, int Cin, int Cout, void* stream);

/* Finish of a weight gradient IN PLACE in the parameter's gradient buffer (train.py:240 `loss.backward()` accumulates
 * into .grad; here the kernel that finishes the gradient does it, so no per-use gradient tensor and no accumulation
 * launch exist):  dst = beta * dst + sum_s partial[s],  partial = the S packed [KH*KW][Cout][Cin] slabs of dvg_conv_wgrad,
 * dst addressed in the nn layout, optionally as the channel slice [c_lo, c_lo + Cin) of a weight with Ctot input channels
 * (the x / skip halves of a concat conv):
 *   kind 0: Conv2d weight (Cout, Ctot, KH, KW); kind 1: ConvTranspose2d weight (Ctot, Cout, KH, KW) (spatially flipped,
 *   as dvg_unpack_convT_weight); kind 2: plain packed [KH*KW][Cout][Cin] (contiguous).  Cin % 4 == 0.                */
int dvg_wgrad_finish(const float* partial, int S, float* dst, int kind, int kh, int kw, int cout, int cin, int ctot,
                     int c_lo, float beta, void* stream);
/* dW (Cout, Ctot, 3, 3)[:, c_lo : c_lo + C1] = beta * dW + 2x2 window sums of dK4, the gradient w.r.t. the 4x4 stride-2
 * transposed-conv kernel that `nearest-x2 upsample + conv3x3` (vgg_64.py:93,98-105) runs as; dk4_packed [16][Cout][C1] is
 * the reduced packed output of dvg_conv_wgrad(DVG_MODE_CONVT4S2).                                                    */
int dvg_k4_to_w3(const float* dk4_packed, float* dw, int cout, int c1, int ctot, int c_lo, float beta, void* stream);

/* Weight gradient of the thin first / last layers (ks = 3: stride 1, ks = 4: stride 2, pad 1):
 *   dW[c][ci][a][b] = sum dout[n][oy][ox][c] * inp[n][ci][S*oy+a-1][S*ox+b-1]
 * inp NCHW (N,nc,Hi,Wi), dout NHWC (N,Ho,Wo,C); partial[rows][C][nc*ks*ks] with
 * rows = dvg_wgrad_thin_rows(ks,N,Hi,Wi).  First layers: inp = frame, dout = du.
 * Last (transposed) layers: inp = dpre (frame side), dout = the layer INPUT x.     */
int dvg_wgrad_thin_rows(int ks, int N, int Hi, int Wi);
int dvg_wgrad_thin(const float* inp_nchw, const float* dout_nhwc, float* partial, int ks, int N,
                   int Hi, int Wi, int nc, int C, void* stream);

/* Compositing step of the Moving-MNIST generator (data/moving_mnist.py:86-90) fused with the layout change of
 * utils.normalize_data (utils.py:86-95): out (T,B,1,S,S) = min(1, sum_d sprite[ids[b,d]] placed at pos[b,d,t] =
 * (sy,sx)), digits added in index order.  sprites (n_sprites,D,D) fp32; ids (B,num_digits) int32; pos
 * (B,num_digits,T,2) int32 with 0 <= sy,sx <= S-D (the host computes the trajectories, moving_mnist.py:49-85). */
int dvg_moving_mnist_compose(const float* sprites, const int* ids, const int* pos, float* out, int n_sprites, int T,
                             int B, int num_digits, int image_size, int digit_size, void* stream);

/* Evaluation metrics of utils.eval_seq (utils.py:220-234): per (sample, channel) image, SSIM as
 * skimage.measure.compare_ssim computes it with its defaults (7x7 uniform window, sample covariance, data range 2
 * for float images, mean over the valid window positions) and PSNR as compare_psnr (data range 1 for non-negative
 * ground truth).  gt / pred: n_images contiguous HxW fp32 images (an NCHW frame batch is B*C of them). */
int dvg_eval_frames(const float* gt, const float* pred, float* ssim, float* psnr, int n_images, int H, int W,
                    void* stream);

/* Fused Adam step over one flat parameter group (train.py:95-106: torch.optim.Adam(lr=0.002) with default
 * betas (0.9, 0.999) and eps 1e-8; torch.optim.Adam arithmetic, non-amsgrad): param / exp_avg / exp_avg_sq are
 * updated in place, `step` is the 1-based step count used for the bias corrections; when `step_dev` is not NULL the
 * count is read from that device int instead (a captured hipGraph replays with a new count every iteration). */
int dvg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, const int* step_dev, void* stream);

/* nn.LSTMCell backward, elementwise part: gate pre-activation gradients dG [B][4H] and
 * dc_prev [B][H] from dh', dc' (either may be NULL), the saved activated gates, c, c'.
 * The GEMM parts (dx = dG W_ih, dW_ih = dG^T x, ...) go through dvg_gemm_nt_bias_act. */
/* out[m][n] (+)= sum_r a[r][m] b[r][n] (a [R][M], b [R][N], row strides lda / ldb / ldo): dW = dY^T X of nn.Linear / nn.LSTMCell
 * (lstm.py:50-55) over the R = steps x batch rows of a BPTT pass in ONE launch, no transposed copies; colsum0 / colsum1
 * (NULL = none) receive the column sums of a - the bias gradient(s) - overwritten or, colsum_accumulate, added to.  ABI 8. */
int dvg_gemm_tn(const float* a, const float* b, float* out, float* colsum0, float* colsum1, int R, int M, int N,
                int lda, int ldb, int ldo, int accumulate, int colsum_accumulate, void* stream);
int dvg_lstm_gates_bwd(const float* dh, const float* dc, const float* gates, const float* c_prev,
                       const float* c_new, float* dG, float* dc_prev, int B, int H, void* stream);

/* Train-mode GP backward (gradients of dvg_gp_predict(train_mode=1) outputs mean / var
 * (without likelihood noise) / kl): upstream gmean [D][B], gvar [D][B], gkl [D] (any may
 * be NULL = zero) -> dh [B][D], dz [D][M], dm [D][M], dls [D][M][M] (lower), dc, ds, dell
 * [D] w.r.t. the soft-plus'ed hyper-parameters.  One workgroup per latent dim, all in LDS.  fp64 inside, fp32 I/O
 * (ABI 6).  r04: the data points are processed in chunks of dvg_gp_bwd_chunk(B, M) so that the fp64 working set fits the
 * 160 KB of LDS for every B <= 128 at M <= 40 (M = 40: one pass up to B = 71, two chunks of 64 at B = 128; until r03 B > 71
 * fell back to fp32 arithmetic); dvg_gp_bwd_precision still returns 32 where not even small chunks fit (M = 64, B > 40). */
int dvg_gp_bwd_precision(int B, int M);
size_t dvg_gp_bwd_lds_bytes(int B, int M);
int dvg_gp_bwd_chunk(int B, int M);
int dvg_gp_train_bwd(const float* h, const float* z, const float* var_mean, const float* chol_var,
                     const float* mean_const, const float* outputscale, const float* lengthscale,
                     const float* gmean, const float* gvar, const float* gkl, float* dh, float* dz,
                     float* dm, float* dls, float* dc, float* ds, float* dell, int B, int D, int M,
                     float jitter, int param_period /* ABI 8, as dvg_gp_predict */,
                     int step_group /* ABI 8, as dvg_gp_predict.  The PARAMETER gradients are written per workgroup:
                                       dz, dm [G * P][M], dls [G * P][M][M], dc, ds, dell [G * P] with P = param_period (or D)
                                       and G = ceil(S / max(step_group, 1)) - each row already summed over its workgroup's
                                       steps; dvg_sum_steps_multi adds the G rows per latent dim up (G = 1: nothing left) */,
                     void* stream);
/* dst_k[i] = sum over s < S of src_k[s * n_k + i] for count <= 8 tensors in one launch (host arrays of device pointers / sizes):
 * the per-(step, latent dim) parameter gradients of a closure's S side-by-side time steps -> one gradient per parameter.  ABI 8. */
int dvg_sum_steps_multi(const float* const* src, float* const* dst, const long* n, int count, int S, void* stream);

/* VariationalELBO(likelihood, gp_layer, num_data, combine_terms=True)(pred, target) with the GaussianLikelihood's
 * expected log-probability (train.py:102,112; called at train.py:164-169,225-226): per latent dim d
 *   elbo_d = (1/B) sum_b [ -((y_db - mean_db)^2 + var_db) / (2 sig2_d) - log(sig2_d)/2 - log(2 pi)/2 ] - kl_d / num_data,
 *   sig2_d = softplus(raw_noise_d) + 1e-4.
 * mean, var [D][B] and kl [D] are dvg_gp_predict's train-mode outputs; target is addressed as
 * target[d * t_stride_d + b * t_stride_b] (the reference passes h_target.transpose(0,1), a strided view).
 * dvg_gp_elbo_bwd: given gelbo [D] -> gmean, gvar [D][B], gkl [D], gtarget [D][B] (may be NULL), graw_noise [D]
 * (soft-plus chain included).  ABI 5. */
int dvg_gp_elbo(const float* mean, const float* var, const float* kl, const float* target, long t_stride_d,
                long t_stride_b, const float* raw_noise, float* elbo, int B, int D, int num_data,
                int noise_period /* ABI 8; 0 = D: workgroup d reads raw_noise[d % noise_period] */, void* stream);
int dvg_gp_elbo_bwd(const float* mean, const float* var, const float* kl, const float* target, long t_stride_d,
                    long t_stride_b, const float* raw_noise, const float* gelbo, float* gmean, float* gvar,
                    float* gkl, float* gtarget, float* graw_noise, int B, int D, int num_data, int noise_period,
                    void* stream);

/* ------------------------------------------------------------------ *
 * Small elementwise helpers on the path
 * ------------------------------------------------------------------ */

/* (N,C,H,W) contiguous -> NHWC and back (skip tensors handed to / taken from
 * callers that insist on contiguous NCHW).                                    */
int dvg_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, void* stream);
int dvg_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, void* stream);

/* ------------------------------------------------------------------ *
 * Debug hooks (tools/diag_*.py).  Not part of the product path: they hand the next launches of
 * one kernel family a device buffer that every workgroup fills with clock64()/wall_clock64()
 * phase stamps.  buf == NULL (the default) disables stamping; the kernels then pay one
 * uniform branch.  Not thread-safe: set, launch, synchronise, reset from ONE host thread.
 * ------------------------------------------------------------------ */
void dvg_debug_set_clockbuf(void* buf, unsigned records);        /* conv_igemm2 kernels: 8 x u64 per workgroup */
void dvg_debug_set_gp_clockbuf(void* buf, unsigned records);     /* gp_predict_kernel: 12 x u64 per workgroup  */
void dvg_debug_set_wgrad_clockbuf(void* buf, unsigned records);  /* wgrad_igemm_kernel: 4 x u64 per workgroup  */

/* ---- GPtrigger_gen's bookkeeping on the device (generate_frames.py:220-232,275,283-296; ABI 8) ------------------------------
 * The reference pulls the predictive variance to the host at every step (`.cpu().numpy()` + np.linalg.norm), slides a 12-long
 * window in numpy and branches in Python.  Here the norm, the window statistics, the decision and the branch select are three
 * tiny kernels, so the loop has no host round trip and can be captured as a hipGraph; the logs are read back once.
 *   dvg_gp_var_norms      norms[b] = || var[:, b] ||_2, var (D,B): the warm-up's recorded values for every sample (:275)
 *   dvg_gp_trigger_step   value = norm of sample `col` (:230 reads sample [3]); ctx (window floats) <- [ctx[1:], value] (:231);
 *                         threshold = mean(ctx) + coef * std(ctx) (population std, float32 like the reference's arrays; :288);
 *                         *flag = flags[slot] = value > threshold; values[slot], thresholds[slot] logged
 *   dvg_gp_trigger_replay the decisions (and thresholds) ANOTHER batch index would take on a recorded sequence of n main-loop
 *                         values from its own initial window ctx0 - the main loop's value is that of sample [3] whatever the
 *                         index (:230), so an index whose decisions equal the recorded rollout's IS that rollout
 *   dvg_gp_trigger_select vec (B,D) = *flag ? sample_db^T : h_pred; state_out[k] = *flag ? state_old[k] : state_new[k] - a
 *                         triggered step decodes the GP sample and does NOT step the LSTM (:289-296); n_state <= 8 tensors of
 *                         state_elems floats (host arrays of device pointers)                                            */
int dvg_gp_var_norms(const float* var, float* norms, int D, int B, void* stream);
int dvg_gp_trigger_step(const float* var, int D, int B, int col, float* ctx, int window, float coef, int* flag, float* values,
                        float* thresholds, int* flags, int slot, void* stream);
int dvg_gp_trigger_replay(const float* values, int n, const float* ctx0, int window, float coef, int* flags, float* thresholds,
                          void* stream);
int dvg_gp_trigger_select(const int* flag, const float* sample_db, const float* h_pred, float* vec, int D, int B, int n_state,
                          long state_elems, const float* const* state_old, const float* const* state_new,
                          float* const* state_out, void* stream);

/* ---- losses of the step closures and their gradients in one pass (train.py:188,223,227-239; ABI 8) ---------------------------
 * dvg_frame_losses: pred [S][K][n] - per time step the K decoder calls in the reference's call order (x_pred, x_target_pred,
 * x_pred_gp, :227-232) -, target [S][n] (the step's ground-truth frame): sums[k] = sum over steps and elements of
 * (pred - target)^2 (= n x the sum over steps of nn.MSELoss of call k) and dpred = 2 w[k] (pred - target), the gradient of
 * sum_k w[k] sums[k] - the frame terms of `loss` (:239) with w[k] = weight_k / n - in ONE read of pred / target.  w on the
 * device; partial: dvg_frame_losses_blocks(S * n) * K floats of workspace; n % 4 == 0, K <= 3.  Deterministic.
 * dvg_mse_sum_grad: *sum = sum (a - b)^2, da = 2 scale (a - b) (NULL: value only) for the latent MSE terms.              */
int dvg_frame_losses_blocks(long elems);
int dvg_frame_losses(const float* pred, const float* target, float* sums, float* dpred, long n, int S, int K, const float* w,
                     float* partial, void* stream);
int dvg_mse_sum_grad(const float* a, const float* b, float* sum, float* da, long n, float scale, void* stream);

/* optimizer.zero_grad() of adjacent parameter groups of the gradient arena as ONE fill (train.py:201-203 zeroes three modules),
 * and the device-side Adam step counts t0..t3 (NULL: none) advanced by one in the same launch (dvg_adam_step's step_dev): a
 * captured iteration needs no one-element increment launch per optimiser step.  ABI 8.                                   */
int dvg_zero_tick(float* g, long n, int* t0, int* t1, int* t2, int* t3, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DVG_HIP_H */
