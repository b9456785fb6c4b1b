/* sprk.h — C ABI of libsprk.so: the MI355X (gfx950) kernels behind spr_pick's joint
 * denoise+detect hot path.
 *
 * The reference (nextpyp/spr_pick) has no FFI on this path: it calls stock torch.nn /
 * NumPy from Python.  Each entry point below therefore replaces a *torch or NumPy call
 * site* of the reference (cited per function as file:line under /root/reference) and
 * is what a maintainer binds with ctypes from the reference's Python modules — see
 * INTEGRATION.md for the stubs.
 *
 * Conventions
 *   - all tensors are fp32, contiguous NCHW, DEVICE pointers (unless marked host);
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on it and
 *     re-entrant per stream; nothing is allocated or synchronised inside;
 *   - the caller owns every buffer incl. the workspace `ws` (size from *_ws_bytes);
 *   - return 0 on success, a negative SPRK_E* code otherwise; sprk_last_error() gives
 *     a thread-local message.
 */
#ifndef SPRK_H
#define SPRK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPRK_OK 0
#define SPRK_EINVAL (-1)   /* bad argument / unsupported geometry */
#define SPRK_EWORKSPACE (-2) /* workspace too small */
#define SPRK_ELAUNCH (-3)  /* hip launch error */

#define SPRK_ACT_NONE 0
#define SPRK_ACT_LEAKY 1   /* LeakyReLU(0.1) */
#define SPRK_ACT_RELU 2

/* Geometry of one 2-D convolution  y[N,Cout,Hout,Wout] = conv(in[N,C1+C2,Hin,Win], w).
 * The conv input is the channel concatenation of up to two sources:
 *   src1: x  [N,C1,Hin,Win]           (up1 = 0)
 *         x  [N,C1,Hin/2,Win/2]       (up1 = 1: nearest x2 upsampling fused into the load)
 *   src2: x2 [N,C2,Hin,Win]           (C2 may be 0, x2 NULL)
 * which is torch.cat((Upsample(2)(a), skip), 1) of the reference's U-Nets
 * (models/joint_network_v2.py:212-227) without materialising it.
 * Padding is asymmetric and explicit: rows/cols outside [0,Hin)x[0,Win) read as zero,
 * Hout/Wout are given by the caller.  ShiftConv2d (joint_network_v2.py:565-584) is
 * pad_top = 2*(KH/2), pad_left = KW/2, Hout = Hin, Wout = Win. */
typedef struct sprk_conv_geom {
    int32_t N, C1, C2, Hin, Win, up1;
    int32_t Cout, Hout, Wout;
    int32_t KH, KW, stride, dil, pad_top, pad_left;
    int32_t dtype;   /* SPRK_DT_*: precision of the MFMA OPERANDS (see below); tensors are fp32 either way */
} sprk_conv_geom;

/* Operand precision of the matrix-core convolutions (BASELINE configs[4]: "fp16 MFMA conv").
 * SPRK_DT_F32 (0, the default): v_mfma_f32_16x16x4_f32 — exact fp32 products, what every parity figure of the fp32
 * path refers to.  SPRK_DT_BF16 / SPRK_DT_F16: the two operands of each product (activation or gradient, and
 * weight) are rounded to bf16 / fp16 on their way into v_mfma_f32_16x16x32_{bf16,f16}; products and sums are fp32;
 * inputs, outputs, master weights and weight gradients stay fp32 tensors.  It is a REQUEST: layers the 16-bit
 * kernels do not cover (strided / dilated / tiny detector layers, < 33 output channels, maps below 16x16) run in
 * fp32; sprk_conv16_launch_count() tells which path ran.  Error bound per output: 2u * sum_k |a_k w_k| with
 * u = 2^-8 (bf16) / 2^-11 (fp16). */
#define SPRK_DT_F32 0
#define SPRK_DT_BF16 1
#define SPRK_DT_F16 2
#define SPRK_DT_MASK 0xff
/* or-ed into dtype: take the 16-bit kernel for every layer it covers, also where the library's own choice would be
 * the fp32 Winograd kernel because it is faster there (wide 3x3 layers on planes >= 64x64); used by the parity tests */
#define SPRK_DT_FORCE 0x100
/* or-ed into dtype: run this call on the plain per-output-element kernels (no MFMA, no Winograd): the on-device
 * cross-check the parity tests use.  Per call — the library keeps no mode switch. */
#define SPRK_DT_NAIVE 0x200
/* or-ed into dtype: PIN the algorithm to the layer's structure — the choice between the Winograd and the direct kernel
 * (whose fp32 results differ in the last bits) then depends only on kernel size, channel counts and the divisibility
 * of the plane (H % 8, W % 32 | H % 16, W % 16), never on how many images or workgroups the call has.  Whole-image and
 * halo-tiled inference set it, so that a window of a micrograph is computed by exactly the arithmetic the whole
 * micrograph gets (bit-identical score maps, hence identical picks; tests/test_gpu_pipeline.py). */
#define SPRK_DT_PIN 0x400
/* 16-bit ACTIVATION tensors in HBM (round 4; BASELINE configs[4]).  With SPRK_DT_BF16 / SPRK_DT_F16 alone the tensors of
 * a call are fp32 and only the MFMA operands are rounded on their way in.  SPRK_DT_X16: the activation INPUTS of the call —
 * x and x2 of a forward call, gy of backward-data, x, x2 AND gy of backward-weight — are tensors of the operand type
 * (the `const float *` parameters then point at 2-byte elements).  SPRK_DT_Y16: the activation OUTPUT — y of forward,
 * gin (and mask_y) of backward-data — is such a tensor: the fp32 accumulator is rounded once, at the store.  Weights,
 * biases, weight gradients and all arithmetic stay fp32.  Only the kernels built for it take these calls
 * (sprk_conv2d_storage16 tells which of a layer's three calls exist); any other geometry returns SPRK_EINVAL. */
#define SPRK_DT_X16 0x8000
#define SPRK_DT_Y16 0x10000
/* element types of the tensors of an elementwise call, 4 bits each: SPRK_DT_F32 / _BF16 / _F16 */
#define SPRK_IO2(a, b) ((a) | ((b) << 4))
#define SPRK_IO3(a, b, c) ((a) | ((b) << 4) | ((c) << 8))

/* Optional fused epilogue of sprk_conv2d_fwd, applied in this order:
 *   v = acc (+ res[n,co,oy+res_off,ox+res_off])            res: [N,Cout,res_h,res_w]
 *   v = scale ? v*scale[co] + shift[co] : v + (bias ? bias[co] : 0)
 *   v = act(v)
 * (residual add + eval-mode BatchNorm affine + ReLU of ResidA, feature_extractor.py:384-416).
 * up2 != 0: y is [N,Cout,2*Hout,2*Wout] and every value is stored to its 2x2 block — the
 * nn.Upsample(scale_factor=2, mode="nearest") that follows the conv in the reference's decoder
 * blocks (joint_network_v2.py:69-97), fused into the store. */
typedef struct sprk_conv_epilogue {
    const float *bias, *scale, *shift, *res;
    int32_t res_h, res_w, res_off;
    int32_t act;
    int32_t up2;
} sprk_conv_epilogue;

/* ABI version: bumped whenever a signature or a struct of this header changes incompatibly (100 = round 1; 300: the
 * geometry carries `dtype`, sprk_act_bwd / sprk_bn_* take the stride / group arguments, the head and prepared-weight
 * entry points exist).  A binding must refuse a library whose sprk_version() differs from the header it was
 * written against, and may compare sprk_struct_bytes(0 | 1 | 2) with its own sizeof(sprk_conv_geom |
 * sprk_conv_epilogue | sprk_reduce_item). */
#define SPRK_ABI_VERSION 410
const char *sprk_last_error(void);
int sprk_version(void);
size_t sprk_struct_bytes(int which);
/* number of HIP kernel launches issued by this library since load (diagnostics) */
long sprk_launch_count(void);
/* number of convolutions (forward or backward-data) that took the Winograd F(2x2,3x3) kernel (diagnostics) */
long sprk_wino_launch_count(void);
/* number of convolution launches (forward, backward-data or backward-weight) that ran on the 16-bit-operand
 * kernels (diagnostics / tests) */
long sprk_conv16_launch_count(void);
/* ... of those, the backward-weight launches */
long sprk_wgrad16_launch_count(void);

/* ---- convolution: replaces F.conv2d/F.pad/crop of nn.Conv2d / ShiftConv2d ------------
 * reference: models/joint_network_v2.py:565-584 (ShiftConv2d), :33-153 (all U-Net convs),
 * models/joint_network_v2_shallow.py:33-150, models/feature_extractor.py:285,335-343,
 * models/classifier.py:11.  w is the reference layout [Cout][C1+C2][KH][KW]. */
size_t sprk_conv2d_fwd_ws_bytes(const sprk_conv_geom *g);
int sprk_conv2d_fwd(const float *x, const float *x2, const float *w, float *y,
                    const sprk_conv_geom *g, const sprk_conv_epilogue *ep,
                    void *ws, size_t ws_bytes, void *stream);
/* gin[N,C1+C2,Hin,Win] = d loss / d (concatenated conv input), given gy = d loss / d (pre-activation y) */
size_t sprk_conv2d_bwd_data_ws_bytes(const sprk_conv_geom *g);
int sprk_conv2d_bwd_data(const float *gy, const float *w, float *gin, const sprk_conv_geom *g,
                         void *ws, size_t ws_bytes, void *stream);
/* ---- fused per-pixel head (inference) ------------------------------------------------------------------------------
 * out = W3 . lrelu(W2 . lrelu(W1 . f + b1) + b2) + b3 in ONE launch, LeakyReLU(0.1): the output_block + output_conv of
 * the blind-spot U-Net (K0 = N1 = 384, N3 = 2: models/joint_network_v2.py:123-153, 241-244) and of the sigma net
 * (K0 = N1 = 96, N3 = 1: models/joint_network_v2_shallow.py).  The middle width is 96.  A 128-pixel tile stays in the
 * workgroup from the input features to the outputs: the two hidden tensors (2 x 25.8 GB at 4096^2) never reach HBM.
 * f [B,K0,HW], w1 [N1,K0], w2 [96,N1], w3 [N3,96] (the reference's [Cout,Cin,1,1] layouts), out [B,N3,HW]; HW % 128 == 0.
 * Same channel order in every sum as three sprk_conv2d_fwd calls, another tiling: results agree to fp32 rounding.
 * Training keeps the three calls (the hidden activations are needed by the weight gradients). */
size_t sprk_head1x1_fwd_ws_bytes(int K0, int N1);
int sprk_head1x1_fwd(const float *f, const float *w1, const float *b1, const float *w2, const float *b2,
                     const float *w3, const float *b3, float *out, int B, int K0, int N1, int N3, long HW, void *ws,
                     size_t ws_bytes, void *stream);

/* The blind-spot head reading the rotated stack itself: d [4B,C=96,P,P] (the output of decode_block_1) -> out [B,N3,P,P].
 * Shift2d((1,0)) + chunk(4) + rotate({0,270,180,90}) + cat(dim=1) (joint_network_v2.py:230-239, i.e.
 * sprk_unrot4_shift_concat_fwd) are the address computation of the kernel's input gather: the [B,384,P,P] feature tensor
 * (25.8 GB at 4096^2) is neither written nor read.  Same arithmetic as unrot + sprk_head1x1_fwd: bit-identical. */
int sprk_head1x1_unrot_fwd(const float *d, const float *w1, const float *b1, const float *w2, const float *b2,
                           const float *w3, const float *b3, float *out, int B, int C, int P, int N3, void *ws,
                           size_t ws_bytes, void *stream);

/* The same with the activation backward of the layer that PRODUCED this convolution's input fused in:
 *   gin = (d loss / d conv input) * act'(mask_y),   mask_y = the saved conv input [N,C1+C2,Hin,Win] (post-activation
 * output of that layer), mask_act = its SPRK_ACT_*.  gin is then that layer's pre-activation gradient and its own
 * sprk_act_bwd pass (read gy, read y, write gpre) is not needed; only its bias gradient remains (sprk_act_bwd with
 * SPRK_ACT_NONE: one read).  In conv -> conv chains (joint_network_v2.py:33-97: every block of the U-Nets) this removes
 * two thirds of the activation-backward traffic.  The Winograd kernel applies the mask in its output transform; the
 * other kernels finish with an in-place pass, so the call is valid for every geometry without upsampled input. */
int sprk_conv2d_bwd_data_masked(const float *gy, const float *w, float *gin, const sprk_conv_geom *g,
                                const float *mask_y, int mask_act, void *ws, size_t ws_bytes, void *stream);
/* bit 0 / 1 / 2: the forward / backward-data / backward-weight call of this layer exists for 16-bit activation tensors
 * (SPRK_DT_X16 / SPRK_DT_Y16; g->dtype carries the operand type, ep the forward epilogue or NULL) */
int sprk_conv2d_storage16(const sprk_conv_geom *g, const sprk_conv_epilogue *ep);
/* gw[Cout][C1+C2][KH][KW] = d loss / d w  (overwritten, not accumulated) */
size_t sprk_conv2d_bwd_weight_ws_bytes(const sprk_conv_geom *g);
int sprk_conv2d_bwd_weight(const float *x, const float *x2, const float *gy, float *gw,
                           const sprk_conv_geom *g, void *ws, size_t ws_bytes, void *stream);
/* ---- prepared weights: one launch per training step instead of one per convolution call ------------------------
 * Every convolution call re-lays its weights out in its workspace before the main kernel (k-chunked slabs, Winograd
 * G g G^T, 16-bit slabs): ~100 small launches per training step, 2-3 % of it.  A caller whose weights change once per
 * step (an optimiser) can hoist them:
 *   1. once per layer and direction, with a workspace that it KEEPS: sprk_conv2d_fwd_wprep / _bwd_data_wprep describe
 *      the transform that call would run (no launch; item->kind == 0: that path has none);
 *   2. after every weight update: sprk_prepare_weights(items, n) — all transforms in one launch per 40 items;
 *   3. the convolution calls pass the same workspace and SPRK_DT_WPREP in dtype: the transform launch is skipped.
 * Same device code either way: results are bit-identical to the plain calls.  The items are opaque.
 * A SPRK_DT_WPREP call may add SPRK_DT_WPREP_KIND(item.kind): the library then checks that the kernel path the call takes
 * reads a transform of that kind and returns SPRK_EINVAL otherwise (instead of reading a slab in another layout). */
#define SPRK_DT_WPREP 0x800
#define SPRK_DT_WPREP_KIND(kind) (((kind) & 7) << 12)
#define SPRK_DT_WPREP_KIND_OF(dtype) (((dtype) >> 12) & 7)
typedef struct sprk_wprep_item {
    const float *w;
    void *dst;
    int32_t kind, blocks;
    int32_t p[12];
} sprk_wprep_item;
int sprk_conv2d_fwd_wprep(const float *w, const sprk_conv_geom *g, const sprk_conv_epilogue *ep, void *ws,
                          size_t ws_bytes, sprk_wprep_item *item);
int sprk_conv2d_bwd_data_wprep(const float *w, const sprk_conv_geom *g, void *ws, size_t ws_bytes, sprk_wprep_item *item);
int sprk_prepare_weights(const sprk_wprep_item *items, int n, void *stream);

/* gpre[N,C,H,W] = gy * act'(y) where y is the saved POST-activation output (gpre may alias gy;
 * with act == NONE and up2 == 0 nothing is written to gpre and it may be NULL); if gbias != NULL
 * also gbias[c] = sum over n,h,w of gpre (conv bias gradient).  up2 != 0: gy and y are the
 * [N,C,2H,2W] tensors of a conv with the fused-upsample epilogue; gy is first summed over each
 * 2x2 block (backward of nn.Upsample).  ws: C*nsplit floats. */
size_t sprk_act_bwd_ws_bytes(int N, int C, int HW);
/* g_image_stride: elements between consecutive images of gy, 0 = dense (C*H*W, or 4*C*H*W with up2).  A larger
 * stride reads gy out of a channel slice of a wider tensor (the two halves of a concat layer's input gradient are
 * handed on as views); y and gpre are always dense. */
/* io = SPRK_IO3(type of gy, type of y, type of gpre): every tensor fp32 or the call's 16-bit type (all bf16 or all fp16);
 * gbias and the arithmetic are fp32.  With act == NONE and different gy / gpre types the call is a change of storage
 * type (+ the bias sum). */
int sprk_act_bwd(const void *gy, const void *y, void *gpre, float *gbias, int act,
                 int N, int C, int H, int W, int up2, long g_image_stride, int io,
                 void *ws, size_t ws_bytes, void *stream);
/* ---- deferred second-stage reductions ------------------------------------------------------
 * Backward-weight and the bias gradient are two-stage sums: the main kernel leaves per-workgroup partial sums in
 * ws and a small second kernel adds them in a fixed order.  A training step runs ~80 of those second kernels
 * (5 us of dispatch each).  The *_partial entry points run only the main kernel and describe the pending sum in
 * *item; sprk_reduce_items then finishes any number of them in one launch per 48 items.  Between the two calls the
 * item's ws must stay untouched (give every call its own ws) and dst is undefined.  item->kind == SPRK_RED_NONE:
 * the call already produced its final result (kernels without a partial stage).  The plain entry points above are
 * the *_partial call followed by sprk_reduce_items on its one item: results are bit-identical either way. */
#define SPRK_RED_NONE 0
#define SPRK_RED_ROWS 1   /* dst[i] = sum_p src[p * n + i],                        i < n          */
#define SPRK_RED_COLS 2   /* dst[i] = sum_p src[i * parts + p],                    i < n          */
#define SPRK_RED_WGRAD 3  /* dst[co * K + k] = sum_p src[(p * K + k) * CoutP + co], co < Cout, k < K */
typedef struct sprk_reduce_item {
    const float *src;
    float *dst;
    int kind, parts, n, K, Cout, CoutP;
} sprk_reduce_item;
int sprk_conv2d_bwd_weight_partial(const float *x, const float *x2, const float *gy, float *gw,
                                   const sprk_conv_geom *g, void *ws, size_t ws_bytes,
                                   sprk_reduce_item *item, void *stream);
int sprk_act_bwd_partial(const void *gy, const void *y, void *gpre, float *gbias, int act,
                         int N, int C, int H, int W, int up2, long g_image_stride, int io,
                         void *ws, size_t ws_bytes, sprk_reduce_item *item, void *stream);
/* every sum: four interleaved chains over p (p mod 4), combined as (s0 + s1) + (s2 + s3) */
int sprk_reduce_items(const sprk_reduce_item *items, int n, void *stream);

/* split the gradient of a fused (upsample2(a) ++ b) conv input: ga[N,C1,H/2,W/2] = 2x2 sums
 * of gin[:, :C1], gb[N,C2,H,W] = gin[:, C1:].  (autograd of nn.Upsample + torch.cat) */
int sprk_concat_up_bwd(const float *gin, float *ga, float *gb, int N, int C1, int C2, int H, int W,
                       int up1, void *stream);

/* ---- U-Net plumbing (HBM-bound) --------------------------------------------------------
 * Shift2d((1,0)) + MaxPool2d(2): models/joint_network_v2.py:27-30, models/utility.py:46-72;
 * shift = 0 gives the plain MaxPool2d(2) of the sigma net (joint_network_v2_shallow.py). */
/* io = SPRK_IO2(type of x, type of y) / SPRK_IO3(type of gy, type of x, type of gx) */
int sprk_shift_maxpool2_fwd(const void *x, void *y, int NC, int H, int W, int shift, int io, void *stream);
/* act != SPRK_ACT_NONE: x is the post-activation output of a convolution consumed by this pool only; gx is multiplied
 * by act'(x) and is then that convolution's PRE-activation gradient (its sprk_act_bwd pass reduces to the bias sum) */
int sprk_shift_maxpool2_bwd(const void *gy, const void *x, void *gx, int NC, int H, int W, int shift, int act,
                            int io, void *stream);
/* rotate(x,{0,90,180,270}) + cat(dim=0): joint_network_v2.py:198-200, utils/data.py:43-68.
 * x [B,C,P,P] -> y [4B,C,P,P]; bwd sums the four inverse rotations into gx. */
int sprk_rot4_stack_fwd(const float *x, float *y, int B, int C, int P, void *stream);
int sprk_rot4_stack_bwd(const float *gy, float *gx, int B, int C, int P, void *stream);
/* Shift2d((1,0)) + chunk(4) + rotate({0,270,180,90}) + cat(dim=1): joint_network_v2.py:230-239.
 * d [4B,C,P,P] -> f [B,4C,P,P] */
/* io = SPRK_IO2(type of the input, type of the output) */
int sprk_unrot4_shift_concat_fwd(const void *d, void *f, int B, int C, int P, int io, void *stream);
int sprk_unrot4_shift_concat_bwd(const void *gf, void *gd, int B, int C, int P, int io, void *stream);

/* ---- BatchNorm2d (+ optional ReLU), detector: joint_network_v2.py:547,558;
 * feature_extractor.py:287-288,320-324,338-346,412-414.
 * train: batch statistics (biased var for normalisation, unbiased for the running update,
 * momentum 0.1) saved to save_mean/save_invstd [groups][C]; eval: running statistics.
 * groups: the batch is `groups` independent passes of N / groups images stacked along N (the reference calls the
 * detector once per pass: patches, then their flipped copies): each group is normalised with its own batch
 * statistics, the running averages are updated group after group and the parameter gradients are summed in
 * group order, as if the module had been called once per pass.  groups = 1: plain BatchNorm2d.
 * ws (sprk_bn_ws_bytes): per-slice fp64 partial sums of the two-kernel reduction. */
size_t sprk_bn_ws_bytes(int N, int C, int HW, int groups);
int sprk_bn_train_fwd(const float *x, float *y, const float *gamma, const float *beta,
                      float *running_mean, float *running_var, float *save_mean, float *save_invstd,
                      int N, int C, int HW, int groups, float momentum, float eps, int relu,
                      void *ws, size_t ws_bytes, void *stream);
int sprk_bn_eval_fwd(const float *x, float *y, const float *gamma, const float *beta,
                     const float *running_mean, const float *running_var,
                     int N, int C, int HW, float eps, int relu, void *stream);
/* gy is d loss / d y (post-ReLU if relu); y is the saved output (for the ReLU mask). */
int sprk_bn_train_bwd(const float *gy, const float *x, const float *y, const float *gamma,
                      const float *save_mean, const float *save_invstd,
                      float *gx, float *ggamma, float *gbeta,
                      int N, int C, int HW, int groups, int relu, void *ws, size_t ws_bytes, void *stream);

/* ---- per-pixel maths of the pipeline ---------------------------------------------------
 * reparameterize: z = mu + eps * A^2 on out_stats [B,2,H,W] (joint_network_v2.py:469-475) */
int sprk_reparam_fwd(const float *out_stats, const float *eps, float *z, int B, int HW, void *stream);
/* g_out_stats [B,2,HW] (overwritten): d/dmu = gz, d/dA = gz * eps * 2A */
int sprk_reparam_bwd(const float *gz, const float *out_stats, const float *eps, float *g_out_stats,
                     int B, int HW, void *stream);
/* _sigmoid: clamp(sigmoid(x), 1e-4, 1-1e-4)  (denoiser_v2.py:32-34) */
int sprk_sigmoid_clamp_fwd(const float *x, float *p, long n, void *stream);
int sprk_sigmoid_clamp_bwd(const float *gp, const float *x, float *gx, long n, void *stream);
/* Small fused pieces of the training step's tail (ABI 410).  Each replaces a chain of framework launches by one: a
 * kernel of a replayed step costs ~5 us whatever it does, and these chains were a quarter of the step's launches.
 *
 * ResidA's residual (models/feature_extractor.py:384-416: x = x[:, :, edge:-edge, edge:-edge], optionally
 * x[:, :, ::s, ::s]; y = y + x):   out[nc][i][j] = y[nc][i][j] + x[nc][off + s i][off + s j]   (y == NULL: the crop
 * alone, for the 1x1 projection); NC = N * C planes, y / out [NC][Ho][Wo], x [NC][Hx][Wx].  crop_embed_bwd is the
 * gradient with respect to x: g scattered to those positions, zero elsewhere (the gradient with respect to y is g). */
int sprk_crop_add_fwd(const float *y, const float *x, float *out, long NC, int Ho, int Wo, int Hx, int Wx, int off, int stride,
                      void *stream);
int sprk_crop_embed_bwd(const float *g, float *gx, long NC, int Ho, int Wo, int Hx, int Wx, int off, int stride, void *stream);
/* Noise level of an image from the estimator's map (denoiser_v2.py:392-402; NoiseValue.UNKNOWN_VARIABLE):
 *   z[b] = mean(est[b]) - 4;  noise_std[b] = softplus(z[b]) + 1e-3   (softplus with torch's threshold 20)
 * est [B][HW]; z is kept for the backward pass:  g_est[b][i] = g[b] * sigmoid(z[b]) / HW.  One workgroup per image
 * (meant for patches; the whole-micrograph path keeps the framework's reduction). */
int sprk_noise_std_fwd(const float *est, float *noise_std, float *z, int B, int HW, void *stream);
int sprk_noise_std_bwd(const float *g, const float *z, float *g_est, int B, int HW, void *stream);
/* The training loss of the joint pipeline (denoiser_v2.py:516-519):
 *   consis[0] = mean((p - flip(pf))^2);   final[b] = alpha * loss_out[b] + (1 - alpha) * pred[0] + w_consis * consis[0]
 * p, pf: [B][1][H][W] scores of the batch and of the flipped batch, pf NOT yet flipped back (axis 0: along W, 1: along
 * H); loss_out [B] per-image denoising loss; pred [1] the PU loss.  joint_loss_bwd: g_final [B] ->
 *   g_loss_out[b] = alpha g_final[b];  g_pred[0] = (1 - alpha) S;  gp = w_consis S 2 (p - flip(pf)) / n;  gpf = -flip(gp)
 * with S = sum_b g_final[b], n = B H W.  One workgroup, sums in a fixed order. */
int sprk_joint_loss_fwd(const float *loss_out, const float *pred, const float *p, const float *pf, float *final_loss,
                        float *consis, int B, int H, int W, int axis, float alpha, float w_consis, void *stream);
int sprk_joint_loss_bwd(const float *g_final, const float *p, const float *pf, float *g_loss_out, float *g_pred, float *gp,
                        float *gpf, int B, int H, int W, int axis, float alpha, float w_consis, void *stream);
/* PU detection loss and its gradient in one launch (utils/losses.py:303-349: BCE on the labelled patches + slack *
 * binomial generalised-expectation penalty on the unlabelled ones):
 *   y[i] >= 0: labelled with target y[i];  y[i] == -1: unlabelled;  N = number of unlabelled, n_lab of labelled
 *   cls = sum_lab -(y log p + (1-y) log(1-p)) / max(n_lab, 1)
 *   mu = sum_unl p, var = sum_unl p (1-p);  q = softmax_k(-(mu - k)^2 / (2 (var + 1e-7))), k = 0..N
 *   loss[0] = cls - slack * sum_k log_binom[N][k] q[k];   gp[i] = d loss / d p[i]
 * log_binom: [B+1][B+1] table, row N = binom.logpmf(0..N; N, tau) (the host computes it once per (B, tau)).
 * p in (0, 1) (sigmoid_clamp output).  One workgroup; sums in a fixed order. */
int sprk_pu_loss(const float *p, const float *y, const float *log_binom, int B, float slack,
                 float *loss, float *gp, void *stream);
/* Adam update of many parameter tensors in ONE launch (train.py:128-140: Adam(lr, betas = (0.9, 0.99)), eps 1e-8, no
 * weight decay, no amsgrad; the formulas of torch.optim.Adam):
 *   t = step_in[0] + 1;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2
 *   p -= lr[0] / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps);   step_out[0] = t
 * items / start live in DEVICE memory (built once: the addresses of a training run do not change): item i owns
 * workgroups start[i] .. start[i+1]-1, 1024 elements each.  lr, step_in, step_out: device scalars (step_in != step_out),
 * so the call can sit in a captured graph and the learning-rate ramp needs no host sync. */
typedef struct sprk_adam_item {
    float *p;
    const float *g;
    float *m, *v;
    long n;
} sprk_adam_item;
int sprk_adam_multi(const sprk_adam_item *items, const int *start, int n_items, int n_blocks,
                    const float *lr, const float *step_in, float *step_out,
                    float beta1, float beta2, float eps, void *stream);
/* SSDN likelihood + posterior mean (denoiser_v2.py:405-424 noise model, :449-462 likelihood, :514 mean):
 *   var_x = A^2, var_y = var_x + var_n
 *   style SPRK_NOISE_GAUSSIAN: s = noise_std[b],                       var_n = s^2
 *   style SPRK_NOISE_POISSON : s = sqrt(max(mu, 1e-3) * noise_std[b]), var_n = s^2   (per pixel; Hasinoff 2012)
 *   nll = (x-mu)^2/var_y + log var_y - 0.05 s      -> loss[b] = mean over HW
 *   pme = (x var_x + mu var_n)/var_y ;  model_std = sqrt(var_x)
 * x [B,1,HW], out_stats [B,2,HW], noise_std [B] (the remapped estimate softplus(.-4)+1e-3);
 * pme/model_std [B,HW] (may be null); noise_std_map [B,HW]: s per pixel, written for POISSON only (may be null); loss [B].
 * ws: B*nblk floats of partial sums. */
enum { SPRK_NOISE_GAUSSIAN = 0, SPRK_NOISE_POISSON = 1 };
size_t sprk_ssdn_ws_bytes(int B, int HW);
int sprk_ssdn_fwd(const float *x, const float *out_stats, const float *noise_std,
                  float *loss, float *pme, float *model_std, float *noise_std_map, int style, int B, int HW,
                  void *ws, size_t ws_bytes, void *stream);
/* gloss [B] = d L / d loss[b];  g_out_stats [B,2,HW] (overwritten), g_noise_std [B] (d L / d noise_std[b]) */
int sprk_ssdn_bwd(const float *gloss, const float *x, const float *out_stats, const float *noise_std,
                  float *g_out_stats, float *g_noise_std, int style, int B, int HW,
                  void *ws, size_t ws_bytes, void *stream);

/* ---- training-patch feed ----------------------------------------------------------------
 * replaces MicrographDataset.__getitem__ (train branch) — datasets/micrograph.py:60-122 — plus the
 * DataLoader collation (train.py:1085-1091): PIL crop box (x-P/2, y-P/2, x+P/2, y+P/2) with zero
 * fill outside the image, RandomHorizontalFlip, to_tensor (uint8 -> float32/255; float images
 * unchanged) and the CWH->CHW permute that transposes the patch:
 *   out[b,0,u,v] = mic[image_b][y_b - P/2 + v][x_b - P/2 + (flip_b ? P-1-u : u)]
 * mics: all micrographs packed back to back in one device buffer of `dtype`
 * (SPRK_MIC_U8 | SPRK_MIC_F32); offsets[n_mics] (elements, device int64); dims[n_mics][2] =
 * (rows, cols) device int32; items[B][4] = (image, x, y, flip) device int32; out [B,1,P,P]. */
#define SPRK_MIC_U8 0
#define SPRK_MIC_F32 1
int sprk_gather_patches(const void *mics, int dtype, const long *offsets, const int *dims, const int *items,
                        float *out, int n_mics, int B, int P, void *stream);

/* ---- 2-D greedy non-maximum suppression ------------------------------------------------
 * replaces non_maximum_suppression(x, r, contam=set(), threshold) —
 * utils/algorithms.py:59-103, call site train.py:564.  Exact greedy semantics incl. the
 * reference's clip-to-H/W border behaviour; ties: score desc, then flat index desc.
 * scores [H,W] device; out_scores[max_out], out_xy[max_out][2] (x=col, y=row) device,
 * written in pick order (descending score).
 * out_count: device int32[2]: [0] = number of picks found (if it exceeds max_out the
 * stored list is incomplete: call again with a larger max_out), [1] = pixels still
 * undecided after `rounds` relaxation launches.  The suppression fixed point is reached
 * by repeated launches (no host sync inside): when [1] != 0 call again with resume = 1
 * (state is kept in ws) until it is 0; typical score maps need < 10 rounds. */
size_t sprk_nms2d_ws_bytes(int H, int W, int max_out);
int sprk_nms2d(const float *scores, int H, int W, int r, float threshold,
               float *out_scores, int32_t *out_xy, int32_t *out_count, int max_out,
               int rounds, int resume, void *ws, size_t ws_bytes, void *stream);

/* ---- in-library kernel timing (bench.py's roofline leg) --------------------------------
 * sprk_prof_enable(mask): bit k set = every launch of kernel class k is bracketed by HIP events
 * on its own stream (an event pair costs the GPU front end a few microseconds, so the timed
 * region of bench.py enables the dominant class only); sprk_prof_collect synchronises those
 * events and returns, per kernel class (0 = conv_mfma_kernel<4, 6, *>, the widest direct
 * forward / backward-data tile shape; 1 = conv_wgrad_mfma_kernel (all instantiations); 2 = the
 * other conv_mfma_kernel instantiations and wino_conv_kernel<3>; 3 = wino_conv_kernel<6>, the
 * Winograd kernel of the 96-channel 3x3 layers, the largest single kernel of a training step;
 * 4 = wino_wgrad_kernel, the Winograd backward-weight kernel of the largest layers),
 * the launch count, the summed duration in ms and the summed algorithmic (direct-convolution)
 * FLOPs. */
void sprk_prof_enable(int mask);
int sprk_prof_collect(int kclass, long *launches, double *ms, double *flops);
/* the same plus the summed ALGORITHMIC HBM bytes of the launches (every tensor read and written once, fp32): classes
 * 5 (16-bit-operand forward / backward-data kernels) and 6 (16-bit-operand backward-weight kernels) are HBM-bound
 * and are priced in GB/s; 0 for the other classes */
int sprk_prof_collect_bytes(int kclass, long *launches, double *ms, double *flops, double *bytes);

#ifdef __cplusplus
}
#endif
#endif /* SPRK_H */
