/* ctdet_hip.h -- C ABI of the MI355X-native CenterNet hot path (libctdet_hip.so).
 *
 * Drop-in boundary: these entry points are what the reference's own native binding for this path
 * would bind.  In the reference the binding is the pybind11 module `detectron2._C`
 * (detectron2/layers/csrc/vision.cpp:70-117) whose functions take at::Tensor; here every function
 * takes plain device pointers + explicit shapes + a hipStream_t (as void*), the caller owns all
 * buffers (no hidden allocation, workspace sizes are queried), and every function returns 0 or a
 * negative errno-style code with the message available from ctdet_last_error().
 *
 * Device layout: activations NHWC ("pixel rows" of `*_stride` elements), f16 (throughput mode) or
 * f32 (exact mode); conv weights pre-packed with ctdet-side layout (see ctdet_conv_desc).
 * All launches go to the stream passed in; nothing synchronises; all functions are graph-capturable, and since ABI 7 every
 * launch is a KERNEL: no hipMemsetAsync / hipMemcpyAsync is issued on the caller's stream (a captured step holds kernel nodes
 * only -- a memset node of a replayed training step was seen to run out of order; ctdet_decode_status, which copies a status
 * word to the host and is not part of a step, is the exception).
 */
#ifndef CTDET_HIP_H
#define CTDET_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* CTDET_DT_F16X3 is a COMPUTE mode only (ctdet_conv_desc.compute_dtype): tensors are f32, contractions run as three f16
 * products per term on the f16 matrix pipe (see ctdet_conv_desc). */
enum ctdet_dtype { CTDET_DT_F16 = 0, CTDET_DT_F32 = 1, CTDET_DT_U8 = 2, CTDET_DT_F16X3 = 3 };
enum ctdet_act { CTDET_AC_NONE = 0, CTDET_AC_RELU = 1, CTDET_AC_SIGMOID_CLAMP = 2 };

/* Geometry of one conv-shaped contraction.
 * compute_dtype F16: x is f16 NHWC, weights f16 packed [Cout_pad][Kpad], k ordered per `korder`,
 *   Kpad = roundup(R*S*Cin, 32), Cout_pad = roundup(Cout, tile) with tile = ctdet_conv_cout_tile(Cout);
 *   MFMA f16 x f16 -> f32 accumulate; y is out_dtype (f16 or f32).
 * compute_dtype F32 (the reference's own arithmetic): x, y f32, weights f32 packed [Cout_pad][Kpad], k tap-major,
 *   Kpad = roundup(R*S*Cin, 16), Cout_pad = roundup(Cout, tile); v_mfma_f32_16x16x4_f32 (bit-for-bit a k-ordered
 *   f32 fma chain) when Cin and the pixel strides are multiples of 4, a scalar f32 chain otherwise.
 * compute_dtype F16X3 (f32 tensors at the f16 matrix rate): x, y, residual f32 exactly as for F32; the weights are the F32
 *   packed image passed through ctdet_split_weights (same size: each group of 4 k = {w_hi[4], w_lo[4]} f16, w_hi = f16(w),
 *   w_lo = f16(w - w_hi)); the kernels split every activation the same way in registers and evaluate
 *   a*w = a_hi*w_hi + a_lo*w_hi + a_hi*w_lo with v_mfma_f32_16x16x32_f16 and f32 accumulation (dropped term ~2^-22; a
 *   K-long dot product is as accurate as the f32 chain).  Values must lie inside the f16 range (|x| < 65504). */
typedef struct ctdet_conv_desc {
  int32_t B, H, W, Cin, in_stride;
  int32_t Cout, Ho, Wo, out_stride;
  int32_t R, S, stride, pad, dil;
  int32_t Kpad, Cout_pad;
  int32_t compute_dtype, out_dtype;
  int32_t act;          /* ctdet_act, applied after scale/bias/residual */
  int32_t res_stride;   /* pixel stride of the residual tensor (same dtype as y) */
  float clamp_lo, clamp_hi; /* for CTDET_AC_SIGMOID_CLAMP */
  int32_t korder;       /* f16 packing order of k: 0 = tap-major  k = (r*S+s)*Cin + c;
                           1 = chunk-major k = ((c/32)*R*S + r*S+s)*32 + c%32 (needs Cin % 32 == 0; keeps the taps
                           of one 32-channel chunk adjacent in time => L2-friendly; not for ctdet_dcnv2_fwd) */
                        /* 2 (F16X3, 3x3 / stride 1 / pad 1, Cin % 16 == 0, maps divisible by 8x32; ctdet_conv2d_fwd only):
                           "pair" packing for the halo-resident kernel -- per cout row, per 16-channel chunk, five tap pairs
                           (taps 2s, 2s+1; the tenth tap is zero), each 128 bytes: X = for q = 0..3 {w_hi[tap 2s][4q..4q+3],
                           w_hi[tap 2s+1][4q..4q+3]} f16, then Y = the same of w_lo; Kpad = Cin / 16 * 160 (4-byte units) */
                        /* 3 (as 2, Cin % 32 == 0): "cross-chunk pair" packing -- per cout row, per PAIR of 16-channel chunks
                           (A, B) nine 128-byte {X, Y} steps: steps 0..3 taps (2s, 2s+1) of A, step 4 tap 8 of A with tap 8
                           of B, steps 5..8 taps (2(s-5), 2(s-5)+1) of B; no zero tap; Kpad = Cin / 32 * 288 */
  int32_t in_dil;       /* 0/1 = none.  >1: the input is read as if zero-stuffed by this factor (input-gradient of a
                           strided conv expressed as a conv over dY); Ho/Wo may then exceed the formula by < in_dil */
} ctdet_conv_desc;

const char* ctdet_last_error(void);
int32_t ctdet_abi_version(void);
/* Kernel-selection switches for tests and tuning (bit set = that specialised kernel is NOT used / the alternative form IS
 * used); process-wide, default 0.  The launch path reads this word, never the environment. */
enum ctdet_tuning {
  CTDET_TUNING_NO_HALO = 1, CTDET_TUNING_NO_WIN = 2, CTDET_TUNING_DCN_MIXED = 4, CTDET_TUNING_NO_WGRAD_WINDOW = 8,
  CTDET_TUNING_NO_COL2IM_WINDOW = 16, CTDET_TUNING_NO_F32_DCN_WINDOW = 32,
  CTDET_TUNING_DCN_WINDOW_V1 = 64, CTDET_TUNING_NO_SMALL_GRID_TILES = 128
};
int32_t ctdet_set_tuning_flags(uint32_t flags);
uint32_t ctdet_get_tuning_flags(void);
int32_t ctdet_conv_cout_tile(int32_t cout);

/* f32 packed conv weights (the compute_dtype F32 image, n floats, n % 4 == 0, 16-byte aligned) -> the F16X3 image of the
 * same size (see ctdet_conv_desc).  Replaces nothing in the reference (its convs are cuDNN fp32); it is the pack step of
 * the F16X3 compute mode. */
int32_t ctdet_split_weights(const float* w_packed_f32, void* w_split, int64_t n, void* stream);

/* y = act(conv(x, w) * scale + bias + residual).  Replaces torch.nn.Conv2d (+BatchNorm2d eval +ReLU
 * +residual add) at detectron2/modeling/backbone/dla.py:59-73,86-94,212-220,249-259 and the head convs at
 * detectron2/modeling/meta_arch/centernet.py:115-121.  scale/bias/residual may be NULL. */
int32_t ctdet_conv2d_fwd(const ctdet_conv_desc* d, const void* x, const void* w_packed, const float* scale,
                         const float* bias, const void* residual, void* y, void* stream);

/* Root of the DLA tree (dla.py:86-94): 1x1 conv over torch.cat(xs, dim=channels) without materialising the
 * concat.  xs[j] are NHWC tensors of identical B,H,W with cins[j] channels and pixel stride strides[j]
 * (nsrc <= 4); d->Cin = sum(cins), d->in_stride is ignored. */
int32_t ctdet_conv1x1_cat_fwd(const ctdet_conv_desc* d, const void* const* xs, const int32_t* cins,
                              const int32_t* strides, int32_t nsrc, const void* w_packed, const float* scale,
                              const float* bias, const void* residual, void* y, void* stream);

/* Modulated deformable conv v2 forward, batched and fused (sampling -> MFMA, no columns buffer).
 * Replaces _C.modulated_deform_conv_forward (detectron2/layers/csrc/vision.cpp:85-88,
 * deform_conv_cuda.cu:804-927, deform_conv.py:214-234).  offset_mask is the raw f32 output of the
 * 27-channel conv_offset_mask conv, [B*Ho*Wo, om_stride]: ch 2k = dh, 2k+1 = dw of tap k, ch 18+k = mask
 * logit (sigmoid applied here) or, with mask_is_prob != 0, the already-sigmoided mask the reference's
 * functional API passes.  scale/bias fold the conv bias and the following BatchNorm. */
int32_t ctdet_dcnv2_fwd(const ctdet_conv_desc* d, const void* x, const float* offset_mask, int32_t om_stride,
                        int32_t mask_is_prob, const void* w_packed, const float* scale, const float* bias, void* y, void* stream);

/* ctdet_dcnv2_fwd in the f16x3 mode that ALSO writes the sampled columns (modulated_deformable_im2col, kernel.cu:786-868:
 * cols_out f32 [M][9*Cin], k = tap*Cin + c, value = mask * bilinear(x)) as a by-product of the forward pass -- training keeps
 * them for the weight gradient (deform_conv_cuda.cu:1098-1106) instead of sampling the layer a second time in the backward.
 * Only where the LDS-window kernel serves the layer: ctdet_dcnv2_cols_supported(d, x, y) -> 1 (3x3/s1/p1, map divisible by 8x16,
 * Cin % 16 == 0, packed rows % 64 == 0, 16-byte aligned tensors); otherwise call ctdet_dcnv2_fwd and ctdet_dcn_cols. */
int32_t ctdet_dcnv2_cols_supported(const ctdet_conv_desc* d, const void* x, const void* y);
int32_t ctdet_dcnv2_fwd_cols(const ctdet_conv_desc* d, const void* x, const float* offset_mask, int32_t om_stride,
                             int32_t mask_is_prob, const void* w_packed, const float* scale, const float* bias, void* y,
                             float* cols_out, void* stream);

/* DCNv2 together with the conv that produces its offsets and mask logits (the reference's DCN wrapper: conv_offset_mask 3x3 /
 * s1 / p1, Cin -> 27, then modulated_deform_conv on the same input; detectron2/layers/deform_conv.py and the wrapper the
 * CenterNet project uses) in ONE kernel: the offset conv is evaluated from the LDS window the sampling uses, its output never
 * goes to memory unless om_out is given (f32 [M][om_out_stride], channels 0..27 written; a backward pass needs it).
 * w_off_packed: ctdet_pack_weights of the [27,Cin,3,3] weight with rows_pad 32, chunk-major; b_off: 28 f32 (27 used).
 * f16 compute, Cout <= 64 packed to 64 rows, maps divisible by 8x16, Cin % 32 == 0: ctdet_dcnv2_offset_supported(d) says
 * whether a descriptor qualifies (otherwise: ctdet_conv2d_fwd + ctdet_dcnv2_fwd). */
int32_t ctdet_dcnv2_offset_supported(const ctdet_conv_desc* d);
int32_t ctdet_dcnv2_offset_fwd(const ctdet_conv_desc* d, const void* x, const void* w_off_packed, const float* b_off,
                               float* om_out, int32_t om_out_stride, const void* w_packed, const float* scale,
                               const float* bias, void* y, void* stream);

/* CenterNet.preprocess_image (centernet.py:173-185) + ImageList.from_tensors padding
 * (detectron2/structures/image_list.py:58-130): img is [B,3,H,W] (u8 or f32, CHW, batch stride given in
 * elements), out is NHWC [B,Hp,Wp,out_stride] with channels 0..2 = (x/255 - mean)/std, the rest 0.
 * border > 0: out is [B,Hp+2*border,Wp+2*border,out_stride] whose zero frame the caller cleared once; only the
 * interior is written (the 7x7 stem then runs as a pad-0 conv with no bounds checks).  out_stride: a multiple of 8, or 4
 * with f32 output (4-channel pixels halve the stem's K). */
int32_t ctdet_preprocess(const void* img, int32_t img_dtype, void* out, int32_t out_dtype, int32_t B, int32_t H,
                         int32_t W, int32_t Hp, int32_t Wp, int64_t img_batch_stride, const float* mean3,
                         const float* std3, int32_t out_stride, int32_t border, void* stream);

/* Fused CenterNet head (detectron2/modeling/meta_arch/centernet.py:115-121,151-154): for every head h
 *   y[h] = act_h( W2_h * relu(conv3x3_p1(x, W1_h) + b1_h) + b2_h ),   hidden width 256,
 * without materialising the hidden maps.  x: f16 NHWC [B,H,W,in_stride] (Cin % 32 == 0, H % 8 == 0, W % 16 == 0);
 * w1: the nheads 3x3 weights concatenated along Cout and packed chunk-major (ctdet_pack_weights korder 1) [nheads*256]
 * [9*Cin]; b1 f32 [nheads*256]; w2[h]: f16 row-major [round_up(cout,16)][256] (zero rows beyond cout); b2[h] f32
 * [round_up(cout,16)]; y[h]: f32 NHWC [B,H,W,y_stride[h]] (round_up(cout,4) channels written). */
typedef struct ctdet_head_desc {
  int32_t nheads, B, H, W, Cin, in_stride;
  const void* w2[4];
  const void* b2[4];
  void* y[4];
  int32_t y_stride[4], cout[4], act[4];
  float clamp_lo, clamp_hi;
} ctdet_head_desc;
int32_t ctdet_head_fused_fwd(const ctdet_head_desc* d, const void* x, const void* w1, const float* b1, void* stream);

/* DLA base layers fused for inference: (x/255 - mean)/std (centernet.py:193-200) -> base_layer 7x7 3->16 -> level0 3x3
 * 16->16 -> level1 3x3 stride 2 16->32, each + folded BatchNorm + ReLU (dla.py:204-215, called at dla.py:230-233 for
 * levels 0-1).  images: [B,3,H,W] planar uint8 or f32 (img_dtype CTDET_U8 / CTDET_F32), zero-padded bottom/right to Hp x
 * Wp after normalisation (ImageList.from_tensors); Hp % 16 == 0, Wp % 32 == 0.  Weights f16: w_stem [16][224] with
 * k = (r*8 + s)*4 + c (tap column s = 7 and channel c = 3 zero), w_l0 [16][160] and w_l1 [32][160] with k = (r*3 + s)*16 + c
 * (ctdet_pack_weights korder 0).  scale/bias: the folded BatchNorm of each layer, f32.  out: f16 NHWC
 * [B,Hp/2,Wp/2,out_stride] (32 channels written).  pooled (may be NULL): f16 NHWC [B,Hp/4,Wp/4,pool_stride], the
 * MaxPool2d(2) of `out` that level2's Tree starts with (dla.py:128-129, 139).  The intermediate maps are rounded to f16
 * exactly where the layer-by-layer path rounds them. */
typedef struct ctdet_dla_base_desc {
  int32_t B, H, W, Hp, Wp, img_dtype;
  int64_t img_batch_stride;   /* elements between images */
  float mean[3], std[3];
  int32_t out_stride;
  int32_t pool_stride;        /* pixel stride of `pooled` (elements); ignored when pooled is NULL */
} ctdet_dla_base_desc;
int32_t ctdet_dla_base_fwd(const ctdet_dla_base_desc* d, const void* images, const void* w_stem, const float* scale_stem,
                           const float* bias_stem, const void* w_l0, const float* scale_l0, const float* bias_l0,
                           const void* w_l1, const float* scale_l1, const float* bias_l1, void* out, void* pooled,
                           void* stream);
/* The same three layers (dla.py:204-215 after centernet.py:193-200) in f16x3 arithmetic: f32 tensors, every product as
 * hi*hi + lo*hi + hi*lo on the f16 matrix pipe with f32 accumulation -- what ctdet_conv2d computes for them one by one in
 * CTDET_F16X3 mode.  w_*: the layout-0 images of ctdet_pack_weights_x3 (stem: [16][224], k = (r*8 + s)*4 + c with tap column
 * s = 7 and channel c = 3 zero; level0 [16][144],
 * level1 [32][144], 16-byte groups {w_hi[4], w_lo[4]}); scale_*: folded BatchNorm scale times the pack's row scale.
 * out f32 [B,Hp/2,Wp/2,out_stride >= 32], pooled (may be NULL) f32 [B,Hp/4,Wp/4,pool_stride >= 32]. */
int32_t ctdet_dla_base_x3_fwd(const ctdet_dla_base_desc* d, const void* images, const void* w_stem, const float* scale_stem,
                              const float* bias_stem, const void* w_l0, const float* scale_l0, const float* bias_l0,
                              const void* w_l1, const float* scale_l1, const float* bias_l1, float* out, float* pooled,
                              void* stream);

/* nn.MaxPool2d(2, stride=2) on NHWC (dla.py:128-129). */
int32_t ctdet_maxpool2x2(const void* x, void* y, int32_t dtype, int32_t B, int32_t H, int32_t W, int32_t C,
                         int32_t in_stride, int32_t out_stride, void* stream);

/* F.max_pool2d(x, 3, stride=2, padding=1) of BasicStem (detectron2/modeling/backbone/resnet.py:341-345), NHWC;
 * output [B, (H-1)/2+1, (W-1)/2+1, C]; padded taps do not take part. */
int32_t ctdet_maxpool3x3s2(const void* x, void* y, int32_t dtype, int32_t B, int32_t H, int32_t W, int32_t C,
                           int32_t in_stride, int32_t out_stride, void* stream);
/* nn.MaxPool2d(kernel_size=3, stride=2, ceil_mode=True) of the VoVNet stages (detectron2/modeling/backbone/vovnet.py:291-292),
 * NHWC; output [B, ceil((H-3)/2)+1, ceil((W-3)/2)+1, C] (a last window that would start outside the map is dropped). */
int32_t ctdet_maxpool3x3s2_ceil(const void* x, void* y, int32_t dtype, int32_t B, int32_t H, int32_t W, int32_t C,
                                int32_t in_stride, int32_t out_stride, void* stream);
/* eSE attention of VoVNet (vovnet.py:200-213): ctdet_global_avgpool: out f32 [B][C] = mean over the HW pixels of NHWC x;
 * ctdet_ese_scale: y = x * hsigmoid(s[b][c]) (+ identity), hsigmoid(v) = relu6(v + 3) / 6 (vovnet.py:186-197); s is the
 * raw output of the module's 1x1 `fc` conv on the pooled vector; identity (may be NULL) is the block input of the later
 * OSA blocks of a stage (vovnet.py:268-271). */
int32_t ctdet_global_avgpool(const void* x, int32_t dtype, int32_t B, int32_t HW, int32_t C, int32_t stride, float* out,
                             void* stream);
/* *flag (int32 on the device, set to 1 by the caller) becomes 0 if any value of the f32 map x[M][C] (pixel stride `stride`
 * elements) is inf or NaN: the per-step guard of the eval step on the size / offset maps (ABI 7). */
int32_t ctdet_finite_flag(const float* x, int64_t M, int32_t C, int32_t stride, int32_t* flag, void* stream);
int32_t ctdet_ese_scale(const void* x, int32_t x_stride, const float* s, const void* identity, int32_t identity_stride,
                        void* y, int32_t y_stride, int32_t dtype, int32_t B, int32_t HW, int32_t C, void* stream);

/* Packs an f32 OIHW conv weight [O,I,R,S] (nn.Conv2d.weight as the reference stores it) into the f16 [rows_pad][Kpad]
 * operand of ctdet_conv2d_fwd / ctdet_dcnv2_fwd, zero padding included.  korder as in ctdet_conv_desc.  transposed = 0:
 * rows = O, channels = I (chans_pad >= I).  transposed = 1: the input-gradient operand (rows = I, channels = O, taps
 * flipped), i.e. dX = conv(dY, packed) -- what autograd of nn.Conv2d computes.  transposed = 2 / 3 ([O,I,3,3] weights, korder 0):
 * the operand of DCNv2's d(columns) contraction, a 1x1 conv from dY's O channels to 9*I column channels in tap-major /
 * chunk-major row order (rows_pad >= 9*I, chans_pad >= O) -- see ctdet_dcn_col2im_coord's dcol_chunked. */
int32_t ctdet_pack_weights(const float* w, void* packed, int32_t O, int32_t I, int32_t R, int32_t S, int32_t chans_pad,
                           int32_t rows_pad, int32_t Kpad, int32_t korder, int32_t transposed, void* stream);
/* n packs in one launch: a DEVICE table of descriptors with the arguments of ctdet_pack_weights; blk0 = number of
 * 256-thread blocks of the descriptors before it (ceil(rows_pad*Kpad / 256) each), total_blocks their sum.  A training step
 * re-packs every weight after the optimizer update with this call instead of one launch per layer and direction. */
typedef struct ctdet_pack_desc {
  const float* w; void* packed;
  int32_t O, I, R, S, chans_pad, rows_pad, Kpad, korder, transposed, blk0;
} ctdet_pack_desc;
int32_t ctdet_pack_weights_batch(const ctdet_pack_desc* table_dev, int32_t n, int32_t total_blocks, void* stream);

/* Weight packing for CTDET_DT_F16X3 contractions in one launch per weight (round 4; replaces a chain of ~30 torch kernels per
 * conv: permute, pad, row maximum, scale, split, pair interleave): the f32 OIHW parameter -> the split operand the kernels
 * read.  Every packed row is scaled by the power of two that brings its largest magnitude into [1024, 2048) -- the lo halves
 * then stay f16 normals -- and the inverse is written to scale_out[row] for row < scale_n (1.0 for rows of padding): pass
 * it as the `scale` of ctdet_conv2d_fwd / ctdet_dcnv2_fwd (multiply it into a folded BatchNorm scale if there is one).
 * layout 0: tap-major split image, rows of Kpad f32 units (Kpad % 4 == 0, >= R*S*chans_pad), each group of 4 k = {hi[4],
 *           lo[4]} f16 -- ctdet_conv_desc.korder 0: 1x1 / strided / 7x7 convs, DCNv2, the d(columns) operand;
 * layout 3: tap-pair image of the 3x3 halo kernel, korder 3 (chans_pad % 32 == 0, Kpad = chans_pad/32*288, rows_pad % 32 == 0);
 * layout 2: the same for an odd number of 16-channel chunks, korder 2 (Kpad = chans_pad/16*160);
 * layout 5: tap-major rows, every group of 8 k = {hi[8], lo[8]} (Kpad % 8 == 0): the operand ctdet_dcn_col2im_fused reads.
 * transposed: as in ctdet_pack_weights (1: the input-gradient operand, rows = input channels, taps flipped; 2 / 3: DCNv2's
 * d(columns) operand -- give R = S = 1 for those, the weight itself is [O, I, 3, 3]). */
int32_t ctdet_pack_weights_x3(const float* w, void* packed, float* scale_out, int32_t O, int32_t I, int32_t R, int32_t S,
                              int32_t chans_pad, int32_t rows_pad, int32_t Kpad, int32_t layout, int32_t transposed,
                              int32_t scale_n, void* stream);
/* n such packs in one launch from a DEVICE table; one 256-thread block per packed row: blk0 = sum of rows_pad of the
 * descriptors before it, total_blocks = the sum over all. */
typedef struct ctdet_pack3_desc {
  const float* w; void* packed; float* scale_out;
  int32_t O, I, R, S, chans_pad, rows_pad, Kpad, layout, transposed, scale_n, blk0, pad_;
} ctdet_pack3_desc;
int32_t ctdet_pack_weights_x3_batch(const ctdet_pack3_desc* table_dev, int32_t n, int32_t total_blocks, void* stream);

/* y = ConvTranspose2d(C, C, 2f, stride=f, padding=f/2, groups=C, bias=False)(x) + skip  (dla.py:162-177).
 * w is f32 [2f][2f][C] (the ConvTranspose2d weight [C,1,2f,2f] with the channel dim moved last); skip may be NULL. */
int32_t ctdet_dwconvT_add(const void* x, const float* w, const void* skip, void* y, int32_t dtype, int32_t B,
                          int32_t H, int32_t W, int32_t C, int32_t f, int32_t in_stride, int32_t skip_stride,
                          int32_t out_stride, void* stream);

/* Batched ctdet_decode (centernet.py:399-458): heat f32 NHWC [B,H,W,heat_stride] (already sigmoid+clamp; the first C
 * channels are classes, any C >= 1), wh/reg f32 with pixel strides; outputs boxes [B,K,4] f32, scores [B,K] f32, classes
 * [B,K] i32, inds [B,K] i32 (spatial index y*W+x; may be NULL).  Order: score desc, ties by c*H*W+y*W+x asc.  reg may
 * be NULL.  The heat map is read once; workspace: ctdet_decode_workspace_bytes of the same B, H, W, C, K.
 * heat_floor: a lower bound the caller promises for every positive heat value -- 1e-4f for the map `_sigmoid` clamps
 * (centernet.py:13-15) -- or 0 for none.  Results are the same either way; with the bound, the plateau a trained
 * network's background forms exactly on the clamp is skipped by the selection passes and only consulted (lowest flat index
 * first, which is the tie rule) when an image has fewer than K peaks above it.  A positive value below a non-zero
 * heat_floor is reported by ctdet_decode_status. */
size_t ctdet_decode_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t C, int32_t K);
int32_t ctdet_decode(const float* heat, int32_t heat_stride, const float* wh, int32_t wh_stride, const float* reg,
                     int32_t reg_stride, int32_t B, int32_t H, int32_t W, int32_t C, int32_t K, float down_ratio,
                     float heat_floor, void* workspace, float* boxes, float* scores, int32_t* classes, int32_t* inds,
                     void* stream);
/* CenterNet.inference_single_image (centernet.py:251-261) + detector_postprocess
 * (detectron2/modeling/postprocessing.py:11-72, structures/boxes.py:184-213,271-278) for a whole batch:
 * keep k < max_det with score > score_thresh, scale boxes by (scale_x, scale_y), clip to (out_w, out_h), drop
 * empty boxes, compact in order.  img_params f32 [B,4] = {scale_x, scale_y, out_w, out_h}; counts i32 [B]. */
int32_t ctdet_postprocess(const float* boxes, const float* scores, const int32_t* classes, int32_t B, int32_t K,
                          int32_t max_det, float score_thresh, const float* img_params, float* out_boxes,
                          float* out_scores, int32_t* out_classes, int32_t* counts, void* stream);
/* reads back the per-image status words of the last decode on this workspace (device->host copy + sync):
 * returns 0 if every image decoded exactly, -75 (EOVERFLOW) if a candidate buffer overflowed (cannot happen for a
 * workspace of the queried size), -22 (EINVAL) if a positive heat value lay below the promised heat_floor.
 * Test/diagnostic helper, not on the hot path. */
int32_t ctdet_decode_status(const void* workspace, int32_t B, int32_t H, int32_t W, int32_t C, int32_t K, void* stream);

/* gen_heatmap + gaussian_radius + draw_umich_gaussian, batched on device
 * (detectron2/data/detection_utils.py:600-705).  boxes f32 [B,Nmax,4] XYXY input pixels, classes i64
 * [B,Nmax], counts i32 [B].  hm f32 NHWC [B,H,W,C] is zeroed here.  wh/reg f32 [B,128,2], ind i64 [B,128],
 * reg_mask u8 [B,128]. */
int32_t ctdet_gaussian_targets(const float* boxes, const int64_t* classes, const int32_t* counts, int32_t B,
                               int32_t Nmax, int32_t H, int32_t W, int32_t C, float* hm, float* wh, float* reg,
                               int64_t* ind, uint8_t* reg_mask, void* stream);
/* gaussian_radius((h, w)) for a grid of integer sizes (test hook for detection_utils.py:654-680):
 * out_radius f64 [n], out_int i32 [n] = max(0, (int)radius). */
int32_t ctdet_gaussian_radius(const int32_t* hw_pairs, int32_t n, double* out_radius, int32_t* out_int, void* stream);

/* FocalLoss/_neg_loss forward + gradient wrt the logits, fused (centernet.py:204,323-369).
 * logits, gt: f32 NHWC [B,H,W,C]; alpha f32 [C].  partial: f32 workspace [3*nblocks] from
 * ctdet_focal_loss_workspace_bytes.  Outputs: loss f32[1]; stats f32[4] = {pos_loss, neg_loss, num_pos, 1/num_pos (1 if 0)};
 * grad f32 NHWC (d loss / d logits, times grad_scale), may be NULL for forward only. */
size_t ctdet_focal_loss_workspace_bytes(int64_t numel);
int32_t ctdet_focal_loss(const float* logits, const float* gt, const float* alpha, int32_t B, int32_t H, int32_t W,
                         int32_t C, float grad_scale, void* workspace, float* loss, float* stats, float* grad,
                         void* stream);

/* RegL1Loss forward + gradient (centernet.py:372-397): pred f32 NHWC [B,H,W,pred_stride] (2 channels used
 * starting at pred), mask u8 [B,N], ind i64 [B,N], target f32 [B,N,2]; loss f32[1];
 * grad (same layout/stride as pred, 2 channels, must be pre-zeroed by the caller) may be NULL. */
int32_t ctdet_reg_l1_loss(const float* pred, int32_t pred_stride, const uint8_t* mask, const int64_t* ind,
                          const float* target, int32_t B, int32_t N, int32_t HW, float grad_scale, float* loss,
                          float* grad, int32_t grad_stride, void* stream);

/* torch.optim.SGD step with momentum/weight decay (detectron2/solver/build.py:93-137):
 * g' = g + wd*p; buf = mom*buf + g'; p -= lr*buf   (first step: buf = g'). lr is read from device memory
 * so the captured graph stays valid across the WarmupMultiStepLR schedule. */
int32_t ctdet_sgd_momentum(float* param, const float* grad, float* momentum_buf, int64_t n, const float* lr_dev,
                           float momentum, float weight_decay, int32_t first_step, void* stream);

/* The same step over a flat buffer holding several parameter groups back to back: run r covers elements
 * [run_end[r-1], run_end[r]) (device i64 array, ascending, run_end[nruns-1] == n) with weight decay run_weight_decay[r] and
 * learning rate lr_table[run_lr_index[r]] (all device memory: the captured training graph stays valid across the schedule). */
int32_t ctdet_sgd_momentum_runs(float* param, const float* grad, float* momentum_buf, int64_t n, const int64_t* run_end,
                                const int32_t* run_lr_index, const float* run_weight_decay, const float* lr_table,
                                int32_t nruns, float momentum, int32_t first_step, void* stream);

/* ---- training-side entry points (f32 statistics and weight gradients; activations and activation gradients f16 --
 * the throughput mode -- or f32 -- the reference's precision -- selected by `dtype` (ctdet_dtype) / the descriptor's
 * compute_dtype).  Input gradients of plain convs are ctdet_conv2d_fwd calls with transposed/flipped weights (in_dil
 * for strided layers). */

/* nn.BatchNorm2d in training mode + optional residual add + ReLU (dla.py:59-73,86-94; deform_conv.py:501-519):
 * z = act(gamma*(y-mean)*invstd + beta + res); batch statistics over the M rows; running stats updated with
 * `momentum` (unbiased variance), saved mean/invstd/scale(=gamma*invstd)/shift for the backward. */
size_t ctdet_chan_workspace_bytes(int32_t C);
int32_t ctdet_bn_train_fwd(const void* y, int32_t y_stride, const void* res, int32_t res_stride, void* z,
                           int32_t z_stride, int32_t M, int32_t C, const float* gamma, const float* beta, float eps,
                           float momentum, float* running_mean, float* running_var, float* save_mean,
                           float* save_invstd, float* scale, float* shift, void* workspace, int32_t relu, int32_t dtype,
                           void* stream);
/* backward of the above: g = dz*(z>0) if relu; dgamma = sum g*xhat, dbeta = sum g,
 * dy = scale*(g - dbeta/M - xhat*dgamma/M); dres (optional) = g.  With y == NULL it is the backward of
 * "bias + activation" (dy = g, dbeta = bias gradient, dgamma untouched semantics: 0).  dgamma / dbeta are written (not
 * accumulated) multiplied by grad_mult -- the 1 / (loss scale * world size) every parameter gradient carries; dy uses
 * the plain sums. */
int32_t ctdet_bn_train_bwd(const void* dz, int32_t dz_stride, const void* z, int32_t z_stride, const void* y,
                           int32_t y_stride, const float* mean, const float* invstd, const float* scale, int32_t M,
                           int32_t C, int32_t relu, void* dy, int32_t dy_stride, void* dres, int32_t dres_stride,
                           float* dgamma, float* dbeta, float grad_mult, void* workspace, int32_t dtype, void* stream);
/* weight gradient of a conv: dw f32 [Cout][R*S*Cin] (tap-major k) += scale * sum over pixels; dw must be zeroed by
 * the caller.  Geometry from the descriptor (out_stride = pixel stride of dy; compute_dtype = dtype of x and dy). */
int32_t ctdet_conv_wgrad(const ctdet_conv_desc* d, const void* x, const void* dy, float* dw, float scale, void* stream);
/* The same sums accumulated in the PARAMETER's layout: with k = tap*cin_k + c (taps*cin_k = R*S*Cin of the descriptor),
 * dw[(n*cin_real + c)*taps + tap] += scale * sum, channels c >= cin_real and rows n >= cout_real (the channel padding of x
 * and dy) dropped -- dw can be the OIHW weight's slice of an optimizer's flat gradient buffer, so no permute / add kernels follow the weight gradient.
 * A 3x3 conv: taps = 9, cin_k = Cin; DCNv2's weight gradient as a 1x1 conv over the columns [M][9*Cin]: taps = 9, cin_k = Cin. */
int32_t ctdet_conv_wgrad_oihw(const ctdet_conv_desc* d, const void* x, const void* dy, float* dw, float scale, int32_t taps,
                              int32_t cin_k, int32_t cin_real, int32_t cout_real, void* stream);
/* n finished tap-major weight gradients (ctdet_conv_wgrad's layout, [cout][taps][cin_k] f32) added into their parameters'
 * OIHW gradients in one launch per 24 tensors: dst[i][(o*cin_real + c)*taps + t] += src[i][(o*taps + t)*cin_k + c].
 * The six arrays are HOST arrays of length n (device pointers in src / dst). */
int32_t ctdet_grad_scatter_oihw(const void* const* src, void* const* dst, const int32_t* cout, const int32_t* cin_real,
                                const int32_t* cin_k, const int32_t* taps, int32_t n, void* stream);
/* Interleave of the four output phases of a stride-2 3x3 conv's input gradient (f16 NHWC):
 * dst[b,y,x,c] = src[b,(y+1)/2,(x+1)/2,((y&1)*2+(x&1))*C + c]; src is the [B,Hs,Ws,>=4C] result of the 2x2 "phase" conv
 * over dY (ops_train.conv_dgrad), dst the [B,H,W,C] gradient (C % 8 == 0). */
int32_t ctdet_depth_to_space2(const void* src, int32_t src_stride, void* dst, int32_t dst_stride, int32_t B, int32_t H,
                              int32_t W, int32_t C, int32_t Hs, int32_t Ws, int32_t dtype, void* stream);
/* Training pieces of the VoVNet backbone (ABI 7).  ctdet_maxpool3x3s2_bwd: autograd of F.max_pool2d(x, 3, 2, 1)
 * (ceil_nopad = 0; resnet.py:341-345) / nn.MaxPool2d(3, 2, ceil_mode=True) (ceil_nopad = 1; vovnet.py:291-292): the gradient of a
 * window goes to its first maximum in scan order (PyTorch's rule), gathered per input element (no atomics).
 * eSE attention (vovnet.py:200-213), y = x * hsigmoid(s[b][c]) (+ identity): ctdet_ese_dot: out f32 [B][C] = sum over the pixels
 * of dy * x (the gradient reaching hsigmoid(s) before its slope); ctdet_ese_bwd: dx = dy * gate[b][c] + pooled_grad[b][c]
 * (gate = hsigmoid(s); pooled_grad = d(mean) / HW, the gradient that comes back through the fc layer and the average pool). */
int32_t ctdet_maxpool3x3s2_bwd(const void* x, int32_t x_stride, const void* dz, int32_t dz_stride, void* dx, int32_t dx_stride,
                               int32_t dtype, int32_t B, int32_t H, int32_t W, int32_t C, int32_t ceil_nopad, void* stream);
int32_t ctdet_ese_dot(const void* dy, int32_t dy_stride, const void* x, int32_t x_stride, int32_t dtype, int32_t B, int32_t HW,
                      int32_t C, float* out, void* stream);
int32_t ctdet_ese_bwd(const void* dy, int32_t dy_stride, const float* gate, const float* pooled_grad, void* dx, int32_t dx_stride,
                      int32_t dtype, int32_t B, int32_t HW, int32_t C, void* stream);
int32_t ctdet_maxpool2x2_bwd(const void* x, int32_t x_stride, const void* dz, int32_t dz_stride, void* dx,
                             int32_t dx_stride, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream);
/* depthwise ConvTranspose2d backward: dx f16, dw f32 [2f][2f][C] (+=, zeroed by the caller); w as in the forward */
int32_t ctdet_dwconvT_bwd(const void* x, int32_t x_stride, const void* dz, int32_t dz_stride, const float* w, void* dx,
                          int32_t dx_stride, float* dw, int32_t B, int32_t H, int32_t W, int32_t C, int32_t f,
                          int32_t dtype, void* stream);
/* DCNv2 3x3/s1/p1 training pieces: columns [M][9*Cin] f16 (= modulated_deformable_im2col, kernel.cu:786-868) and the
 * backward through the sampler (col2im :871-949 + coordinate/mask gradients :952-1066): dcol [M][9*Cin] f16 ->
 * dx f32 dense [B*H*W][Cin] (+= atomics, zeroed by the caller), dom [M][dom_stride] f32 or (f16 data only) f16: offset and
 * mask-logit gradients in channels 0..26, zeros in the padding channels 27..dom_stride-1 -- every element is written, the
 * caller does not clear it (a 32-channel f16 dom is directly the dY of the offset conv's backward).
 * dcol_chunked (f16, Cin % 32 == 0): a pixel's dcol row is [Cin/32][9][32] instead of [9][Cin] -- the order the scatter
 * kernel consumes it in (one contiguous 576-byte run per 32-channel chunk); the producer gets it by permuting the rows of the
 * weight matrix of the d(columns) contraction.
 * mask_is_prob: channels 18..26 of om are sigmoid-ed masks (the reference's functional API) and dom carries d/d(mask). */
int32_t ctdet_dcn_cols(const void* x, int32_t x_stride, const float* om, int32_t om_stride, void* col, int32_t B,
                       int32_t H, int32_t W, int32_t Cin, int32_t mask_is_prob, int32_t dtype, void* stream);
int32_t ctdet_dcn_col2im_coord(const void* dcol, const void* x, int32_t x_stride, const float* om, int32_t om_stride,
                               float* dx, void* dom, int32_t dom_stride, int32_t dom_dtype, int32_t B, int32_t H, int32_t W,
                               int32_t Cin, int32_t mask_is_prob, int32_t dcol_chunked, int32_t dtype, void* stream);

/* The f16x3 training mode's form of the two steps above in one kernel: d(columns) = dY . W (deform_conv_cuda.cu:1003-1009) is
 * computed per tile on the matrix pipe (f16x3 products) inside the scatter kernel instead of being written to and read from
 * memory (9*Cin floats per pixel: 604 MB for a 64-channel 128x128 layer at batch 16).  dy f32 [M][dy_stride], its first K
 * channels used (K % 32 == 0; channels beyond the layer's couts must be zero); w_packed / w_scale: ctdet_pack_weights_x3 of the
 * [Cout, Cin, 3, 3] weight with layout 5, transposed 3, chans_pad = Kpad = K, rows_pad >= 9*Cin.  Other arguments as
 * ctdet_dcn_col2im_coord (f32 x, f32 dom).  Returns 0 if launched, 1 if the shape does not qualify (map not divisible by 8x16,
 * Cin % 32 != 0, ...): nothing was done and the caller takes the two-step path. */
int32_t ctdet_dcn_col2im_fused(const float* dy, int32_t dy_stride, int32_t K, const void* w_packed, const float* w_scale,
                               const float* x, int32_t x_stride, const float* om, int32_t om_stride, float* dx, float* dom,
                               int32_t dom_stride, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t mask_is_prob, void* stream);

/* ---- data-parallel exchange over RCCL / xGMI (one process per GPU) ---------------------------------------------
 * What DistributedDataParallel's reducer does over NCCL in the reference (detectron2/engine/defaults.py:279-285,
 * launched per process by engine/launch.py:24-94).  Rank 0 makes a unique id and hands its CTDET_COMM_ID_BYTES to the
 * other ranks by any out-of-band channel; every rank then calls ctdet_comm_init with the GPU it owns current.
 * ctdet_allreduce_bucket: in-place SUM all-reduce of `count` f32 of the flat gradient buffer on `stream` (gradients are
 * pre-divided by the world size on the producer side, so SUM is DDP's mean); ctdet_bcast: the initial parameter
 * broadcast.  librccl.so is opened on first use. */
#define CTDET_COMM_ID_BYTES 128
int32_t ctdet_comm_unique_id(void* id_out);
int32_t ctdet_comm_init(const void* id, int32_t rank, int32_t world, void** comm_out);
int32_t ctdet_allreduce_bucket(void* comm, float* buf, int64_t count, void* stream);
int32_t ctdet_bcast(void* comm, float* buf, int64_t count, int32_t root, void* stream);
int32_t ctdet_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif
