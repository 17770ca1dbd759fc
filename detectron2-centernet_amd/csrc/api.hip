// C-ABI entry points of libctdet_hip.so (see include/ctdet_hip.h for the contract and the reference
// interfaces each function replaces).  No torch types, no allocation, no synchronisation (except the
// explicit diagnostic helper ctdet_decode_status).
#include "common.h"
#include "../../include/ctdet_hip.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <atomic>

static thread_local char g_err[512] = "";
void ctdet_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// launchers from the kernel files
int launch_preprocess(const void*, int, void*, int, int, int, int, int, int, long, const float*, const float*, int, int,
                      hipStream_t);
int launch_maxpool2x2(const void*, void*, int, int, int, int, int, int, int, hipStream_t);
int launch_maxpool3x3s2(const void*, void*, int, int, int, int, int, int, int, int, hipStream_t);
int launch_global_avgpool(const void*, int, int, int, int, int, float*, hipStream_t);
int launch_ese_scale(const void*, int, const float*, const void*, int, void*, int, int, int, int, int, hipStream_t);
int launch_finite_flag(const float*, long, int, int, int*, hipStream_t);
int launch_pack_weights(const float*, void*, int, int, int, int, int, int, int, int, int, hipStream_t);
int launch_dwconvT_add(const void*, const float*, const void*, void*, int, int, int, int, int, int, int, int, int,
                       hipStream_t);
size_t decode_workspace_bytes(int B, int H, int W, int C, int K);
int decode_status_words(int H, int W, int C, int K, long* ws_words, int* below_word);
int launch_decode(const DecArgs&, hipStream_t);
int launch_postprocess(const float*, const float*, const int*, int, int, int, float, const float*, float*, float*, int*,
                       int*, hipStream_t);
int launch_gaussian_radius(const int*, int, double*, int*, hipStream_t);
int launch_gaussian_targets(const float*, const int64_t*, const int*, int, int, int, int, int, float*, float*, float*,
                            int64_t*, uint8_t*, hipStream_t);
size_t focal_workspace_bytes(long numel);
int launch_focal_loss(const float*, const float*, const float*, int, int, int, int, float, void*, float*, float*, float*,
                      hipStream_t);
int launch_reg_l1(const float*, int, const uint8_t*, const int64_t*, const float*, int, int, int, float, float*, float*,
                  int, hipStream_t);
int launch_sgd(float*, const float*, float*, long, const float*, float, float, int, hipStream_t);
int launch_sgd_runs(float*, const float*, float*, long, const long*, const int*, const float*, const float*, int, float, int,
                    hipStream_t);

struct WgradArgs {
  const void* x; const void* dy; float* dw;
  int B, H, W, Cin, in_stride, Cout, Ho, Wo, dy_stride, R, S, stride, pad, dil, K, M, msplit;
  float scale;
  int lw, lh;
  int perm_rs, perm_cin, cin_real, cout_real;
};
int launch_grad_scatter_oihw(const void* const*, void* const*, const int*, const int*, const int*, const int*, int, hipStream_t);
int launch_pack_weights_batch(const void*, int, int, hipStream_t);
int launch_pack_weights_x3(const float*, void*, float*, int, int, int, int, int, int, int, int, int, int, hipStream_t);
int launch_pack_weights_x3_batch(const void*, int, int, hipStream_t);
int launch_split_weights(const float*, void*, long, hipStream_t);
bool dcn_offset_fused_ok(const ConvArgs& a);
bool dcn_split_window_ok(const ConvArgs& a);
size_t chan_reduce_workspace_bytes(int C);
int launch_bn_train_fwd(const f16*, int, const f16*, int, f16*, int, int, int, const float*, const float*, float, float,
                        float*, float*, float*, float*, float*, float*, void*, int, hipStream_t);
int launch_bn_train_bwd(const f16*, int, const f16*, int, const f16*, int, const float*, const float*, const float*, int,
                        int, int, f16*, int, f16*, int, float*, float*, float, void*, hipStream_t);
int launch_conv_wgrad(const WgradArgs&, hipStream_t);
int launch_conv_wgrad_f32(const WgradArgs&, hipStream_t);
int launch_conv_wgrad_x3(const WgradArgs&, hipStream_t);
int launch_bn_train_fwd_f32(const float*, int, const float*, int, float*, int, int, int, const float*, const float*, float,
                            float, float*, float*, float*, float*, float*, float*, void*, int, hipStream_t);
int launch_bn_train_bwd_f32(const float*, int, const float*, int, const float*, int, const float*, const float*, const float*,
                            int, int, int, float*, int, float*, int, float*, float*, float, void*, hipStream_t);
int launch_maxpool2x2_bwd_f32(const float*, int, const float*, int, float*, int, int, int, int, int, hipStream_t);
int launch_maxpool3x3s2_bwd(const void*, int, const void*, int, void*, int, int, int, int, int, int, int, int, int, hipStream_t);
int launch_ese_dot(const void*, int, const void*, int, int, int, int, int, float*, hipStream_t);
int launch_ese_bwd(const void*, int, const float*, const float*, void*, int, int, int, int, int, hipStream_t);
int launch_dwconvT_bwd_f32(const float*, int, const float*, int, const float*, float*, int, float*, int, int, int, int, int,
                           hipStream_t);
int launch_dcn_cols_f32(const float*, int, const float*, int, float*, int, int, int, int, int, hipStream_t);
int launch_dcn_col2im_coord_f32(const float*, const float*, int, const float*, int, float*, float*, int, int, int, int, int, int,
                                int, int, hipStream_t);
int launch_dcn_col2im_fused(const float*, int, int, const void*, const float*, const float*, int, const float*, int, float*, float*, int,
                            int, int, int, int, int, hipStream_t);
int launch_maxpool2x2_bwd(const f16*, int, const f16*, int, f16*, int, int, int, int, int, hipStream_t);
int launch_depth_to_space2(const void*, int, void*, int, int, int, int, int, int, int, int, hipStream_t);
int launch_dwconvT_bwd(const f16*, int, const f16*, int, const float*, f16*, int, float*, int, int, int, int, int,
                       hipStream_t);
int launch_dcn_cols(const f16*, int, const float*, int, f16*, int, int, int, int, int, hipStream_t);
int launch_dcn_col2im_coord(const f16*, const f16*, int, const float*, int, float*, void*, int, int, int, int, int, int, int, int,
                            hipStream_t);

static int fill_args(const ctdet_conv_desc* d, ConvArgs& a) {
  CTDET_CHECK(d != nullptr, "conv: null descriptor");
  CTDET_CHECK(d->B >= 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "conv: bad shape B=%d H=%d W=%d Cin=%d Cout=%d",
              d->B, d->H, d->W, d->Cin, d->Cout);
  CTDET_CHECK(d->R > 0 && d->S > 0 && d->stride > 0 && d->dil > 0 && d->pad >= 0, "conv: bad kernel geometry");
  const int idl = d->in_dil > 1 ? d->in_dil : 1;
  const int ho = ((d->H - 1) * idl + 1 + 2 * d->pad - (d->dil * (d->R - 1) + 1)) / d->stride + 1;
  const int wo = ((d->W - 1) * idl + 1 + 2 * d->pad - (d->dil * (d->S - 1) + 1)) / d->stride + 1;
  CTDET_CHECK(d->Ho >= ho && d->Ho < ho + idl && d->Wo >= wo && d->Wo < wo + idl,
              "conv: output size %dx%d does not match geometry (expected %dx%d)", d->Ho, d->Wo, ho, wo);
  CTDET_CHECK(d->in_stride >= d->Cin && d->out_stride >= d->Cout, "conv: pixel strides smaller than channel counts");
  memset(&a, 0, sizeof(a));
  a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.in_stride = d->in_stride;
  a.Cout = d->Cout; a.Ho = d->Ho; a.Wo = d->Wo; a.out_stride = d->out_stride; a.res_stride = d->res_stride;
  a.R = d->R; a.S = d->S; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
  a.K = d->R * d->S * d->Cin; a.Kpad = d->Kpad; a.Cout_pad = d->Cout_pad;
  a.M = d->B * d->Ho * d->Wo;
  a.act = d->act; a.clamp_lo = d->clamp_lo; a.clamp_hi = d->clamp_hi; a.korder = d->korder; a.in_dil = idl;
  CTDET_CHECK(d->korder == 0 || (d->korder == 1 && d->Cin % 32 == 0 && d->compute_dtype == CTDET_DT_F16) ||
                  (d->korder == 2 && d->Cin % 16 == 0 && d->compute_dtype == CTDET_DT_F16X3 && d->R == 3 && d->S == 3) ||
                  (d->korder == 3 && d->Cin % 32 == 0 && d->compute_dtype == CTDET_DT_F16X3 && d->R == 3 && d->S == 3),
              "conv: korder=%d invalid for Cin=%d", d->korder, d->Cin);
  CTDET_CHECK((long)d->B * d->Ho * d->Wo < (1L << 31), "conv: too many output pixels");
  return 0;
}

static std::atomic<unsigned> g_tuning{0};
unsigned ctdet_tuning_flags() { return g_tuning.load(std::memory_order_relaxed); }

int ctdet_device_cu_count() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = cache[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    hipDeviceProp_t prop;
    n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    cache[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

extern "C" {

const char* ctdet_last_error(void) { return g_err; }
int32_t ctdet_set_tuning_flags(uint32_t flags) { g_tuning.store(flags, std::memory_order_relaxed); return 0; }
uint32_t ctdet_get_tuning_flags(void) { return g_tuning.load(std::memory_order_relaxed); }
int32_t ctdet_abi_version(void) { return 7; }
int32_t ctdet_conv_cout_tile(int32_t cout) {
  if (cout <= 16) return 16;
  if (cout <= 32) return 32;
  if (cout <= 64) return 64;
  if (cout <= 128) return 128;          // = pick_bc (conv_common.h)
  if (cout % 128 == 0) return 128;
  if (cout % 64 == 0) return 64;
  return 32;
}

int32_t ctdet_conv2d_fwd(const ctdet_conv_desc* d, const void* x, const void* w_packed, const float* scale,
                         const float* bias, const void* residual, void* y, void* stream) {
  ConvArgs a;
  int rc = fill_args(d, a);
  if (rc) return rc;
  if (a.M == 0) return 0;
  CTDET_CHECK(x && w_packed && y, "conv: null pointer");
  a.x = x; a.w = w_packed; a.scale = scale; a.bias = bias; a.res = residual; a.y = y;
  if (d->compute_dtype == CTDET_DT_F16) return launch_conv_f16(a, d->out_dtype, false, (hipStream_t)stream);
  if (d->compute_dtype == CTDET_DT_F32 || d->compute_dtype == CTDET_DT_F16X3) {
    CTDET_CHECK(d->out_dtype == CTDET_DT_F32, "conv(f32 / f16x3): output must be f32");
    return launch_conv_f32(a, false, d->compute_dtype == CTDET_DT_F16X3, (hipStream_t)stream);
  }
  CTDET_CHECK(false, "conv: bad compute dtype %d", d->compute_dtype);
}

int32_t ctdet_conv1x1_cat_fwd(const ctdet_conv_desc* d, const void* const* xs, const int32_t* cins,
                              const int32_t* strides, int32_t nsrc, const void* w_packed, const float* scale,
                              const float* bias, const void* residual, void* y, void* stream) {
  ConvArgs a;
  int rc = fill_args(d, a);
  if (rc) return rc;
  if (a.M == 0) return 0;
  CTDET_CHECK(xs && cins && strides && w_packed && y, "conv1x1_cat: null pointer");
  CTDET_CHECK(nsrc >= 1 && nsrc <= 4, "conv1x1_cat: nsrc=%d must be 1..4", nsrc);
  CTDET_CHECK(d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0, "conv1x1_cat: only 1x1 stride-1 convs");
  a.korder = 0;
  const int align = d->compute_dtype == CTDET_DT_F16 ? 8 : (d->compute_dtype == CTDET_DT_F16X3 ? 4 : 1);
  int cum = 0;
  for (int j = 0; j < 4; ++j) {
    if (j < nsrc) {
      CTDET_CHECK(xs[j] && cins[j] > 0 && cins[j] % align == 0 && strides[j] >= cins[j] && strides[j] % align == 0,
                  "conv1x1_cat: bad source %d (cin=%d stride=%d)", j, cins[j], strides[j]);
      cum += cins[j];
      a.xs[j] = xs[j]; a.xs_stride[j] = strides[j];
    } else {
      a.xs[j] = xs[nsrc - 1]; a.xs_stride[j] = strides[nsrc - 1];
    }
    a.xs_cend[j] = cum;
  }
  CTDET_CHECK(cum == d->Cin, "conv1x1_cat: sources sum to %d channels, descriptor says %d", cum, d->Cin);
  a.nsrc = nsrc < 2 ? 2 : nsrc;  // always take the multi-source path (a single source is sources {0, 0-length})
  if (nsrc == 1) { a.nsrc = 1; a.in_stride = strides[0]; }
  a.x = xs[0]; a.w = w_packed; a.scale = scale; a.bias = bias; a.res = residual; a.y = y;
  if (d->compute_dtype == CTDET_DT_F16) return launch_conv_f16(a, d->out_dtype, false, (hipStream_t)stream);
  if (d->compute_dtype == CTDET_DT_F32 || d->compute_dtype == CTDET_DT_F16X3) {
    CTDET_CHECK(d->out_dtype == CTDET_DT_F32, "conv1x1_cat(f32 / f16x3): output must be f32");
    return launch_conv_f32(a, false, d->compute_dtype == CTDET_DT_F16X3, (hipStream_t)stream);
  }
  CTDET_CHECK(false, "conv1x1_cat: bad compute dtype %d", d->compute_dtype);
}

int32_t ctdet_dcnv2_fwd(const ctdet_conv_desc* d, const void* x, const float* offset_mask, int32_t om_stride,
                        int32_t mask_is_prob, const void* w_packed, const float* scale, const float* bias, void* y, void* stream) {
  ConvArgs a;
  int rc = fill_args(d, a);
  if (rc) return rc;
  if (a.M == 0) return 0;
  CTDET_CHECK(x && w_packed && y && offset_mask, "dcnv2: null pointer");
  CTDET_CHECK(om_stride >= 3 * d->R * d->S, "dcnv2: om_stride=%d < 3*R*S", om_stride);
  // korder 0: tap-major weights -> gather-from-global kernel; korder 1: chunk-major -> LDS-window kernel
  a.x = x; a.w = w_packed; a.scale = scale; a.bias = bias; a.res = nullptr; a.y = y;
  a.om = offset_mask; a.om_stride = om_stride; a.mask_is_prob = mask_is_prob;
  if (d->compute_dtype == CTDET_DT_F16) return launch_conv_f16(a, d->out_dtype, true, (hipStream_t)stream);
  if (d->compute_dtype == CTDET_DT_F32 || d->compute_dtype == CTDET_DT_F16X3) {
    CTDET_CHECK(d->out_dtype == CTDET_DT_F32, "dcnv2(f32 / f16x3): output must be f32");
    return launch_conv_f32(a, true, d->compute_dtype == CTDET_DT_F16X3, (hipStream_t)stream);
  }
  CTDET_CHECK(false, "dcnv2: bad compute dtype %d", d->compute_dtype);
}

int32_t ctdet_dcnv2_cols_supported(const ctdet_conv_desc* d, const void* x, const void* y) {
  ConvArgs a;
  if (fill_args(d, a) || d->compute_dtype != CTDET_DT_F16X3 || d->out_dtype != CTDET_DT_F32) return 0;
  a.x = x; a.y = const_cast<void*>(y);
  return dcn_split_window_ok(a) ? 1 : 0;
}

int32_t ctdet_dcnv2_fwd_cols(const ctdet_conv_desc* d, const void* x, const float* offset_mask, int32_t om_stride,
                             int32_t mask_is_prob, const void* w_packed, const float* scale, const float* bias, void* y,
                             float* cols_out, void* stream) {
  ConvArgs a;
  int rc = fill_args(d, a);
  if (rc) return rc;
  if (a.M == 0) return 0;
  CTDET_CHECK(x && w_packed && y && offset_mask && cols_out, "dcnv2_fwd_cols: null pointer");
  CTDET_CHECK(om_stride >= 3 * d->R * d->S, "dcnv2: om_stride=%d < 3*R*S", om_stride);
  CTDET_CHECK(d->compute_dtype == CTDET_DT_F16X3 && d->out_dtype == CTDET_DT_F32 && (((size_t)cols_out) & 15) == 0,
              "dcnv2_fwd_cols: the f16x3 mode's entry point (f32 tensors), 16-byte aligned columns");
  a.x = x; a.w = w_packed; a.scale = scale; a.bias = bias; a.res = nullptr; a.y = y;
  a.om = offset_mask; a.om_stride = om_stride; a.mask_is_prob = mask_is_prob;
  CTDET_CHECK(dcn_split_window_ok(a), "dcnv2_fwd_cols: shape not served by the LDS-window kernel (ask ctdet_dcnv2_cols_supported first)");
  a.cols_out = cols_out;
  return launch_conv_f32(a, true, true, (hipStream_t)stream);
}

int32_t ctdet_dcnv2_offset_supported(const ctdet_conv_desc* d) {
  ConvArgs a;
  if (fill_args(d, a) || d->compute_dtype != CTDET_DT_F16) return 0;
  a.y = nullptr;
  return dcn_offset_fused_ok(a) ? 1 : 0;
}

int32_t ctdet_dcnv2_offset_fwd(const ctdet_conv_desc* d, const void* x, const void* w_off_packed, const float* b_off,
                               float* om_out, int32_t om_out_stride, const void* w_packed, const float* scale,
                               const float* bias, void* y, void* stream) {
  ConvArgs a;
  int rc = fill_args(d, a);
  if (rc) return rc;
  if (a.M == 0) return 0;
  CTDET_CHECK(x && w_off_packed && b_off && w_packed && y, "dcnv2_offset: null pointer");
  CTDET_CHECK(d->compute_dtype == CTDET_DT_F16, "dcnv2_offset: f16 only");
  CTDET_CHECK(!om_out || (om_out_stride >= 28 && om_out_stride % 4 == 0 && ((size_t)om_out & 15) == 0),
              "dcnv2_offset: om_out needs a 16-byte aligned row of >= 28 floats (stride %d)", om_out_stride);
  a.x = x; a.w = w_packed; a.scale = scale; a.bias = bias; a.res = nullptr; a.y = y;
  a.w_off = w_off_packed; a.b_off = b_off; a.om_out = om_out; a.om_out_stride = om_out_stride;
  return launch_conv_f16(a, d->out_dtype, true, (hipStream_t)stream);
}

int32_t ctdet_preprocess(const void* img, int32_t img_dtype, void* out, int32_t out_dtype, int32_t B, int32_t H,
                         int32_t W, int32_t Hp, int32_t Wp, int64_t img_batch_stride, const float* mean3,
                         const float* std3, int32_t out_stride, int32_t border, void* stream) {
  CTDET_CHECK(img && out && mean3 && std3, "preprocess: null pointer");
  return launch_preprocess(img, img_dtype, out, out_dtype, B, H, W, Hp, Wp, (long)img_batch_stride, mean3, std3,
                           out_stride, border, (hipStream_t)stream);
}

int32_t ctdet_head_fused_fwd(const ctdet_head_desc* d, const void* x, const void* w1, const float* b1, void* stream) {
  CTDET_CHECK(d && x && w1 && b1, "head_fused: null pointer");
  HeadArgs a = {};
  a.x = x; a.w1 = w1; a.b1 = b1;
  a.nheads = d->nheads; a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.in_stride = d->in_stride;
  a.clamp_lo = d->clamp_lo; a.clamp_hi = d->clamp_hi;
  CTDET_CHECK(d->nheads >= 1 && d->nheads <= 4, "head_fused: nheads=%d out of range", d->nheads);
  for (int h = 0; h < d->nheads; ++h) {
    CTDET_CHECK(d->w2[h] && d->b2[h] && d->y[h], "head_fused: head %d: null pointer", h);
    a.w2[h] = d->w2[h]; a.b2[h] = (const float*)d->b2[h]; a.y[h] = (float*)d->y[h];
    a.y_stride[h] = d->y_stride[h]; a.cout[h] = d->cout[h]; a.act[h] = d->act[h];
  }
  return launch_head_fused(a, (hipStream_t)stream);
}

int32_t ctdet_dla_base_fwd(const ctdet_dla_base_desc* d, const void* images, const void* w_stem, const float* scale_stem,
                           const float* bias_stem, const void* w_l0, const float* scale_l0, const float* bias_l0,
                           const void* w_l1, const float* scale_l1, const float* bias_l1, void* out, void* pooled,
                           void* stream) {
  CTDET_CHECK(d && images && w_stem && scale_stem && bias_stem && w_l0 && scale_l0 && bias_l0 && w_l1 && scale_l1 && bias_l1 &&
              out, "dla_base: null pointer");
  BaseArgs a = {};
  a.img = images; a.img_dtype = d->img_dtype; a.img_batch_stride = (long)d->img_batch_stride;
  a.B = d->B; a.H = d->H; a.W = d->W; a.Hp = d->Hp; a.Wp = d->Wp;
  for (int i = 0; i < 3; ++i) { a.mean[i] = d->mean[i]; a.stdv[i] = d->std[i]; }
  a.w0 = w_stem; a.s0 = scale_stem; a.b0 = bias_stem;
  a.w1 = w_l0; a.s1 = scale_l0; a.b1 = bias_l0;
  a.w2 = w_l1; a.s2 = scale_l1; a.b2 = bias_l1;
  a.y = out; a.out_stride = d->out_stride;
  a.pool = pooled; a.pool_stride = d->pool_stride;
  return launch_dla_base(a, (hipStream_t)stream);
}

int32_t ctdet_dla_base_x3_fwd(const ctdet_dla_base_desc* d, const void* images, const void* w_stem, const float* scale_stem,
                              const float* bias_stem, const void* w_l0, const float* scale_l0, const float* bias_l0,
                              const void* w_l1, const float* scale_l1, const float* bias_l1, float* out, float* pooled,
                              void* stream) {
  CTDET_CHECK(d && images && w_stem && scale_stem && bias_stem && w_l0 && scale_l0 && bias_l0 && w_l1 && scale_l1 && bias_l1 &&
              out, "dla_base(f16x3): null pointer");
  BaseArgs a = {};
  a.img = images; a.img_dtype = d->img_dtype; a.img_batch_stride = (long)d->img_batch_stride;
  a.B = d->B; a.H = d->H; a.W = d->W; a.Hp = d->Hp; a.Wp = d->Wp;
  for (int i = 0; i < 3; ++i) { a.mean[i] = d->mean[i]; a.stdv[i] = d->std[i]; }
  a.w0 = w_stem; a.s0 = scale_stem; a.b0 = bias_stem;
  a.w1 = w_l0; a.s1 = scale_l0; a.b1 = bias_l0;
  a.w2 = w_l1; a.s2 = scale_l1; a.b2 = bias_l1;
  a.y = out; a.out_stride = d->out_stride;
  a.pool = pooled; a.pool_stride = d->pool_stride;
  return launch_dla_base_x3(a, (hipStream_t)stream);
}

int32_t ctdet_maxpool2x2(const void* x, void* y, int32_t dtype, int32_t B, int32_t H, int32_t W, int32_t C,
                         int32_t in_stride, int32_t out_stride, void* stream) {
  CTDET_CHECK(x && y, "maxpool2x2: null pointer");
  return launch_maxpool2x2(x, y, dtype, B, H, W, C, in_stride, out_stride, (hipStream_t)stream);
}

int32_t ctdet_maxpool3x3s2(const void* x, void* y, int32_t dtype, int32_t B, int32_t H, int32_t W, int32_t C,
                           int32_t in_stride, int32_t out_stride, void* stream) {
  CTDET_CHECK(x && y, "maxpool3x3s2: null pointer");
  return launch_maxpool3x3s2(x, y, dtype, B, H, W, C, in_stride, out_stride, 0, (hipStream_t)stream);
}

int32_t ctdet_maxpool3x3s2_ceil(const void* x, void* y, int32_t dtype, int32_t B, int32_t H, int32_t W, int32_t C,
                                int32_t in_stride, int32_t out_stride, void* stream) {
  CTDET_CHECK(x && y, "maxpool3x3s2_ceil: null pointer");
  return launch_maxpool3x3s2(x, y, dtype, B, H, W, C, in_stride, out_stride, 1, (hipStream_t)stream);
}

int32_t ctdet_global_avgpool(const void* x, int32_t dtype, int32_t B, int32_t HW, int32_t C, int32_t stride, float* out,
                             void* stream) {
  CTDET_CHECK(x && out, "global_avgpool: null pointer");
  return launch_global_avgpool(x, dtype, B, HW, C, stride, out, (hipStream_t)stream);
}

int32_t ctdet_finite_flag(const float* x, int64_t M, int32_t C, int32_t stride, int32_t* flag, void* stream) {
  CTDET_CHECK(x && flag, "finite_flag: null pointer");
  return launch_finite_flag(x, (long)M, C, stride, (int*)flag, (hipStream_t)stream);
}

int32_t ctdet_ese_scale(const void* x, int32_t x_stride, const float* s, const void* identity, int32_t identity_stride,
                        void* y, int32_t y_stride, int32_t dtype, int32_t B, int32_t HW, int32_t C, void* stream) {
  CTDET_CHECK(x && s && y, "ese_scale: null pointer");
  return launch_ese_scale(x, x_stride, s, identity, identity_stride, y, y_stride, dtype, B, HW, C, (hipStream_t)stream);
}

int32_t ctdet_pack_weights(const float* w, void* packed, int32_t O, int32_t I, int32_t R, int32_t S, int32_t chans_pad,
                           int32_t rows_pad, int32_t Kpad, int32_t korder, int32_t transposed, void* stream) {
  CTDET_CHECK(w && packed, "pack_weights: null pointer");
  return launch_pack_weights(w, packed, O, I, R, S, chans_pad, rows_pad, Kpad, korder, transposed, (hipStream_t)stream);
}

int32_t ctdet_split_weights(const float* w_packed_f32, void* w_split, int64_t n, void* stream) {
  CTDET_CHECK(w_packed_f32 && w_split && n >= 0, "split_weights: bad arguments");
  return launch_split_weights(w_packed_f32, w_split, (long)n, (hipStream_t)stream);
}

int32_t ctdet_pack_weights_batch(const ctdet_pack_desc* table_dev, int32_t n, int32_t total_blocks, void* stream) {
  CTDET_CHECK(n >= 0 && total_blocks >= 0 && (n == 0 || table_dev), "pack_weights_batch: bad arguments");
  static_assert(sizeof(ctdet_pack_desc) == 56, "ctdet_pack_desc layout");
  return launch_pack_weights_batch(table_dev, n, total_blocks, (hipStream_t)stream);
}

int32_t ctdet_pack_weights_x3(const float* w, void* packed, float* scale_out, int32_t O, int32_t I, int32_t R, int32_t S,
                              int32_t chans_pad, int32_t rows_pad, int32_t Kpad, int32_t layout, int32_t transposed,
                              int32_t scale_n, void* stream) {
  return launch_pack_weights_x3(w, packed, scale_out, O, I, R, S, chans_pad, rows_pad, Kpad, layout, transposed, scale_n,
                                (hipStream_t)stream);
}

int32_t ctdet_pack_weights_x3_batch(const ctdet_pack3_desc* table_dev, int32_t n, int32_t total_blocks, void* stream) {
  CTDET_CHECK(n >= 0 && total_blocks >= 0 && (n == 0 || table_dev), "pack_weights_x3_batch: bad arguments");
  static_assert(sizeof(ctdet_pack3_desc) == 72, "ctdet_pack3_desc layout");
  return launch_pack_weights_x3_batch(table_dev, n, total_blocks, (hipStream_t)stream);
}

int32_t ctdet_dwconvT_add(const void* x, const float* w, const void* skip, void* y, int32_t dtype, int32_t B,
                          int32_t H, int32_t W, int32_t C, int32_t f, int32_t in_stride, int32_t skip_stride,
                          int32_t out_stride, void* stream) {
  CTDET_CHECK(x && w && y, "dwconvT: null pointer");
  return launch_dwconvT_add(x, w, skip, y, dtype, B, H, W, C, f, in_stride, skip_stride, out_stride,
                            (hipStream_t)stream);
}

size_t ctdet_decode_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t C, int32_t K) {
  return decode_workspace_bytes(B, H, W, C, K);
}

int32_t ctdet_decode(const float* heat, int32_t heat_stride, const float* wh, int32_t wh_stride, const float* reg,
                     int32_t reg_stride, int32_t B, int32_t H, int32_t W, int32_t C, int32_t K, float down_ratio,
                     float heat_floor, void* workspace, float* boxes, float* scores, int32_t* classes, int32_t* inds,
                     void* stream) {
  CTDET_CHECK(heat && wh && workspace && boxes && scores && classes, "decode: null pointer");
  CTDET_CHECK(B >= 0 && H > 0 && W > 0, "decode: bad shape B=%d H=%d W=%d", B, H, W);
  CTDET_CHECK(heat_floor >= 0.f && heat_floor < 1.f, "decode: heat_floor %g outside [0, 1)", (double)heat_floor);
  DecArgs a;
  a.heat = heat; a.wh = wh; a.reg = reg; a.heat_stride = heat_stride; a.wh_stride = wh_stride; a.reg_stride = reg_stride;
  a.B = B; a.H = H; a.W = W; a.C = C; a.K = K; a.down_ratio = down_ratio;
  memcpy(&a.floor_bits, &heat_floor, 4);
  a.ws = (uint32_t*)workspace; a.boxes = boxes; a.scores = scores; a.classes = classes; a.inds = inds;
  return launch_decode(a, (hipStream_t)stream);
}

int32_t ctdet_postprocess(const float* boxes, const float* scores, const int32_t* classes, int32_t B, int32_t K,
                          int32_t max_det, float score_thresh, const float* img_params, float* out_boxes,
                          float* out_scores, int32_t* out_classes, int32_t* counts, void* stream) {
  CTDET_CHECK(boxes && scores && classes && img_params && out_boxes && out_scores && out_classes && counts,
              "postprocess: null pointer");
  return launch_postprocess(boxes, scores, classes, B, K, max_det, score_thresh, img_params, out_boxes, out_scores,
                            out_classes, counts, (hipStream_t)stream);
}

int32_t ctdet_decode_status(const void* workspace, int32_t B, int32_t H, int32_t W, int32_t C, int32_t K, void* stream) {
  long words = 0;
  int below = 0;
  const int flag = decode_status_words(H, W, C, K, &words, &below);
  for (int b = 0; b < B; ++b) {
    uint32_t st[16];
    hipError_t e = hipMemcpyAsync(st, (const char*)workspace + (size_t)words * 4 * b, sizeof(st), hipMemcpyDeviceToHost,
                                  (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    CTDET_CHECK(e == hipSuccess, "decode_status: copy failed: %s", hipGetErrorString(e));
    if (st[flag]) {
      ctdet_set_error("decode: image %d overflowed the candidate buffer", b);
      return -75;
    }
    if (st[below]) {
      ctdet_set_error("decode: image %d holds positive heat values below the promised heat_floor", b);
      return -22;
    }
  }
  return 0;
}

int32_t ctdet_gaussian_targets(const float* boxes, const int64_t* classes, const int32_t* counts, int32_t B,
                               int32_t Nmax, int32_t H, int32_t W, int32_t C, float* hm, float* wh, float* reg,
                               int64_t* ind, uint8_t* reg_mask, void* stream) {
  CTDET_CHECK(boxes && classes && counts && hm && wh && reg && ind && reg_mask, "gaussian_targets: null pointer");
  CTDET_CHECK(Nmax >= 1, "gaussian_targets: Nmax must be >= 1");
  return launch_gaussian_targets(boxes, classes, counts, B, Nmax, H, W, C, hm, wh, reg, ind, reg_mask,
                                 (hipStream_t)stream);
}

int32_t ctdet_gaussian_radius(const int32_t* hw_pairs, int32_t n, double* out_radius, int32_t* out_int, void* stream) {
  CTDET_CHECK(hw_pairs, "gaussian_radius: null pointer");
  return launch_gaussian_radius(hw_pairs, n, out_radius, out_int, (hipStream_t)stream);
}

size_t ctdet_focal_loss_workspace_bytes(int64_t numel) { return focal_workspace_bytes((long)numel); }

int32_t ctdet_focal_loss(const float* logits, const float* gt, const float* alpha, int32_t B, int32_t H, int32_t W,
                         int32_t C, float grad_scale, void* workspace, float* loss, float* stats, float* grad,
                         void* stream) {
  CTDET_CHECK(logits && gt && alpha && workspace && loss && stats, "focal_loss: null pointer");
  return launch_focal_loss(logits, gt, alpha, B, H, W, C, grad_scale, workspace, loss, stats, grad, (hipStream_t)stream);
}

int32_t ctdet_reg_l1_loss(const float* pred, int32_t pred_stride, const uint8_t* mask, const int64_t* ind,
                          const float* target, int32_t B, int32_t N, int32_t HW, float grad_scale, float* loss,
                          float* grad, int32_t grad_stride, void* stream) {
  CTDET_CHECK(pred && mask && ind && target && loss, "reg_l1_loss: null pointer");
  return launch_reg_l1(pred, pred_stride, mask, ind, target, B, N, HW, grad_scale, loss, grad, grad_stride,
                       (hipStream_t)stream);
}

size_t ctdet_chan_workspace_bytes(int32_t C) { return chan_reduce_workspace_bytes(C); }

int32_t ctdet_bn_train_fwd(const void* y, int32_t y_stride, const void* res, int32_t res_stride, void* z,
                           int32_t z_stride, int32_t M, int32_t C, const float* gamma, const float* beta, float eps,
                           float momentum, float* running_mean, float* running_var, float* save_mean,
                           float* save_invstd, float* scale, float* shift, void* workspace, int32_t relu, int32_t dtype,
                           void* stream) {
  CTDET_CHECK(y && z && gamma && beta && save_mean && save_invstd && scale && shift && workspace, "bn_train_fwd: null pointer");
  CTDET_CHECK(M > 0, "bn_train_fwd: empty batch");
  if (dtype == CTDET_DT_F32)
    return launch_bn_train_fwd_f32((const float*)y, y_stride, (const float*)res, res_stride, (float*)z, z_stride, M, C, gamma,
                                   beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, scale, shift,
                                   workspace, relu, (hipStream_t)stream);
  return launch_bn_train_fwd((const f16*)y, y_stride, (const f16*)res, res_stride, (f16*)z, z_stride, M, C, gamma, beta,
                             eps, momentum, running_mean, running_var, save_mean, save_invstd, scale, shift, workspace,
                             relu, (hipStream_t)stream);
}

int32_t ctdet_bn_train_bwd(const void* dz, int32_t dz_stride, const void* z, int32_t z_stride, const void* y,
                           int32_t y_stride, const float* mean, const float* invstd, const float* scale, int32_t M,
                           int32_t C, int32_t relu, void* dy, int32_t dy_stride, void* dres, int32_t dres_stride,
                           float* dgamma, float* dbeta, float grad_mult, void* workspace, int32_t dtype, void* stream) {
  CTDET_CHECK(dz && dy && dgamma && dbeta && workspace, "bn_train_bwd: null pointer");
  CTDET_CHECK(!relu || z, "bn_train_bwd: relu backward needs z");
  CTDET_CHECK(!y || (mean && invstd && scale), "bn_train_bwd: statistics missing");
  if (dtype == CTDET_DT_F32)
    return launch_bn_train_bwd_f32((const float*)dz, dz_stride, (const float*)z, z_stride, (const float*)y, y_stride, mean,
                                   invstd, scale, M, C, relu, (float*)dy, dy_stride, (float*)dres, dres_stride, dgamma,
                                   dbeta, grad_mult, workspace, (hipStream_t)stream);
  return launch_bn_train_bwd((const f16*)dz, dz_stride, (const f16*)z, z_stride, (const f16*)y, y_stride, mean, invstd,
                             scale, M, C, relu, (f16*)dy, dy_stride, (f16*)dres, dres_stride, dgamma, dbeta, grad_mult,
                             workspace, (hipStream_t)stream);
}

int32_t ctdet_conv_wgrad(const ctdet_conv_desc* d, const void* x, const void* dy, float* dw, float scale, void* stream) {
  return ctdet_conv_wgrad_oihw(d, x, dy, dw, scale, 0, 0, 0, 0, stream);
}

int32_t ctdet_conv_wgrad_oihw(const ctdet_conv_desc* d, const void* x, const void* dy, float* dw, float scale, int32_t taps,
                              int32_t cin_k, int32_t cin_real, int32_t cout_real, void* stream) {
  CTDET_CHECK(d && x && dy && dw, "conv_wgrad: null pointer");
  CTDET_CHECK(taps == 0 || (cin_k > 0 && cin_real > 0 && cin_real <= cin_k && taps * cin_k == d->R * d->S * d->Cin &&
                            cout_real > 0 && cout_real <= d->Cout),
              "conv_wgrad_oihw: taps=%d cin_k=%d cin_real=%d do not factor K=%d", taps, cin_k, cin_real, d->R * d->S * d->Cin);
  WgradArgs a;
  a.perm_rs = taps; a.perm_cin = cin_k; a.cin_real = cin_real; a.cout_real = cout_real;
  a.x = x; a.dy = dy; a.dw = dw;
  a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.in_stride = d->in_stride; a.Cout = d->Cout; a.Ho = d->Ho;
  a.Wo = d->Wo; a.dy_stride = d->out_stride; a.R = d->R; a.S = d->S; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
  a.K = d->R * d->S * d->Cin; a.M = d->B * d->Ho * d->Wo; a.msplit = 1; a.scale = scale;
  if (a.M == 0) return 0;
  if (d->compute_dtype == CTDET_DT_F32) return launch_conv_wgrad_f32(a, (hipStream_t)stream);
  if (d->compute_dtype == CTDET_DT_F16X3) return launch_conv_wgrad_x3(a, (hipStream_t)stream);
  CTDET_CHECK(d->compute_dtype == CTDET_DT_F16, "conv_wgrad: bad compute dtype %d", d->compute_dtype);
  return launch_conv_wgrad(a, (hipStream_t)stream);
}

int32_t ctdet_grad_scatter_oihw(const void* const* src, void* const* dst, const int32_t* cout, const int32_t* cin_real,
                                const int32_t* cin_k, const int32_t* taps, int32_t n, void* stream) {
  CTDET_CHECK(n >= 0 && (n == 0 || (src && dst && cout && cin_real && cin_k && taps)), "grad_scatter_oihw: null pointer");
  return launch_grad_scatter_oihw(src, dst, cout, cin_real, cin_k, taps, n, (hipStream_t)stream);
}

int32_t ctdet_depth_to_space2(const void* src, int32_t src_stride, void* dst, int32_t dst_stride, int32_t B, int32_t H,
                              int32_t W, int32_t C, int32_t Hs, int32_t Ws, int32_t dtype, void* stream) {
  CTDET_CHECK(src && dst, "depth_to_space2: null pointer");
  CTDET_CHECK(dtype == CTDET_DT_F16 || dtype == CTDET_DT_F32, "depth_to_space2: bad dtype %d", dtype);
  return launch_depth_to_space2(src, src_stride, dst, dst_stride, B, H, W, C, Hs, Ws, dtype, (hipStream_t)stream);
}

int32_t ctdet_maxpool3x3s2_bwd(const void* x, int32_t x_stride, const void* dz, int32_t dz_stride, void* dx, int32_t dx_stride,
                               int32_t dtype, int32_t B, int32_t H, int32_t W, int32_t C, int32_t ceil_nopad, void* stream) {
  CTDET_CHECK(x && dz && dx, "maxpool3x3s2_bwd: null pointer");
  int Ho, Wo, pad;
  if (ceil_nopad) {
    pad = 0;
    Ho = (H - 3 + 1) / 2 + 1; Wo = (W - 3 + 1) / 2 + 1;
    if ((Ho - 1) * 2 >= H) --Ho;
    if ((Wo - 1) * 2 >= W) --Wo;
  } else {
    pad = 1;
    Ho = (H - 1) / 2 + 1; Wo = (W - 1) / 2 + 1;
  }
  return launch_maxpool3x3s2_bwd(x, x_stride, dz, dz_stride, dx, dx_stride, dtype, B, H, W, C, pad, Ho, Wo, (hipStream_t)stream);
}

int32_t ctdet_ese_dot(const void* dy, int32_t dy_stride, const void* x, int32_t x_stride, int32_t dtype, int32_t B, int32_t HW,
                      int32_t C, float* out, void* stream) {
  CTDET_CHECK(dy && x && out, "ese_dot: null pointer");
  return launch_ese_dot(dy, dy_stride, x, x_stride, dtype, B, HW, C, out, (hipStream_t)stream);
}

int32_t ctdet_ese_bwd(const void* dy, int32_t dy_stride, const float* gate, const float* pooled_grad, void* dx, int32_t dx_stride,
                      int32_t dtype, int32_t B, int32_t HW, int32_t C, void* stream) {
  CTDET_CHECK(dy && gate && pooled_grad && dx, "ese_bwd: null pointer");
  return launch_ese_bwd(dy, dy_stride, gate, pooled_grad, dx, dx_stride, dtype, B, HW, C, (hipStream_t)stream);
}

int32_t ctdet_maxpool2x2_bwd(const void* x, int32_t x_stride, const void* dz, int32_t dz_stride, void* dx,
                             int32_t dx_stride, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream) {
  CTDET_CHECK(x && dz && dx, "maxpool2x2_bwd: null pointer");
  if (dtype == CTDET_DT_F32)
    return launch_maxpool2x2_bwd_f32((const float*)x, x_stride, (const float*)dz, dz_stride, (float*)dx, dx_stride, B, H, W, C,
                                     (hipStream_t)stream);
  return launch_maxpool2x2_bwd((const f16*)x, x_stride, (const f16*)dz, dz_stride, (f16*)dx, dx_stride, B, H, W, C,
                               (hipStream_t)stream);
}

int32_t ctdet_dwconvT_bwd(const void* x, int32_t x_stride, const void* dz, int32_t dz_stride, const float* w, void* dx,
                          int32_t dx_stride, float* dw, int32_t B, int32_t H, int32_t W, int32_t C, int32_t f,
                          int32_t dtype, void* stream) {
  CTDET_CHECK(x && dz && w && dx && dw, "dwconvT_bwd: null pointer");
  if (dtype == CTDET_DT_F32)
    return launch_dwconvT_bwd_f32((const float*)x, x_stride, (const float*)dz, dz_stride, w, (float*)dx, dx_stride, dw, B, H,
                                  W, C, f, (hipStream_t)stream);
  return launch_dwconvT_bwd((const f16*)x, x_stride, (const f16*)dz, dz_stride, w, (f16*)dx, dx_stride, dw, B, H, W, C, f,
                            (hipStream_t)stream);
}

int32_t ctdet_dcn_cols(const void* x, int32_t x_stride, const float* om, int32_t om_stride, void* col, int32_t B,
                       int32_t H, int32_t W, int32_t Cin, int32_t mask_is_prob, int32_t dtype, void* stream) {
  CTDET_CHECK(x && om && col, "dcn_cols: null pointer");
  if (dtype == CTDET_DT_F32)
    return launch_dcn_cols_f32((const float*)x, x_stride, om, om_stride, (float*)col, B, H, W, Cin, mask_is_prob,
                               (hipStream_t)stream);
  return launch_dcn_cols((const f16*)x, x_stride, om, om_stride, (f16*)col, B, H, W, Cin, mask_is_prob, (hipStream_t)stream);
}

int32_t ctdet_dcn_col2im_coord(const void* dcol, const void* x, int32_t x_stride, const float* om, int32_t om_stride,
                               float* dx, void* dom, int32_t dom_stride, int32_t dom_dtype, int32_t B, int32_t H, int32_t W,
                               int32_t Cin, int32_t mask_is_prob, int32_t dcol_chunked, int32_t dtype, void* stream) {
  CTDET_CHECK(dcol && x && om && dx && dom, "dcn_col2im_coord: null pointer");
  CTDET_CHECK(dom_dtype == CTDET_DT_F32 || (dom_dtype == CTDET_DT_F16 && dtype == CTDET_DT_F16),
              "dcn_col2im_coord: dom dtype %d with data dtype %d", dom_dtype, dtype);
  if (dtype == CTDET_DT_F32 || dtype == CTDET_DT_F16X3)
    return launch_dcn_col2im_coord_f32((const float*)dcol, (const float*)x, x_stride, om, om_stride, dx, (float*)dom, dom_stride,
                                       B, H, W, Cin, mask_is_prob, dcol_chunked, dtype == CTDET_DT_F16X3, (hipStream_t)stream);
  return launch_dcn_col2im_coord((const f16*)dcol, (const f16*)x, x_stride, om, om_stride, dx, dom, dom_stride,
                                 dom_dtype == CTDET_DT_F16, B, H, W, Cin, mask_is_prob, dcol_chunked, (hipStream_t)stream);
}

int32_t ctdet_dcn_col2im_fused(const float* dy, int32_t dy_stride, int32_t K, const void* w_packed, const float* w_scale,
                               const float* x, int32_t x_stride, const float* om, int32_t om_stride, float* dx, float* dom,
                               int32_t dom_stride, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t mask_is_prob, void* stream) {
  CTDET_CHECK(dy && w_packed && w_scale && x && om && dx && dom, "dcn_col2im_fused: null pointer");
  return launch_dcn_col2im_fused(dy, dy_stride, K, w_packed, w_scale, x, x_stride, om, om_stride, dx, dom, dom_stride, B, H, W, Cin,
                                 mask_is_prob, (hipStream_t)stream);
}

int32_t ctdet_sgd_momentum(float* param, const float* grad, float* momentum_buf, int64_t n, const float* lr_dev,
                           float momentum, float weight_decay, int32_t first_step, void* stream) {
  CTDET_CHECK(param && grad && momentum_buf && lr_dev, "sgd: null pointer");
  return launch_sgd(param, grad, momentum_buf, (long)n, lr_dev, momentum, weight_decay, first_step, (hipStream_t)stream);
}

int32_t ctdet_sgd_momentum_runs(float* param, const float* grad, float* momentum_buf, int64_t n, const int64_t* run_end,
                                const int32_t* run_lr_index, const float* run_weight_decay, const float* lr_table,
                                int32_t nruns, float momentum, int32_t first_step, void* stream) {
  CTDET_CHECK(param && grad && momentum_buf && run_end && run_lr_index && run_weight_decay && lr_table, "sgd_runs: null pointer");
  return launch_sgd_runs(param, grad, momentum_buf, (long)n, (const long*)run_end, run_lr_index, run_weight_decay, lr_table,
                         nruns, momentum, first_step, (hipStream_t)stream);
}

}  // extern "C"
