// HBM-bound helper kernels of the DLA-34 forward path (NHWC, 16-byte vector accesses).
//   preprocess   : CenterNet.preprocess_image  (detectron2/modeling/meta_arch/centernet.py:173-185)
//                  x/255, (x-mean)/std, zero pad to size_divisibility, CHW -> NHWC(8 ch, 3 used)
//   maxpool2x2   : Tree.downsample = nn.MaxPool2d(stride)   (detectron2/modeling/backbone/dla.py:128-129,139)
//   dwconvT_add  : IDAUp `up_i` depthwise ConvTranspose2d(k=2f, s=f, p=f/2, groups=C) fused with the
//                  `layers[i] + layers[i-1]` that feeds `node_i`  (dla.py:162-177)
#include "common.h"

template <typename TIn, typename TOut>
__global__ void __launch_bounds__(256) preprocess_kernel(const TIn* __restrict__ img, TOut* __restrict__ out, int B,
                                                         int H, int W, int Hp, int Wp, long img_batch_stride,
                                                         float m0, float m1, float m2, float s0, float s1, float s2,
                                                         int out_stride, int border) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)B * Hp * Wp;
  if (idx >= total) return;
  const int wx = (int)(idx % Wp);
  const long t = idx / Wp;
  const int hy = (int)(t % Hp);
  const int b = (int)(t / Hp);
  float v[3] = {0.f, 0.f, 0.f};
  if (hy < H && wx < W) {
    const TIn* p = img + (long)b * img_batch_stride + (long)hy * W + wx;
    const long plane = (long)H * W;
    // same operation order as the reference: (x / 255 - mean) / std, fp32
    v[0] = ((float)p[0] / 255.f - m0) / s0;
    v[1] = ((float)p[plane] / 255.f - m1) / s1;
    v[2] = ((float)p[2 * plane] / 255.f - m2) / s2;
  }
  // `border` > 0: the output buffer is [B, Hp+2*border, Wp+2*border] with a zero frame the caller cleared once
  // (lets the 7x7 stem run as a pad-0 convolution without bounds checks); only the interior is written here
  TOut* o = out + (((long)b * (Hp + 2 * border) + hy + border) * (Wp + 2 * border) + wx + border) * out_stride;
  if constexpr (sizeof(TOut) == 2) {
    f16x8 r = {(f16)v[0], (f16)v[1], (f16)v[2], 0, 0, 0, 0, 0};
    *(f16x8*)o = r;
  } else {
    *(f32x4*)o = (f32x4){v[0], v[1], v[2], 0.f};
    if (out_stride >= 8) *(f32x4*)(o + 4) = (f32x4){0.f, 0.f, 0.f, 0.f};   // f32 pixels may also be 4 channels wide
  }
}

// generic channel-vector helpers: VEC elements per thread (8 f16 = 16 B, 4 f32 = 16 B)
template <typename T> struct Vec;
template <> struct Vec<f16> { typedef f16x8 type; static constexpr int N = 8; };
template <> struct Vec<float> { typedef f32x4 type; static constexpr int N = 4; };

template <typename T>
__global__ void __launch_bounds__(256) maxpool2x2_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W,
                                                         int C, int in_stride, int out_stride) {
  using V = typename Vec<T>::type;
  constexpr int N = Vec<T>::N;
  const int Ho = H / 2, Wo = W / 2, CV = C / N;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * Ho * Wo * CV) return;
  const int cv = (int)(idx % CV);
  long t = idx / CV;
  const int wo = (int)(t % Wo); t /= Wo;
  const int ho = (int)(t % Ho);
  const int b = (int)(t / Ho);
  const T* p = x + ((long)(b * H + 2 * ho) * W + 2 * wo) * in_stride + cv * N;
  const V v00 = *(const V*)p, v01 = *(const V*)(p + in_stride);
  const V v10 = *(const V*)(p + (long)W * in_stride), v11 = *(const V*)(p + (long)(W + 1) * in_stride);
  V r;
#pragma unroll
  for (int e = 0; e < N; ++e) {
    const float a0 = (float)v00[e] > (float)v01[e] ? (float)v00[e] : (float)v01[e];
    const float a1 = (float)v10[e] > (float)v11[e] ? (float)v10[e] : (float)v11[e];
    r[e] = (T)(a0 > a1 ? a0 : a1);
  }
  *(V*)(y + ((long)(b * Ho + ho) * Wo + wo) * out_stride + cv * N) = r;
}

// BasicStem pooling of the ResNet backbones: F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
// (detectron2/modeling/backbone/resnet.py:341-345); padded taps are skipped (= -inf padding).
template <typename T>
__global__ void __launch_bounds__(256) maxpool3x3s2_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W,
                                                           int C, int in_stride, int out_stride, int pad, int Ho, int Wo) {
  using V = typename Vec<T>::type;
  constexpr int N = Vec<T>::N;
  const int CV = C / N;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * Ho * Wo * CV) return;
  const int cv = (int)(idx % CV);
  long t = idx / CV;
  const int wo = (int)(t % Wo); t /= Wo;
  const int ho = (int)(t % Ho);
  const int b = (int)(t / Ho);
  float m[N];
#pragma unroll
  for (int e = 0; e < N; ++e) m[e] = -INFINITY;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy) {
    const int yy = 2 * ho + dy + 1 - pad;
    if (yy < 0 || yy >= H) continue;
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int xx = 2 * wo + dx + 1 - pad;
      if (xx < 0 || xx >= W) continue;
      const V v = *(const V*)(x + ((long)(b * H + yy) * W + xx) * in_stride + cv * N);
#pragma unroll
      for (int e = 0; e < N; ++e) m[e] = (float)v[e] > m[e] ? (float)v[e] : m[e];
    }
  }
  V r;
#pragma unroll
  for (int e = 0; e < N; ++e) r[e] = (T)m[e];
  *(V*)(y + ((long)(b * Ho + ho) * Wo + wo) * out_stride + cv * N) = r;
}

// Weight packing for the f16 conv kernels in one launch: f32 OIHW parameter -> f16 [Cout_pad][Kpad] with the K order
// the kernels expect (korder 0: k = tap*Cin_pad + c; korder 1: k = ((c/32)*R*S + tap)*32 + c%32), zero padding
// included.  transposed != 0 packs the input-gradient form instead: rows = original input channels, k-channels =
// original output channels, taps flipped (dX = conv(dY, W^T flipped)).  Replaces a chain of ~8 small torch kernels
// per conv and training step (permute, pad, reshape, zeros, cast, copy).
// element (row, k) of the packed operand.  transposed: 0 forward, 1 input-gradient (rows = I, channels = O, taps flipped),
// 2 / 3: the operand of DCNv2's d(columns) contraction for a [O, I, 3, 3] weight -- a 1x1 conv from the O channels of dY to
// 9*I column channels, row order tap-major (tap*I + c) or chunk-major ((c/32)*288 + tap*32 + c%32, what the scatter kernel
// reads in contiguous runs): value w[o][c][tap], no permuted copy of the weight in between
__device__ __forceinline__ float pack_value(const float* __restrict__ w, int O, int I, int R, int S, int chans_pad, int korder,
                                            int transposed, int row, int k) {
  if (transposed >= 2) {
    if (row >= 9 * I || k >= O) return 0.f;
    int tap, c;
    if (transposed == 2) { tap = row / I; c = row - tap * I; }
    else { const int chunk = row / 288, rem = row - chunk * 288; tap = rem >> 5; c = chunk * 32 + (rem & 31); }
    return w[((long)k * I + c) * 9 + tap];
  }
  const int rows = transposed ? I : O, chans = transposed ? O : I;
  const int RS = R * S;
  int tap, c;
  if (korder == 0) { tap = k / chans_pad; c = k - tap * chans_pad; }
  else { const int chunk = k / (RS * 32), rem = k - chunk * (RS * 32); tap = rem >> 5; c = chunk * 32 + (rem & 31); }
  if (row >= rows || tap >= RS || c >= chans) return 0.f;
  const int r = tap / S, s2 = tap - r * S;
  if (!transposed) return w[(((long)row * I + c) * R + r) * S + s2];
  return w[(((long)c * I + row) * R + (R - 1 - r)) * S + (S - 1 - s2)];
}

__global__ void __launch_bounds__(256) pack_weights_kernel(const float* __restrict__ w, f16* __restrict__ out, int O, int I,
                                                           int R, int S, int chans_pad, int rows_pad, int Kpad, int korder,
                                                           int transposed) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)rows_pad * Kpad) return;
  out[idx] = (f16)pack_value(w, O, I, R, S, chans_pad, korder, transposed, (int)(idx / Kpad), (int)(idx % Kpad));
}

// y[b,oy,ox,c] = skip[b,oy,ox,c] + sum_{ky,kx} x[b,iy,ix,c] * w[c,ky,kx],  oy = iy*f - f/2 + ky, k = 2f:
// exactly two input rows/cols contribute per output row/col.  w is f32 [k][k][C] (the PyTorch
// ConvTranspose2d weight [C,1,k,k] transposed once on the host so a tap's channels are contiguous).
template <typename T>
__global__ void __launch_bounds__(256) dwconvT_add_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                          const T* __restrict__ skip, T* __restrict__ y, int B, int H,
                                                          int W, int C, int f, int in_stride, int skip_stride,
                                                          int out_stride) {
  using V = typename Vec<T>::type;
  constexpr int N = Vec<T>::N;
  const int Ho = H * f, Wo = W * f, CV = C / N, k = 2 * f, p = f / 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * Ho * Wo * CV) return;
  const int cv = (int)(idx % CV);
  long t = idx / CV;
  const int ox = (int)(t % Wo); t /= Wo;
  const int oy = (int)(t % Ho);
  const int b = (int)(t / Ho);
  const int iy1 = (oy + p) / f, ky1 = (oy + p) - f * iy1;  // ky1 in [0,f)
  const int ix1 = (ox + p) / f, kx1 = (ox + p) - f * ix1;
  float acc[N];
#pragma unroll
  for (int e = 0; e < N; ++e) acc[e] = 0.f;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy) {
    const int iy = iy1 - dy, ky = ky1 + dy * f;
    if (iy < 0 || iy >= H) continue;
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int ix = ix1 - dx, kx = kx1 + dx * f;
      if (ix < 0 || ix >= W) continue;
      const V xv = *(const V*)(x + ((long)(b * H + iy) * W + ix) * in_stride + cv * N);
      const float* wt = w + (long)(ky * k + kx) * C + cv * N;
#pragma unroll
      for (int e4 = 0; e4 < N; e4 += 4) {
        const f32x4 wv = *(const f32x4*)(wt + e4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e4 + e] = fmaf((float)xv[e4 + e], wv[e], acc[e4 + e]);
      }
    }
  }
  const long opix = (long)(b * Ho + oy) * Wo + ox;
  V r;
  if (skip) {
    const V sv = *(const V*)(skip + opix * skip_stride + cv * N);
#pragma unroll
    for (int e = 0; e < N; ++e) r[e] = (T)(acc[e] + (float)sv[e]);
  } else {
#pragma unroll
    for (int e = 0; e < N; ++e) r[e] = (T)acc[e];
  }
  *(V*)(y + opix * out_stride + cv * N) = r;
}

// the same, for the shapes DLA-34 uses (f = 2, 4, 8; C / N a power of two).  A thread owns one (output column, channel
// vector) and walks DW_ROWS output rows of one kernel-row phase (oy, oy + f, oy + 2f, ...): those rows use the same four
// taps, so the weights are loaded once into registers, and consecutive rows of a phase share an input row, so each output
// costs two input loads, the skip load and the store.  The generic kernel above issues 13 loads per 16 output bytes (8 of
// them weights) plus five runtime integer divisions and reaches 3.2 TB/s.
constexpr int DW_ROWS = 8;
template <typename T, int F>
__global__ void __launch_bounds__(256) dwconvT_add_rows_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                               const T* __restrict__ skip, T* __restrict__ y, int H, int W,
                                                               int C, int cv_shift, int nchunk, int in_stride,
                                                               int skip_stride, int out_stride) {
  using V = typename Vec<T>::type;
  constexpr int N = Vec<T>::N;
  constexpr int K = 2 * F, P = F / 2;
  const int Ho = H * F, Wo = W * F;
  const int phase = blockIdx.y % F;
  const int chunk = (blockIdx.y / F) % nchunk;
  const int b = blockIdx.y / (F * nchunk);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= (Wo << cv_shift)) return;
  const int cv = i & ((1 << cv_shift) - 1), ox = i >> cv_shift;
  const int ky1 = (phase + P) % F, iyo = (phase + P) / F;   // output row (j*F + phase) reads input rows j + iyo, j + iyo - 1
  const int ix1 = (ox + P) / F, kx1 = (ox + P) - F * ix1;
  const bool c1 = ix1 < W, c0 = ix1 >= 1;                   // ix1 >= 0 and ix1 - 1 < W always hold
  // taps [dy][dx]: kernel row ky1 + dy*F, kernel column kx1 + dx*F
  float wt[2][2][N];
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx)
#pragma unroll
      for (int e4 = 0; e4 < N; e4 += 4) {
        const f32x4 wv = *(const f32x4*)(w + ((long)(ky1 + dy * F) * K + kx1 + dx * F) * C + cv * N + e4);
#pragma unroll
        for (int e = 0; e < 4; ++e) wt[dy][dx][e4 + e] = wv[e];
      }
  V zero;
#pragma unroll
  for (int e = 0; e < N; ++e) zero[e] = (T)0.f;
  const T* xb = x + (long)b * H * W * in_stride + cv * N;
  auto load_row = [&](int iy, V& v1, V& v0) {               // input row iy, columns ix1 and ix1 - 1 (zero outside)
    const bool rin = iy >= 0 && iy < H;
    const T* xr = xb + (long)iy * W * in_stride;
    v1 = (rin && c1) ? *(const V*)(xr + (long)ix1 * in_stride) : zero;
    v0 = (rin && c0) ? *(const V*)(xr + (long)(ix1 - 1) * in_stride) : zero;
  };
  const int j0 = chunk * DW_ROWS;
  V lo1, lo0, hi1, hi0;                                      // rows iy - 1 (dy = 1) and iy (dy = 0)
  load_row(j0 + iyo - 1, lo1, lo0);
#pragma unroll
  for (int r = 0; r < DW_ROWS; ++r) {
    const int j = j0 + r, oy = j * F + phase;
    if (oy >= Ho) break;
    load_row(j + iyo, hi1, hi0);
    float acc[N];
#pragma unroll
    for (int e = 0; e < N; ++e) {
      acc[e] = (float)hi1[e] * wt[0][0][e];
      acc[e] = fmaf((float)hi0[e], wt[0][1][e], acc[e]);
      acc[e] = fmaf((float)lo1[e], wt[1][0][e], acc[e]);
      acc[e] = fmaf((float)lo0[e], wt[1][1][e], acc[e]);
    }
    const long opix = (long)(b * Ho + oy) * Wo + ox;
    V o;
    if (skip) {
      const V sv = *(const V*)(skip + opix * skip_stride + cv * N);
#pragma unroll
      for (int e = 0; e < N; ++e) o[e] = (T)(acc[e] + (float)sv[e]);
    } else {
#pragma unroll
      for (int e = 0; e < N; ++e) o[e] = (T)acc[e];
    }
    *(V*)(y + opix * out_stride + cv * N) = o;
    lo1 = hi1; lo0 = hi0;
  }
}

static inline unsigned nblk(long n) { return (unsigned)((n + 255) / 256); }

int launch_preprocess(const void* img, int img_dtype, void* out, int out_dtype, int B, int H, int W, int Hp, int Wp,
                      long img_batch_stride, const float* mean, const float* stdv, int out_stride, int border,
                      hipStream_t s) {
  CTDET_CHECK((out_stride >= 8 && out_stride % 8 == 0) || (out_stride == 4 && out_dtype == CTDET_F32),
              "preprocess: out_stride=%d must be a multiple of 8 (or 4 for f32 output)", out_stride);
  CTDET_CHECK(Hp >= H && Wp >= W && border >= 0, "preprocess: padded size smaller than image");
  const long total = (long)B * Hp * Wp;
  if (total == 0) return 0;
#define PP(TI, TO)                                                                                               \
  hipLaunchKernelGGL((preprocess_kernel<TI, TO>), dim3(nblk(total)), dim3(256), 0, s, (const TI*)img, (TO*)out, \
                     B, H, W, Hp, Wp, img_batch_stride, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2], out_stride, border)
  if (img_dtype == CTDET_U8 && out_dtype == CTDET_F16) PP(uint8_t, f16);
  else if (img_dtype == CTDET_U8 && out_dtype == CTDET_F32) PP(uint8_t, float);
  else if (img_dtype == CTDET_F32 && out_dtype == CTDET_F16) PP(float, f16);
  else if (img_dtype == CTDET_F32 && out_dtype == CTDET_F32) PP(float, float);
  else CTDET_CHECK(false, "preprocess: unsupported dtypes in=%d out=%d", img_dtype, out_dtype);
#undef PP
  CTDET_LAUNCH_CHECK();
  return 0;
}

int launch_maxpool2x2(const void* x, void* y, int dtype, int B, int H, int W, int C, int in_stride, int out_stride,
                      hipStream_t s) {
  CTDET_CHECK(H % 2 == 0 && W % 2 == 0, "maxpool2x2: odd spatial size %dx%d", H, W);
  const int N = dtype == CTDET_F16 ? 8 : 4;
  CTDET_CHECK(C % N == 0 && in_stride % N == 0 && out_stride % N == 0, "maxpool2x2: channels must be multiples of %d", N);
  const long total = (long)B * (H / 2) * (W / 2) * (C / N);
  if (total == 0) return 0;
  if (dtype == CTDET_F16)
    hipLaunchKernelGGL((maxpool2x2_kernel<f16>), dim3(nblk(total)), dim3(256), 0, s, (const f16*)x, (f16*)y, B, H, W, C,
                       in_stride, out_stride);
  else if (dtype == CTDET_F32)
    hipLaunchKernelGGL((maxpool2x2_kernel<float>), dim3(nblk(total)), dim3(256), 0, s, (const float*)x, (float*)y, B, H,
                       W, C, in_stride, out_stride);
  else CTDET_CHECK(false, "maxpool2x2: bad dtype %d", dtype);
  CTDET_LAUNCH_CHECK();
  return 0;
}

// pad 1: F.max_pool2d(x, 3, 2, 1) (BasicStem); pad 0 + ceil: nn.MaxPool2d(3, 2, ceil_mode=True) (VoVNet stages): the last
// window may hang over the bottom / right edge, positions outside the map do not take part
int launch_maxpool3x3s2(const void* x, void* y, int dtype, int B, int H, int W, int C, int in_stride, int out_stride,
                        int ceil_nopad, hipStream_t s) {
  const int N = dtype == CTDET_F16 ? 8 : 4;
  CTDET_CHECK(C % N == 0 && in_stride % N == 0 && out_stride % N == 0, "maxpool3x3s2: channels must be multiples of %d", N);
  int Ho, Wo, pad;
  if (ceil_nopad) {
    pad = 0;
    Ho = (H - 3 + 1) / 2 + 1; Wo = (W - 3 + 1) / 2 + 1;             // ceil((H - 3) / 2) + 1
    if ((Ho - 1) * 2 >= H) --Ho;                                    // the last window must start inside the map
    if ((Wo - 1) * 2 >= W) --Wo;
    CTDET_CHECK(H >= 3 && W >= 3, "maxpool3x3s2(ceil): map %dx%d smaller than the window", H, W);
  } else {
    pad = 1;
    Ho = (H - 1) / 2 + 1; Wo = (W - 1) / 2 + 1;
  }
  const long total = (long)B * Ho * Wo * (C / N);
  if (total == 0) return 0;
  if (dtype == CTDET_F16)
    hipLaunchKernelGGL((maxpool3x3s2_kernel<f16>), dim3(nblk(total)), dim3(256), 0, s, (const f16*)x, (f16*)y, B, H, W, C,
                       in_stride, out_stride, pad, Ho, Wo);
  else if (dtype == CTDET_F32)
    hipLaunchKernelGGL((maxpool3x3s2_kernel<float>), dim3(nblk(total)), dim3(256), 0, s, (const float*)x, (float*)y, B,
                       H, W, C, in_stride, out_stride, pad, Ho, Wo);
  else CTDET_CHECK(false, "maxpool3x3s2: bad dtype %d", dtype);
  CTDET_LAUNCH_CHECK();
  return 0;
}

// ---- eSE attention of VoVNet (vovnet.py:200-213): out[b][c] = mean over the pixels of x[b, :, :, c] (f32), and
// y = x * hsigmoid(s[b][c]) (+ identity), hsigmoid(v) = relu6(v + 3) / 6
template <typename T>
__global__ void __launch_bounds__(256) global_avgpool_kernel(const T* __restrict__ x, int stride, int HW, int C,
                                                             float* __restrict__ out) {
  __shared__ float red[256];
  const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6;
  float acc = 0.f;
  if (c < C)
    for (int p = pl; p < HW; p += 4) acc += (float)x[((long)b * HW + p) * stride + c];
  red[threadIdx.x] = acc;
  __syncthreads();
  if (pl == 0 && c < C)
    out[(long)b * C + c] = ((red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192])) / (float)HW;
}

template <typename T>
__global__ void __launch_bounds__(256) ese_scale_kernel(const T* __restrict__ x, int x_stride, const float* __restrict__ s,
                                                        const T* __restrict__ idn, int idn_stride, T* __restrict__ y,
                                                        int y_stride, int B, int HW, int C) {
  using V = typename Vec<T>::type;
  constexpr int N = Vec<T>::N;
  const int CV = C / N;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * HW * CV) return;
  const int cv = (int)(idx % CV);
  const long m = idx / CV;
  const int b = (int)(m / HW);
  const V v = *(const V*)(x + m * x_stride + cv * N);
  V r;
  V id = v;
  if (idn) id = *(const V*)(idn + m * idn_stride + cv * N);
#pragma unroll
  for (int e = 0; e < N; ++e) {
    const float g = fminf(fmaxf(s[(long)b * C + cv * N + e] + 3.f, 0.f), 6.f) / 6.f;
    float f = (float)v[e] * g;
    if (idn) f += (float)id[e];
    r[e] = (T)f;
  }
  *(V*)(y + m * y_stride + cv * N) = r;
}

// Are all values of an [M][C] f32 map (pixel stride `stride`) finite?  *flag (set to 1 by the caller) becomes 0 otherwise.  The eval step's per-step guard on the size /
// offset maps: torch's isfinite().all() reduces a 4 M element map through a semaphore buffer that it clears with
// hipMemsetAsync -- a memset node in the captured step, which holds kernels only (engine/graph_nodes.py).
__global__ void __launch_bounds__(256) finite_flag_kernel(const float* __restrict__ x, long M, int C, int stride,
                                                          int* __restrict__ flag) {
  const long n = M * C, step = (long)gridDim.x * 256;
  bool bad = false;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += step) {
    const long m = i / C;
    const int c = (int)(i - m * C);
    bad |= !(fabsf(x[m * stride + c]) <= 3.402823466e38f);      // NaN compares false
  }
  if (bad) *flag = 0;
}
int launch_finite_flag(const float* x, long M, int C, int stride, int* flag, hipStream_t s) {
  if (M * C == 0) return 0;
  CTDET_CHECK(C >= 1 && stride >= C, "finite_flag: bad channel count / pixel stride (%d, %d)", C, stride);
  const long nb = (M * C + 255) / 256;
  hipLaunchKernelGGL(finite_flag_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, s, x, M, C, stride, flag);
  CTDET_LAUNCH_CHECK();
  return 0;
}

int launch_global_avgpool(const void* x, int dtype, int B, int HW, int C, int stride, float* out, hipStream_t s) {
  if ((long)B * HW * C == 0) return 0;
  const dim3 grid((C + 63) / 64, B);
  if (dtype == CTDET_F16) hipLaunchKernelGGL((global_avgpool_kernel<f16>), grid, dim3(256), 0, s, (const f16*)x, stride, HW, C, out);
  else if (dtype == CTDET_F32) hipLaunchKernelGGL((global_avgpool_kernel<float>), grid, dim3(256), 0, s, (const float*)x, stride, HW, C, out);
  else CTDET_CHECK(false, "global_avgpool: bad dtype %d", dtype);
  CTDET_LAUNCH_CHECK();
  return 0;
}

int launch_ese_scale(const void* x, int x_stride, const float* sc, const void* idn, int idn_stride, void* y, int y_stride,
                     int dtype, int B, int HW, int C, hipStream_t s) {
  const int N = dtype == CTDET_F16 ? 8 : 4;
  CTDET_CHECK(C % N == 0 && x_stride % N == 0 && y_stride % N == 0 && (!idn || idn_stride % N == 0),
              "ese_scale: channels / strides must be multiples of %d", N);
  const long total = (long)B * HW * (C / N);
  if (total == 0) return 0;
  if (dtype == CTDET_F16)
    hipLaunchKernelGGL((ese_scale_kernel<f16>), dim3(nblk(total)), dim3(256), 0, s, (const f16*)x, x_stride, sc, (const f16*)idn,
                       idn_stride, (f16*)y, y_stride, B, HW, C);
  else if (dtype == CTDET_F32)
    hipLaunchKernelGGL((ese_scale_kernel<float>), dim3(nblk(total)), dim3(256), 0, s, (const float*)x, x_stride, sc,
                       (const float*)idn, idn_stride, (float*)y, y_stride, B, HW, C);
  else CTDET_CHECK(false, "ese_scale: bad dtype %d", dtype);
  CTDET_LAUNCH_CHECK();
  return 0;
}

// All conv weights of a model in one launch (training: every weight changes every step -- ~160 pack launches of a few
// microseconds each otherwise).  The descriptor table lives in device memory; blk0 is the running block count.
struct PackDesc {
  const float* w; f16* out;
  int O, I, R, S, chans_pad, rows_pad, Kpad, korder, transposed, blk0;
};
__global__ void __launch_bounds__(256) pack_weights_batch_kernel(const PackDesc* __restrict__ table, int n) {
  int lo = 0, hi = n - 1;                         // last descriptor with blk0 <= blockIdx.x (block-uniform)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PackDesc d = table[lo];
  const long idx = (long)((int)blockIdx.x - d.blk0) * 256 + threadIdx.x;
  if (idx >= (long)d.rows_pad * d.Kpad) return;
  d.out[idx] = (f16)pack_value(d.w, d.O, d.I, d.R, d.S, d.chans_pad, d.korder, d.transposed, (int)(idx / d.Kpad), (int)(idx % d.Kpad));
}

int launch_pack_weights_batch(const void* table_dev, int n, int total_blocks, hipStream_t s) {
  if (n == 0 || total_blocks == 0) return 0;
  hipLaunchKernelGGL(pack_weights_batch_kernel, dim3(total_blocks), dim3(256), 0, s, (const PackDesc*)table_dev, n);
  CTDET_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Weight packing for the f16x3 kernels (f32 tensors, products as hi*hi + lo*hi + hi*lo on the f16 matrix pipe), one
// workgroup per packed row.  A row is first scaled by the power of two that brings its largest magnitude into [1024, 2048)
// (exact; the lo halves of the row's significant weights then stay f16 normals whatever the layer's weight scale is), the
// inverse goes to scale_out[row] -- the conv epilogue's per-cout multiplier.  Layouts (ctdet_pack3_desc.layout):
//   0  split tap-major image [rows_pad][Kpad f32 units], k = tap*chans_pad + c, every group of 4 k = 16 bytes {hi[4], lo[4]} f16
//      (what ctdet_split_weights makes of an f32 image): 1x1 / strided / 7x7 convs, DCNv2, the d(columns) operand
//   3  tap-pair image of conv3x3_halo_pair2_kernel (chans_pad % 32 == 0): per chunk PAIR (A, B) of 16-channel chunks nine
//      128-byte steps {X, Y}: X = for q in 0..3 {hi[a][4q..4q+3], hi[b][4q..4q+3]}, Y likewise from lo, with (a, b) =
//      taps (2s, 2s+1) of A for s = 0..3, then tap 8 of A with tap 8 of B, then taps (2s, 2s+1) of B;  Kpad = chans_pad/32*288
//   2  the same for an odd number of chunks (conv3x3_halo_pair_kernel): per chunk five steps, taps (2s, 2s+1), the tenth tap
//      zero;  Kpad = chans_pad/16*160
//   5  tap-major rows of Kpad f32 units (Kpad % 8 == 0), every group of 8 k = 32 bytes {hi[8], lo[8]}: operands that a kernel
//      reads straight from memory as 16x16x32 MFMA fragments (the d(columns) GEMM inside dcn_col2im_window_kernel<FUSED>)
// transposed as in ctdet_pack_weights (0 forward, 1 input-gradient operand, 2 / 3 DCNv2's d(columns) operand).
// ------------------------------------------------------------------------------------------------
struct Pack3Desc {
  const float* w; void* out; float* scale_out;
  int O, I, R, S, chans_pad, rows_pad, Kpad, layout, transposed, scale_n, blk0;
  int pad_;
};

__device__ __forceinline__ void pack3_row(const Pack3Desc& d, int row, float* red) {
  const int tid = threadIdx.x;
  const int rows = d.transposed >= 2 ? 9 * d.I : (d.transposed ? d.I : d.O);
  const int K = d.R * d.S * d.chans_pad;
  float amax = 0.f;
  if (row < rows)
    for (int k = tid; k < K; k += 256)
      amax = fmaxf(amax, fabsf(pack_value(d.w, d.O, d.I, d.R, d.S, d.chans_pad, 0, d.transposed, row, k)));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  int ex = 0;
  (void)frexpf(amax, &ex);                       // amax = f * 2^ex, f in [0.5, 1): floor(log2(amax)) = ex - 1
  const float pw = amax > 0.f ? ldexpf(1.f, 11 - ex) : 1.f;
  if (tid == 0 && row < d.scale_n) d.scale_out[row] = amax > 0.f ? ldexpf(1.f, ex - 11) : 1.f;
  auto val = [&](int tap, int c) -> float {      // scaled weight of (row, tap, channel); taps >= R*S and padding are zero
    if (row >= rows || tap >= d.R * d.S) return 0.f;
    return pw * pack_value(d.w, d.O, d.I, d.R, d.S, d.chans_pad, 0, d.transposed, row, tap * d.chans_pad + c);
  };
  f16* const orow = (f16*)d.out + (long)row * d.Kpad * 2;
  if (d.layout == 0) {
    for (int g = tid; g < d.Kpad / 4; g += 256) {
      f16x8 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 4 * g + j;
        const float v = (k < K && row < rows) ? pw * pack_value(d.w, d.O, d.I, d.R, d.S, d.chans_pad, 0, d.transposed, row, k) : 0.f;
        const f16 h = (f16)v;
        o[j] = h; o[4 + j] = (f16)(v - (float)h);
      }
      *(f16x8*)(orow + 8 * g) = o;
    }
    return;
  }
  if (d.layout == 5) {      // groups of 8 k: {hi[8], lo[8]} (the A fragments of a 16x16x32 MFMA straight from memory)
    for (int g = tid; g < d.Kpad / 8; g += 256) {
      f16x8 oh, ol;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 8 * g + j;
        const float v = (k < K && row < rows) ? pw * pack_value(d.w, d.O, d.I, d.R, d.S, d.chans_pad, 0, d.transposed, row, k) : 0.f;
        const f16 h = (f16)v;
        oh[j] = h; ol[j] = (f16)(v - (float)h);
      }
      *(f16x8*)(orow + 16 * g) = oh;
      *(f16x8*)(orow + 16 * g + 8) = ol;
    }
    return;
  }
  // pair layouts: one 16-byte piece = {operand a: 4 channels, operand b: 4 channels} of X (hi) or Y (lo)
  const int npieces = d.Kpad / 4;
  for (int pc = tid; pc < npieces; pc += 256) {
    const int q = pc & 3, xy = (pc >> 2) & 1, st = pc >> 3;       // piece = ((step * 2 + xy) * 4 + q)
    int ta, tb, ca, cb;                                            // (tap, first channel) of the two operands
    if (d.layout == 3) {
      const int cp = st / 9, s9 = st - cp * 9;
      if (s9 < 4) { ta = 2 * s9; tb = ta + 1; ca = cb = 32 * cp; }
      else if (s9 == 4) { ta = tb = 8; ca = 32 * cp; cb = ca + 16; }
      else { ta = 2 * (s9 - 5); tb = ta + 1; ca = cb = 32 * cp + 16; }
    } else {
      const int ch = st / 5, s5 = st - ch * 5;
      ta = 2 * s5; tb = ta + 1; ca = cb = 16 * ch;
    }
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float va = val(ta, ca + 4 * q + j), vb = val(tb, cb + 4 * q + j);
      const f16 ha = (f16)va, hb = (f16)vb;
      o[j] = xy ? (f16)(va - (float)ha) : ha;
      o[4 + j] = xy ? (f16)(vb - (float)hb) : hb;
    }
    *(f16x8*)(orow + 8 * pc) = o;
  }
}

__global__ void __launch_bounds__(256) pack3_kernel(const Pack3Desc d) {
  __shared__ float red[4];
  pack3_row(d, blockIdx.x, red);
}
__global__ void __launch_bounds__(256) pack3_batch_kernel(const Pack3Desc* __restrict__ table, int n) {
  __shared__ float red[4];
  int lo = 0, hi = n - 1;                         // last descriptor with blk0 <= blockIdx.x (block-uniform)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const Pack3Desc d = table[lo];
  const int row = (int)blockIdx.x - d.blk0;
  if (row >= d.rows_pad) return;
  pack3_row(d, row, red);
}

static int pack3_check(const Pack3Desc& d) {
  CTDET_CHECK(d.w && d.out && d.scale_out, "pack_weights_x3: null pointer");
  const int taps = d.R * d.S;
  if (d.transposed >= 2) {
    CTDET_CHECK(d.transposed <= 3 && d.R == 1 && d.S == 1 && (d.layout == 0 || d.layout == 5) && d.chans_pad >= d.O &&
                    d.rows_pad >= 9 * d.I && (d.transposed == 2 || d.I % 32 == 0),
                "pack_weights_x3: the DCNv2 d(columns) operand is a 1x1 contraction (R = S = 1 here) over a [O,I,3,3] weight, layout 0 or 5, rows_pad >= 9*I");
  } else {
    const int rows = d.transposed ? d.I : d.O, chans = d.transposed ? d.O : d.I;
    CTDET_CHECK(d.chans_pad >= chans && d.rows_pad >= rows, "pack_weights_x3: padded sizes too small");
  }
  CTDET_CHECK(d.scale_n >= 0 && d.scale_n <= d.rows_pad, "pack_weights_x3: scale_n=%d beyond rows_pad=%d", d.scale_n, d.rows_pad);
  if (d.layout == 0) CTDET_CHECK(d.Kpad % 4 == 0 && d.Kpad >= taps * d.chans_pad, "pack_weights_x3: Kpad=%d too small / not a multiple of 4", d.Kpad);
  else if (d.layout == 3) CTDET_CHECK(taps == 9 && d.chans_pad % 32 == 0 && d.Kpad == d.chans_pad / 32 * 288, "pack_weights_x3: layout 3 needs a 3x3 kernel, channels %% 32 == 0, Kpad = channels/32*288");
  else if (d.layout == 2) CTDET_CHECK(taps == 9 && d.chans_pad % 16 == 0 && d.Kpad == d.chans_pad / 16 * 160, "pack_weights_x3: layout 2 needs a 3x3 kernel, channels %% 16 == 0, Kpad = channels/16*160");
  else if (d.layout == 5) CTDET_CHECK(d.Kpad % 8 == 0 && d.Kpad >= taps * d.chans_pad, "pack_weights_x3: layout 5 needs Kpad %% 8 == 0 and >= R*S*chans_pad");
  else CTDET_CHECK(false, "pack_weights_x3: bad layout %d", d.layout);
  CTDET_CHECK((((size_t)d.out) & 15) == 0, "pack_weights_x3: output must be 16-byte aligned");
  return 0;
}

int launch_pack_weights_x3(const float* w, void* out, float* scale_out, int O, int I, int R, int S, int chans_pad, int rows_pad,
                           int Kpad, int layout, int transposed, int scale_n, hipStream_t s) {
  Pack3Desc d = {};
  d.w = w; d.out = out; d.scale_out = scale_out;
  d.O = O; d.I = I; d.R = R; d.S = S; d.chans_pad = chans_pad; d.rows_pad = rows_pad; d.Kpad = Kpad; d.layout = layout;
  d.transposed = transposed; d.scale_n = scale_n;
  const int rc = pack3_check(d);
  if (rc) return rc;
  if (rows_pad == 0) return 0;
  hipLaunchKernelGGL(pack3_kernel, dim3(rows_pad), dim3(256), 0, s, d);
  CTDET_LAUNCH_CHECK();
  return 0;
}

int launch_pack_weights_x3_batch(const void* table_dev, int n, int total_blocks, hipStream_t s) {
  if (n == 0 || total_blocks == 0) return 0;
  hipLaunchKernelGGL(pack3_batch_kernel, dim3(total_blocks), dim3(256), 0, s, (const Pack3Desc*)table_dev, n);
  CTDET_LAUNCH_CHECK();
  return 0;
}

int launch_pack_weights(const float* w, void* out, int O, int I, int R, int S, int chans_pad, int rows_pad, int Kpad,
                        int korder, int transposed, hipStream_t s) {
  if (transposed >= 2) {
    CTDET_CHECK(transposed <= 3 && R == 3 && S == 3 && korder == 0 && chans_pad >= O && Kpad >= chans_pad && rows_pad >= 9 * I &&
                    (transposed == 2 || I % 32 == 0),
                "pack_weights: the DCNv2 d(columns) operand needs a [O,I,3,3] weight, korder 0, rows_pad >= 9*I (chunk-major: I %% 32 == 0)");
  } else {
    const int rows = transposed ? I : O, chans = transposed ? O : I;
    CTDET_CHECK(chans_pad >= chans && rows_pad >= rows && Kpad >= R * S * chans_pad, "pack_weights: padded sizes too small");
    CTDET_CHECK(korder == 0 || (korder == 1 && chans_pad % 32 == 0), "pack_weights: chunk-major needs channels %% 32 == 0");
  }
  const long total = (long)rows_pad * Kpad;
  if (total == 0) return 0;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(nblk(total)), dim3(256), 0, s, w, (f16*)out, O, I, R, S, chans_pad,
                     rows_pad, Kpad, korder, transposed);
  CTDET_LAUNCH_CHECK();
  return 0;
}

int launch_dwconvT_add(const void* x, const float* w, const void* skip, void* y, int dtype, int B, int H, int W, int C,
                       int f, int in_stride, int skip_stride, int out_stride, hipStream_t s) {
  const int N = dtype == CTDET_F16 ? 8 : 4;
  CTDET_CHECK(f >= 1 && (f == 1 || f % 2 == 0), "dwconvT: up factor %d unsupported", f);
  CTDET_CHECK(C % N == 0 && in_stride % N == 0 && out_stride % N == 0 && (!skip || skip_stride % N == 0),
              "dwconvT: channels must be multiples of %d", N);
  const long total = (long)B * H * f * W * f * (C / N);
  if (total == 0) return 0;
  const int CV = C / N;
  const int nchunk = (H + DW_ROWS - 1) / DW_ROWS;           // a block walks DW_ROWS output rows of one phase
  const long gy = (long)B * nchunk * f;
  if ((f == 2 || f == 4 || f == 8) && (CV & (CV - 1)) == 0 && gy <= 65535 && (dtype == CTDET_F16 || dtype == CTDET_F32)) {
    int sh = 0;
    while ((1 << sh) < CV) ++sh;
    const dim3 grid(nblk((long)W * f * CV), (unsigned)gy);
#define DW(T, F_)                                                                                                        \
  hipLaunchKernelGGL((dwconvT_add_rows_kernel<T, F_>), grid, dim3(256), 0, s, (const T*)x, w, (const T*)skip, (T*)y, H, W, C, \
                     sh, nchunk, in_stride, skip_stride, out_stride)
    if (dtype == CTDET_F16) { if (f == 2) DW(f16, 2); else if (f == 4) DW(f16, 4); else DW(f16, 8); }
    else { if (f == 2) DW(float, 2); else if (f == 4) DW(float, 4); else DW(float, 8); }
#undef DW
    CTDET_LAUNCH_CHECK();
    return 0;
  }
  if (dtype == CTDET_F16)
    hipLaunchKernelGGL((dwconvT_add_kernel<f16>), dim3(nblk(total)), dim3(256), 0, s, (const f16*)x, w, (const f16*)skip,
                       (f16*)y, B, H, W, C, f, in_stride, skip_stride, out_stride);
  else if (dtype == CTDET_F32)
    hipLaunchKernelGGL((dwconvT_add_kernel<float>), dim3(nblk(total)), dim3(256), 0, s, (const float*)x, w,
                       (const float*)skip, (float*)y, B, H, W, C, f, in_stride, skip_stride, out_stride);
  else CTDET_CHECK(false, "dwconvT: bad dtype %d", dtype);
  CTDET_LAUNCH_CHECK();
  return 0;
}
