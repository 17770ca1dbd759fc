// Training-side kernels of the conv stack (NHWC; f16 or f32 activations, f32 statistics / weight gradients).
//   BatchNorm2d in training mode (batch statistics, momentum 0.1 running stats) forward and backward, fused with
//   ReLU and the residual add of DLABasicBlock (detectron2/modeling/backbone/dla.py:59-73, 86-94; deform_conv.py:501-519)
//   conv weight gradient on MFMA (reduction over pixels, operands read from LDS with ds_read_b64_tr_b16)
//   MaxPool2d(2) backward, depthwise ConvTranspose2d backward (dla.py:129, 162-177)
//   DCNv2 backward pieces: column sampler and col2im + coordinate/mask gradients
//   (detectron2/layers/csrc/deformable/deform_conv_cuda_kernel.cu:786-1066, deform_conv_cuda.cu:929-1129)
// Input-gradient of plain convs reuses the forward implicit-GEMM kernels with transposed/flipped weights.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// ------------------------------------------------------------------------------------------------
// per-channel reductions over the M rows of an NHWC tensor: thread = (row lane, 8-channel vector)
//   mode 0 (BN forward stats):  r0 = sum y,        r1 = sum y^2
//   mode 1 (BN/act backward):   g = dz * (z > 0 if relu);  r0 = sum g,  r1 = sum g * xhat   (xhat = (y-mean)*invstd)
// partial[blk][2][C] f32, reduced in fixed order by chan_finalize (f64) => deterministic.
// ------------------------------------------------------------------------------------------------
// element type of an activation tensor and its 16-byte vector: f16 (8 channels) or f32 (4 channels: the f32 and f16x3 modes)
template <typename T> struct VecT;
template <> struct VecT<f16> { typedef f16x8 type; static constexpr int N = 8; };
template <> struct VecT<float> { typedef f32x4 type; static constexpr int N = 4; };

template <typename T>
struct ChanRedArgs {
  const T* y; int y_stride;      // mode 0: tensor; mode 1: pre-BN conv output (may be null => no xhat term)
  const T* dz; int dz_stride;    // mode 1
  const T* z; int z_stride;      // mode 1: post-activation output (relu mask), may be null
  const float* mean; const float* invstd;
  int M, C, mode, relu;
  float* partial;
};

template <typename T>
__global__ void __launch_bounds__(256) chan_reduce_kernel(ChanRedArgs<T> a) {
  typedef typename VecT<T>::type V;
  constexpr int N = VecT<T>::N;
  __shared__ float red[2][256][N];
  const int CV = a.C / N;
  const int rows_per_pass = 256 / CV;  // CV <= 256
  const int cv = threadIdx.x % CV, rl = threadIdx.x / CV;
  float s0[N], s1[N];
#pragma unroll
  for (int e = 0; e < N; ++e) s0[e] = s1[e] = 0.f;
  float mu[N], is[N];
  if (a.mode == 1 && a.y) {
#pragma unroll
    for (int e = 0; e < N; ++e) { mu[e] = a.mean[cv * N + e]; is[e] = a.invstd[cv * N + e]; }
  }
  if (rl < rows_per_pass) {
    const long step = (long)gridDim.x * rows_per_pass;
    long m = (long)blockIdx.x * rows_per_pass + rl;
    if (a.mode == 0) {
      // four rows in flight per thread: with one 16-byte load outstanding per thread the pass ran at ~2 TB/s (latency x
      // occupancy), the mid-sized maps at a quarter of that
      for (; m + 3 * step < a.M; m += 4 * step) {
        V v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const V*)(a.y + (m + u * step) * a.y_stride + cv * N);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < N; ++e) { const float f = (float)v[u][e]; s0[e] += f; s1[e] += f * f; }
      }
    } else {
      for (; m + step < a.M; m += 2 * step) {      // two rows of the (up to) three tensors in flight
        V g[2], zz[2], yy[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          g[u] = *(const V*)(a.dz + (m + u * step) * a.dz_stride + cv * N);
          zz[u] = g[u]; yy[u] = g[u];
          if (a.relu) zz[u] = *(const V*)(a.z + (m + u * step) * a.z_stride + cv * N);
          if (a.y) yy[u] = *(const V*)(a.y + (m + u * step) * a.y_stride + cv * N);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int e = 0; e < N; ++e) {
            const float gf = (a.relu && !((float)zz[u][e] > 0.f)) ? 0.f : (float)g[u][e];
            s0[e] += gf;
            if (a.y) s1[e] += gf * (((float)yy[u][e] - mu[e]) * is[e]);
          }
      }
    }
    for (; m < a.M; m += step) {
      if (a.mode == 0) {
        const V v = *(const V*)(a.y + m * a.y_stride + cv * N);
#pragma unroll
        for (int e = 0; e < N; ++e) { const float f = (float)v[e]; s0[e] += f; s1[e] += f * f; }
      } else {
        const V g = *(const V*)(a.dz + m * a.dz_stride + cv * N);
        V zz = g, yy = g;
        if (a.relu) zz = *(const V*)(a.z + m * a.z_stride + cv * N);
        if (a.y) yy = *(const V*)(a.y + m * a.y_stride + cv * N);
#pragma unroll
        for (int e = 0; e < N; ++e) {
          const float gf = (a.relu && !((float)zz[e] > 0.f)) ? 0.f : (float)g[e];
          s0[e] += gf;
          if (a.y) s1[e] += gf * (((float)yy[e] - mu[e]) * is[e]);
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < N; ++e) { red[0][threadIdx.x][e] = s0[e]; red[1][threadIdx.x][e] = s1[e]; }
  __syncthreads();
  // tree over the row lanes of a channel vector (fixed order => deterministic): with few channels (C = 8 ... 32: 256 ... 64
  // row lanes) one thread summing them all was most of the kernel's time on the small maps
  for (int n = rows_per_pass; n > 1;) {
    const int half = (n + 1) >> 1;
    if (rl + half < n) {
#pragma unroll
      for (int e = 0; e < N; ++e) {
        red[0][threadIdx.x][e] += red[0][threadIdx.x + half * CV][e];
        red[1][threadIdx.x][e] += red[1][threadIdx.x + half * CV][e];
      }
    }
    __syncthreads();
    n = half;
  }
  if (rl == 0) {
#pragma unroll
    for (int e = 0; e < N; ++e) {
      a.partial[((long)blockIdx.x * 2 + 0) * a.C + cv * N + e] = red[0][cv][e];
      a.partial[((long)blockIdx.x * 2 + 1) * a.C + cv * N + e] = red[1][cv][e];
    }
  }
}

// mode 0: mean/invstd/scale/shift (+ running stats);  mode 1: out0 = sum g (dbeta / dbias), out1 = sum g*xhat (dgamma)
// threads stride over the block partials, fixed-order f64 shuffle + LDS reduction (deterministic)
__global__ void __launch_bounds__(256) chan_finalize_kernel(const float* __restrict__ partial, int nblocks, int C, int M,
                                                            int mode, float eps, float momentum,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ out0, float* __restrict__ out1,
                                                            float* __restrict__ scale, float* __restrict__ shift,
                                                            float* __restrict__ running_mean,
                                                            float* __restrict__ running_var) {
  // one workgroup per channel (the kernel sits on the step's dependency chain 146 times: with one wave per channel and 16
  // serial loads per lane it took 6 us)
  __shared__ double sh[2][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x;
  double s0 = 0, s1 = 0;
  for (int b = threadIdx.x; b < nblocks; b += 256) { s0 += partial[((long)b * 2 + 0) * C + c]; s1 += partial[((long)b * 2 + 1) * C + c]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_down(s0, o, 64); s1 += __shfl_down(s1, o, 64); }
  if (lane == 0) { sh[0][wave] = s0; sh[1][wave] = s1; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  s0 = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
  s1 = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
  if (mode == 0) {
    const double mean = s0 / M;
    double var = s1 / M - mean * mean;
    if (var < 0) var = 0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    out0[c] = (float)mean;
    out1[c] = (float)invstd;
    const float sc = gamma[c] * (float)invstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
      const double unbiased = M > 1 ? var * M / (M - 1) : var;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  } else {
    // the apply kernel needs the plain sums (scale / shift double as their slots); the parameter gradients leave scaled
    // (eps carries the multiplier in this mode: 1 / (loss scale * world size))
    scale[c] = (float)s0;
    shift[c] = (float)s1;
    out0[c] = (float)s0 * eps;
    out1[c] = (float)s1 * eps;
  }
}

// z = act(y * scale + shift + res)   (BN apply)
template <typename T>
__global__ void __launch_bounds__(256) affine_act_kernel(const T* __restrict__ y, int y_stride,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const T* __restrict__ res, int res_stride, T* __restrict__ z,
                                                         int z_stride, long M, int C, int relu) {
  typedef typename VecT<T>::type V;
  constexpr int N = VecT<T>::N;
  const int CV = C / N;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * CV) return;
  const int cv = (int)(idx % CV);
  const long m = idx / CV;
  const V v = *(const V*)(y + m * y_stride + cv * N);
  V r = v;
  if (res) r = *(const V*)(res + m * res_stride + cv * N);
  V o;
#pragma unroll
  for (int e = 0; e < N; ++e) {
    float f = (float)v[e] * scale[cv * N + e] + shift[cv * N + e];
    if (res) f += (float)r[e];
    if (relu) f = fmaxf(f, 0.f);
    o[e] = (T)f;
  }
  *(V*)(z + m * z_stride + cv * N) = o;
}

// the same for channel-vector counts that are powers of two: a thread owns one channel group, keeps its coefficients in
// registers and walks AA_ROWS pixels, four rows in flight
constexpr int AA_ROWS = 8;
template <typename T>
__global__ void __launch_bounds__(256) affine_act_rows_kernel(const T* __restrict__ y, int y_stride,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              const T* __restrict__ res, int res_stride, T* __restrict__ z,
                                                              int z_stride, long M, int cv_shift, int relu) {
  typedef typename VecT<T>::type V;
  constexpr int N = VecT<T>::N;
  const int CV = 1 << cv_shift;
  const int cv = threadIdx.x & (CV - 1), lane_px = threadIdx.x >> cv_shift;
  const int ppb = 256 >> cv_shift;
  const long m0 = (long)blockIdx.x * ppb * AA_ROWS + lane_px;
  float sc[N], sh[N];
#pragma unroll
  for (int e = 0; e < N; ++e) { sc[e] = scale[cv * N + e]; sh[e] = shift[cv * N + e]; }
  auto finish = [&](long m, const V& v, const V& r) {
    V o;
#pragma unroll
    for (int e = 0; e < N; ++e) {
      float f = (float)v[e] * sc[e] + sh[e];      // same expression as the generic kernel
      if (res) f += (float)r[e];
      if (relu) f = fmaxf(f, 0.f);
      o[e] = (T)f;
    }
    *(V*)(z + m * z_stride + cv * N) = o;
  };
  if (m0 + (long)(AA_ROWS - 1) * ppb < M) {
#pragma unroll
    for (int r0 = 0; r0 < AA_ROWS; r0 += 4) {
      V v[4], r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long m = m0 + (long)(r0 + u) * ppb;
        v[u] = *(const V*)(y + m * y_stride + cv * N);
        r[u] = v[u];
        if (res) r[u] = *(const V*)(res + m * res_stride + cv * N);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) finish(m0 + (long)(r0 + u) * ppb, v[u], r[u]);
    }
    return;
  }
  for (int r = 0; r < AA_ROWS; ++r) {
    const long m = m0 + (long)r * ppb;
    if (m >= M) break;
    const V v = *(const V*)(y + m * y_stride + cv * N);
    V rr = v;
    if (res) rr = *(const V*)(res + m * res_stride + cv * N);
    finish(m, v, rr);
  }
}

// BN backward apply: g = dz*(z>0); dy = scale*(g - s0/M - xhat*s1/M); optional dres = g.
// With y == null (bias+act layers): dy = g.
template <typename T>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const T* __restrict__ dz, int dz_stride,
                                                           const T* __restrict__ z, int z_stride,
                                                           const T* __restrict__ y, int y_stride,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ scale, const float* __restrict__ s0,
                                                           const float* __restrict__ s1, T* __restrict__ dy, int dy_stride,
                                                           T* __restrict__ dres, int dres_stride, long M, int C, int relu) {
  typedef typename VecT<T>::type V;
  constexpr int N = VecT<T>::N;
  const int CV = C / N;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * CV) return;
  const int cv = (int)(idx % CV);
  const long m = idx / CV;
  const V g = *(const V*)(dz + m * dz_stride + cv * N);
  V zz = g, yy = g;
  if (relu) zz = *(const V*)(z + m * z_stride + cv * N);
  if (y) yy = *(const V*)(y + m * y_stride + cv * N);
  const float invM = 1.f / (float)M;
  V o, gr;
#pragma unroll
  for (int e = 0; e < N; ++e) {
    const int c = cv * N + e;
    const float gf = (relu && !((float)zz[e] > 0.f)) ? 0.f : (float)g[e];
    gr[e] = (T)gf;
    if (y) {
      const float xh = ((float)yy[e] - mean[c]) * invstd[c];
      o[e] = (T)(scale[c] * (gf - s0[c] * invM - xh * s1[c] * invM));
    } else {
      o[e] = (T)gf;
    }
  }
  *(V*)(dy + m * dy_stride + cv * N) = o;
  if (dres) *(V*)(dres + m * dres_stride + cv * N) = gr;
}

// the same for channel-vector counts that are powers of two (every BatchNorm of DLA-34): a thread owns one channel group,
// keeps its per-channel coefficients in registers and walks BN_ROWS pixels -- 3 loads + 1-2 stores per 16 output bytes
// instead of 13 loads (10 of them the coefficients, re-fetched per pixel)
constexpr int BN_ROWS = 8;
template <typename T>
__global__ void __launch_bounds__(256) bn_bwd_apply_rows_kernel(const T* __restrict__ dz, int dz_stride,
                                                                const T* __restrict__ z, int z_stride,
                                                                const T* __restrict__ y, int y_stride,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ scale, const float* __restrict__ s0,
                                                                const float* __restrict__ s1, T* __restrict__ dy, int dy_stride,
                                                                T* __restrict__ dres, int dres_stride, long M, int cv_shift,
                                                                int relu) {
  typedef typename VecT<T>::type V;
  constexpr int N = VecT<T>::N;
  const int CV = 1 << cv_shift;
  const int cv = threadIdx.x & (CV - 1), lane_px = threadIdx.x >> cv_shift;
  const int ppb = 256 >> cv_shift;                      // pixels a workgroup covers per step
  const long m0 = (long)blockIdx.x * ppb * BN_ROWS + lane_px;
  const float invM = 1.f / (float)M;
  float mu[N], is[N], sc[N], a0[N], a1[N];
  if (y) {
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const int c = cv * N + e;
      mu[e] = mean[c]; is[e] = invstd[c]; sc[e] = scale[c]; a0[e] = s0[c] * invM; a1[e] = s1[c] * invM;
    }
  }
  auto finish = [&](long m, const V& g, const V& zz, const V& yy) {
    V o, gr;
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const float gf = (relu && !((float)zz[e] > 0.f)) ? 0.f : (float)g[e];
      gr[e] = (T)gf;
      if (y) {
        const float xh = ((float)yy[e] - mu[e]) * is[e];
        o[e] = (T)(sc[e] * (gf - a0[e] - xh * a1[e]));     // same expression as the generic kernel
      } else {
        o[e] = (T)gf;
      }
    }
    *(V*)(dy + m * dy_stride + cv * N) = o;
    if (dres) *(V*)(dres + m * dres_stride + cv * N) = gr;
  };
  if (m0 + (long)(BN_ROWS - 1) * ppb < M) {
    // whole range in bounds: four rows of the three tensors in flight before the first use (a loop with a bounds `break`
    // keeps one row's loads outstanding per thread and runs at latency x occupancy)
#pragma unroll
    for (int r0 = 0; r0 < BN_ROWS; r0 += 4) {
      V g[4], zz[4], yy[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long m = m0 + (long)(r0 + u) * ppb;
        g[u] = *(const V*)(dz + m * dz_stride + cv * N);
        zz[u] = g[u]; yy[u] = g[u];
        if (relu) zz[u] = *(const V*)(z + m * z_stride + cv * N);
        if (y) yy[u] = *(const V*)(y + m * y_stride + cv * N);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) finish(m0 + (long)(r0 + u) * ppb, g[u], zz[u], yy[u]);
    }
    return;
  }
  for (int r = 0; r < BN_ROWS; ++r) {
    const long m = m0 + (long)r * ppb;
    if (m >= M) break;
    const V g = *(const V*)(dz + m * dz_stride + cv * N);
    V zz = g, yy = g;
    if (relu) zz = *(const V*)(z + m * z_stride + cv * N);
    if (y) yy = *(const V*)(y + m * y_stride + cv * N);
    finish(m, g, zz, yy);
  }
}

// ------------------------------------------------------------------------------------------------
// conv weight gradient:  dW[n][k] += sum_m dY[m][n] * A[m][k],  A = im2col(x), k = (r*S+s)*Cin + c (tap-major).
// Workgroup tile: 64 couts x 128 k, looping over its share of the pixels 32 at a time.  Both MFMA operands need
// 8 consecutive *pixels* per lane for one channel, i.e. the transpose of the pixel-major NHWC rows: the LDS tiles
// stay row-major [pixel][channel] and ds_read_b64_tr_b16 delivers the 4x16 blocks column-major.
// Results are added to the f32 dW with atomics (split over pixel ranges across workgroups).
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
  const void* x; const void* dy; float* dw;   // x, dy: f16 (f16 mode) or f32 (f16x3 mode: split into hi + lo f16 halves on the way to LDS)
  int B, H, W, Cin, in_stride, Cout, Ho, Wo, dy_stride, R, S, stride, pad, dil, K, M, msplit;
  float scale;   // multiplier applied to every partial sum before it is added to dw
  int lw, lh;    // log2(Wo), log2(Ho) when both are powers of two, else -1
  // output layout: perm_rs == 0: dw[n][k], k = tap*Cin + c (tap-major).  perm_rs > 0: the parameter's own OIHW layout,
  // dw[(n*cin_real + c)*perm_rs + tap] with k = tap*perm_cin + c, channels c >= cin_real (input padding) dropped -- the
  // and rows n >= cout_real (output-channel padding of dy) dropped -- the kernel then accumulates straight into the
  // optimizer's gradient buffer
  int perm_rs, perm_cin, cin_real, cout_real;
};

template <typename A>
__device__ __forceinline__ void wg_add(const A& a, int n, int k, float v) {
  if (a.perm_rs == 0) { atomicAdd(a.dw + (long)n * a.K + k, v); return; }
  if (a.perm_rs == 1) {   // 1x1: the parameter's layout is the kernel's, minus the channel padding
    if (k < a.cin_real && n < a.cout_real) atomicAdd(a.dw + (long)n * a.cin_real + k, v);
    return;
  }
  const int tap = k / a.perm_cin, c = k - tap * a.perm_cin;
  if (c < a.cin_real && n < a.cout_real) atomicAdd(a.dw + ((long)n * a.cin_real + c) * a.perm_rs + tap, v);
}

__device__ __forceinline__ f16x4 lds_tr_read(const f16* p) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  return __builtin_bit_cast(f16x4, v);
}

// ---- operand pieces of the weight-gradient kernels ---------------------------------------------------------------
// A piece = 8 consecutive channels of one pixel as it comes from global memory: one f16x8 (f16 mode) or two f32x4 (the
// f16x3 mode, X3 = true: f32 tensors).  On the way to LDS an f32 piece is split into hi = f16(v) and lo = f16(v - hi) and
// stored into two tiles of the same shape (the lo tile `lo_off` elements behind the hi tile); the MFMAs then run
// dY_hi . X_hi + dY_lo . X_hi + dY_hi . X_lo with f32 accumulation (conv_common.h: the f16x3 arithmetic of the forward
// kernels; the dropped lo . lo term is 2^-22 of the product).
template <bool X3> struct Piece;
template <> struct Piece<false> { f16x8 v; };
template <> struct Piece<true> { f32x4 a, b; };

template <bool X3>
__device__ __forceinline__ Piece<X3> piece_load(const char* p) {
  Piece<X3> r;
  if constexpr (X3) { r.a = *(const f32x4*)p; r.b = *(const f32x4*)(p + 16); }
  else r.v = *(const f16x8*)p;
  return r;
}

template <bool X3>
__device__ __forceinline__ void piece_store(f16* dst, int lo_off, const Piece<X3>& r, bool ok) {
  const f16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr (X3) {
    f16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      hi[j] = (f16)r.a[j]; lo[j] = (f16)(r.a[j] - (float)hi[j]);
      hi[4 + j] = (f16)r.b[j]; lo[4 + j] = (f16)(r.b[j] - (float)hi[4 + j]);
    }
    *(f16x8*)dst = ok ? hi : z8;
    *(f16x8*)(dst + lo_off) = ok ? lo : z8;
  } else {
    *(f16x8*)dst = ok ? r.v : z8;
  }
}

// a transposed 16 x 32 MFMA operand (8 pixels per lane of one channel) from a row-major [pixel][channel] LDS tile
template <bool X3> struct Frag { f16x8 h; f16x8 l; };
template <bool X3>
__device__ __forceinline__ Frag<X3> frag_read(const f16* base, int pitch, int lo_off) {
  Frag<X3> f;
  {
    const f16x4 lo = lds_tr_read(base), hi = lds_tr_read(base + 4 * pitch);
    f.h = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
  if constexpr (X3) {
    const f16x4 lo = lds_tr_read(base + lo_off), hi = lds_tr_read(base + lo_off + 4 * pitch);
    f.l = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
  return f;
}
template <bool X3>
__device__ __forceinline__ f32x4 mma_frag(const Frag<X3>& y, const Frag<X3>& x, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(y.h, x.h, acc, 0, 0, 0);
  if constexpr (X3) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(y.l, x.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(y.h, x.l, acc, 0, 0, 0);
  }
  return acc;
}

#define WG_BN 64
#define WG_BK 128
#define WG_BM 64            // pixels per K-loop step (two MFMA K slabs of 32)
#define WG_LDA (WG_BK + 8)  // row padding (elements) to spread banks
#define WG_LDY (WG_BN + 8)
template <bool PERM, bool X3>
__global__ void __launch_bounds__(256) conv_wgrad_kernel(const WgradArgs a) {
  constexpr int NT = X3 ? 2 : 1, ES = X3 ? 4 : 2;          // LDS tiles per operand (hi, lo); bytes per element in memory
  constexpr int LOA = WG_BM * WG_LDA, LOY = WG_BM * WG_LDY;
  __shared__ __attribute__((aligned(16))) f16 sA[NT * LOA];
  __shared__ __attribute__((aligned(16))) f16 sY[NT * LOY];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // 1-D grid; the gx*gy workgroups of one pixel range get ids 8 apart = the same XCD / L2 (see conv_wgrad_win_kernel)
  const int gx = (a.K + WG_BK - 1) / WG_BK, gy = (a.Cout + WG_BN - 1) / WG_BN, nxy = gx * gy;
  int xy, bz;
  if ((a.msplit & 7) == 0) { const int slot = blockIdx.x >> 3; bz = (slot / nxy) * 8 + (blockIdx.x & 7); xy = slot % nxy; }
  else { xy = blockIdx.x % nxy; bz = blockIdx.x / nxy; }
  const int k0 = (xy % gx) * WG_BK, n0 = (xy / gx) * WG_BN;
  const int wn = wave >> 1, wk = wave & 1;  // wave tile: 32 couts x 64 k
  const int m_per = ((a.M + a.msplit - 1) / a.msplit + WG_BM - 1) / WG_BM * WG_BM;
  const int m_begin = bz * m_per;
  const int m_end = m_begin + m_per < a.M ? m_begin + m_per : a.M;

  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // loaders: A tile 64 rows x 128 k = 1024 8-channel pieces (4 per thread); dY tile 64 x 64 = 512 pieces (2 per thread)
  const int a_row = tid >> 4, a_kg = tid & 15;      // rows a_row + 16 i ; k-group a_kg (8 channels)
  const int y_row = tid >> 3, y_ng = tid & 7;       // rows y_row + 32 i
  const int kk = k0 + a_kg * 8;
  const bool k_ok = kk < a.K;
  const int tap = k_ok ? kk / a.Cin : 0;
  const int c0 = kk - tap * a.Cin;
  const int tr = tap / a.S, ts = tap - tr * a.S;
  const bool n_ok = n0 + y_ng * 8 < a.Cout;
  const char* const xb = (const char*)a.x;
  const char* const yb = (const char*)a.dy;

  const int grp = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  Piece<X3> av[4], yv[2];
  unsigned okbits = 0;
  // the 64-pixel slab `mb` of both operands into registers (unconditional loads from clamped addresses, zeroed when they
  // go to LDS, so the loads of the next slab stay in flight behind the MFMAs of the current one)
  auto fetch = [&](int mb) {
    okbits = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mb + a_row + 16 * i;
      const int mc = m < m_end ? m : m_end - 1;
      int wo, ho, b;
      if (a.lw >= 0) {          // power-of-two maps (every DLA-34 level at 512^2): shifts instead of four divisions per row
        wo = mc & (a.Wo - 1);
        ho = (mc >> a.lw) & (a.Ho - 1);
        b = mc >> (a.lw + a.lh);
      } else {
        wo = mc % a.Wo;
        const int t = mc / a.Wo;
        ho = t % a.Ho;
        b = t / a.Ho;
      }
      const int hi = ho * a.stride - a.pad + tr * a.dil, wi = wo * a.stride - a.pad + ts * a.dil;
      const bool ok = m < m_end && k_ok && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
      av[i] = piece_load<X3>(ok ? xb + (((long)(b * a.H + hi) * a.W + wi) * a.in_stride + c0) * ES : xb);
      okbits |= ok ? 1u << i : 0u;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = mb + y_row + 32 * i;
      const bool ok = m < m_end && n_ok;
      yv[i] = piece_load<X3>(ok ? yb + ((long)m * a.dy_stride + n0 + y_ng * 8) * ES : yb);
      okbits |= ok ? 16u << i : 0u;
    }
  };
  if (m_begin < m_end) fetch(m_begin);
  for (int mb = m_begin; mb < m_end; mb += WG_BM) {
    __syncthreads();  // previous iteration's fragment reads are done
#pragma unroll
    for (int i = 0; i < 4; ++i) piece_store<X3>(sA + (a_row + 16 * i) * WG_LDA + a_kg * 8, LOA, av[i], (okbits >> i) & 1u);
#pragma unroll
    for (int i = 0; i < 2; ++i) piece_store<X3>(sY + (y_row + 32 * i) * WG_LDY + y_ng * 8, LOY, yv[i], (okbits >> (4 + i)) & 1u);
    __syncthreads();
    if (mb + WG_BM < m_end) fetch(mb + WG_BM);
    // fragments: lane group grp covers pixels 8*grp .. 8*grp+7 of a 32-pixel K slab (two 4-row blocks); lane 4q+p
    // addresses row q, columns 4p..4p+3 of a 4x16 block and receives column li of its 4 rows
#pragma unroll
    for (int h = 0; h < WG_BM / 32; ++h) {
      Frag<X3> fy[2], fa[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) fy[i] = frag_read<X3>(sY + (32 * h + 8 * grp + q) * WG_LDY + wn * 32 + i * 16 + 4 * p, WG_LDY, LOY);
#pragma unroll
      for (int j = 0; j < 4; ++j) fa[j] = frag_read<X3>(sA + (32 * h + 8 * grp + q) * WG_LDA + wk * 64 + j * 16 + 4 * p, WG_LDA, LOA);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mma_frag<X3>(fy[i], fa[j], acc[i][j]);
    }
  }
  // D[row = cout][col = k]: lane holds rows 4*(lane>>4)+reg, col lane&15
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kcol = k0 + wk * 64 + j * 16 + (lane & 15);
      if (kcol >= a.K) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 32 + i * 16 + 4 * (lane >> 4) + r;
        if (n < a.Cout) {
          if constexpr (PERM) wg_add(a, n, kcol, acc[i][j][r] * a.scale);
          else atomicAdd(a.dw + (long)n * a.K + kcol, acc[i][j][r] * a.scale);
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// VoVNet pieces of the training step (round 4; the forward kernels are the eval path's, pointwise.hip).
// MaxPool2d(3, stride 2) backward, pad 1 (BasicStem) or pad 0 + ceil_mode (the VoVNet stage pooling, vovnet.py:291-292): windows
// overlap, so the kernel GATHERS -- one thread per input element vector, which looks at the (up to four) windows that contain
// it, re-derives each window's first maximum in (dy, dx) scan order (PyTorch's argmax rule: strict >) and takes that window's
// gradient if the maximum is this element.  No atomics, deterministic.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) maxpool3x3s2_bwd_kernel(const T* __restrict__ x, int x_stride, const T* __restrict__ dz,
                                                               int dz_stride, T* __restrict__ dx, int dx_stride, int B, int H, int W,
                                                               int C, int pad, int Ho, int Wo) {
  typedef typename VecT<T>::type V;
  constexpr int N = VecT<T>::N;
  const int CV = C / N;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * H * W * CV) return;
  const int cv = (int)(idx % CV);
  long t = idx / CV;
  const int xi = (int)(t % W); t /= W;
  const int yi = (int)(t % H);
  const int b = (int)(t / H);
  const V self = *(const V*)(x + ((long)(b * H + yi) * W + xi) * x_stride + cv * N);
  float g[N];
#pragma unroll
  for (int e = 0; e < N; ++e) g[e] = 0.f;
  // window (ho, wo) covers rows 2 ho - pad .. 2 ho - pad + 2
  const int ho_lo = (yi + pad - 2 + 1) >> 1 > 0 ? (yi + pad - 2 + 1) >> 1 : 0, ho_hi = (yi + pad) >> 1 < Ho - 1 ? (yi + pad) >> 1 : Ho - 1;
  const int wo_lo = (xi + pad - 2 + 1) >> 1 > 0 ? (xi + pad - 2 + 1) >> 1 : 0, wo_hi = (xi + pad) >> 1 < Wo - 1 ? (xi + pad) >> 1 : Wo - 1;
  for (int ho = ho_lo; ho <= ho_hi; ++ho)
    for (int wo = wo_lo; wo <= wo_hi; ++wo) {
      float m[N];
      int arg[N];
#pragma unroll
      for (int e = 0; e < N; ++e) { m[e] = -INFINITY; arg[e] = -1; }
      for (int dy = 0; dy < 3; ++dy) {
        const int yy = 2 * ho - pad + dy;
        if (yy < 0 || yy >= H) continue;
        for (int dxx = 0; dxx < 3; ++dxx) {
          const int xx = 2 * wo - pad + dxx;
          if (xx < 0 || xx >= W) continue;
          const V v = *(const V*)(x + ((long)(b * H + yy) * W + xx) * x_stride + cv * N);
#pragma unroll
          for (int e = 0; e < N; ++e)
            if ((float)v[e] > m[e] || arg[e] < 0) { m[e] = (float)v[e]; arg[e] = yy * W + xx; }
        }
      }
      const V gz = *(const V*)(dz + ((long)(b * Ho + ho) * Wo + wo) * dz_stride + cv * N);
#pragma unroll
      for (int e = 0; e < N; ++e)
        if (arg[e] == yi * W + xi) g[e] += (float)gz[e];
    }
  (void)self;
  V o;
#pragma unroll
  for (int e = 0; e < N; ++e) o[e] = (T)g[e];
  *(V*)(dx + ((long)(b * H + yi) * W + xi) * dx_stride + cv * N) = o;
}

template <typename T>
static int launch_maxpool3x3s2_bwd_t(const void* x, int xs, const void* dz, int dzs, void* dx, int dxs, int B, int H, int W, int C,
                                     int pad, int Ho, int Wo, hipStream_t s) {
  constexpr int N = VecT<T>::N;
  CTDET_CHECK(C % N == 0 && xs % N == 0 && dzs % N == 0 && dxs % N == 0, "maxpool3x3s2_bwd: channels / strides must be multiples of %d", N);
  const long total = (long)B * H * W * (C / N);
  if (total == 0) return 0;
  hipLaunchKernelGGL((maxpool3x3s2_bwd_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const T*)x, xs, (const T*)dz,
                     dzs, (T*)dx, dxs, B, H, W, C, pad, Ho, Wo);
  CTDET_LAUNCH_CHECK();
  return 0;
}
int launch_maxpool3x3s2_bwd(const void* x, int xs, const void* dz, int dzs, void* dx, int dxs, int dtype, int B, int H, int W, int C,
                            int pad, int Ho, int Wo, hipStream_t s) {
  if (dtype == CTDET_F16) return launch_maxpool3x3s2_bwd_t<f16>(x, xs, dz, dzs, dx, dxs, B, H, W, C, pad, Ho, Wo, s);
  if (dtype == CTDET_F32) return launch_maxpool3x3s2_bwd_t<float>(x, xs, dz, dzs, dx, dxs, B, H, W, C, pad, Ho, Wo, s);
  CTDET_CHECK(false, "maxpool3x3s2_bwd: bad dtype %d", dtype);
  return 0;
}

// eSE attention backward (vovnet.py:200-213; forward: y = x * hsigmoid(s[b][c]) (+ identity), s = fc(mean over pixels of x)):
//   ese_dot_kernel    r[b][c] = sum over the pixels of dy * x          (the gradient reaching hsigmoid(s), before its slope)
//   ese_bwd_kernel    dx = dy * g[b][c] + gp[b][c]                      (g = hsigmoid(s); gp = d(mean) / HW, the pooled path)
// The B x C numbers in between (hsigmoid's slope, the fc layer's three gradients) are a handful of tiny device-side torch ops.
template <typename T>
__global__ void __launch_bounds__(256) ese_dot_kernel(const T* __restrict__ dy, int dy_stride, const T* __restrict__ x, int x_stride,
                                                      int HW, int C, float* __restrict__ out) {
  __shared__ float red[256];
  const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6;
  float acc = 0.f;
  if (c < C)
    for (int p = pl; p < HW; p += 4) acc += (float)dy[((long)b * HW + p) * dy_stride + c] * (float)x[((long)b * HW + p) * x_stride + c];
  red[threadIdx.x] = acc;
  __syncthreads();
  if (pl == 0 && c < C)
    out[(long)b * C + c] = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
}
template <typename T>
__global__ void __launch_bounds__(256) ese_bwd_kernel(const T* __restrict__ dy, int dy_stride, const float* __restrict__ g,
                                                      const float* __restrict__ gp, T* __restrict__ dx, int dx_stride, int B, int HW,
                                                      int C) {
  typedef typename VecT<T>::type V;
  constexpr int N = VecT<T>::N;
  const int CV = C / N;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * HW * CV) return;
  const int cv = (int)(idx % CV);
  const long m = idx / CV;
  const int b = (int)(m / HW);
  const V v = *(const V*)(dy + m * dy_stride + cv * N);
  V r;
#pragma unroll
  for (int e = 0; e < N; ++e) r[e] = (T)((float)v[e] * g[(long)b * C + cv * N + e] + gp[(long)b * C + cv * N + e]);
  *(V*)(dx + m * dx_stride + cv * N) = r;
}
int launch_ese_dot(const void* dy, int dys, const void* x, int xs, int dtype, int B, int HW, int C, float* out, hipStream_t s) {
  if ((long)B * HW * C == 0) return 0;
  const dim3 grid((C + 63) / 64, B);
  if (dtype == CTDET_F16) hipLaunchKernelGGL((ese_dot_kernel<f16>), grid, dim3(256), 0, s, (const f16*)dy, dys, (const f16*)x, xs, HW, C, out);
  else if (dtype == CTDET_F32) hipLaunchKernelGGL((ese_dot_kernel<float>), grid, dim3(256), 0, s, (const float*)dy, dys, (const float*)x, xs, HW, C, out);
  else CTDET_CHECK(false, "ese_dot: bad dtype %d", dtype);
  CTDET_LAUNCH_CHECK();
  return 0;
}
int launch_ese_bwd(const void* dy, int dys, const float* g, const float* gp, void* dx, int dxs, int dtype, int B, int HW, int C,
                   hipStream_t s) {
  const int N = dtype == CTDET_F16 ? 8 : 4;
  CTDET_CHECK(C % N == 0 && dys % N == 0 && dxs % N == 0, "ese_bwd: channels / strides must be multiples of %d", N);
  const long total = (long)B * HW * (C / N);
  if (total == 0) return 0;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == CTDET_F16) hipLaunchKernelGGL((ese_bwd_kernel<f16>), grid, dim3(256), 0, s, (const f16*)dy, dys, g, gp, (f16*)dx, dxs, B, HW, C);
  else if (dtype == CTDET_F32) hipLaunchKernelGGL((ese_bwd_kernel<float>), grid, dim3(256), 0, s, (const float*)dy, dys, g, gp, (float*)dx, dxs, B, HW, C);
  else CTDET_CHECK(false, "ese_bwd: bad dtype %d", dtype);
  CTDET_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// MaxPool2d(2,2) backward: the gradient goes to the first maximum in (dy,dx) scan order (PyTorch's argmax rule)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) maxpool2x2_bwd_kernel(const T* __restrict__ x, int x_stride,
                                                             const T* __restrict__ dz, int dz_stride,
                                                             T* __restrict__ dx, int dx_stride, int B, int H, int W, int C) {
  typedef typename VecT<T>::type V;
  constexpr int N = VecT<T>::N;
  const int Ho = H / 2, Wo = W / 2, CV = C / N;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * Ho * Wo * CV) return;
  const int cv = (int)(idx % CV);
  long t = idx / CV;
  const int wo = (int)(t % Wo); t /= Wo;
  const int ho = (int)(t % Ho);
  const int b = (int)(t / Ho);
  const long p00 = (long)(b * H + 2 * ho) * W + 2 * wo;
  const long offs[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
  V v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = *(const V*)(x + offs[i] * x_stride + cv * N);
  const V g = *(const V*)(dz + ((long)(b * Ho + ho) * Wo + wo) * dz_stride + cv * N);
  V o[4];
#pragma unroll
  for (int e = 0; e < N; ++e) {
    int best = 0;
    float bv = (float)v[0][e];
#pragma unroll
    for (int i = 1; i < 4; ++i)
      if ((float)v[i][e] > bv) { bv = (float)v[i][e]; best = i; }
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i][e] = i == best ? g[e] : (T)0.f;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) *(V*)(dx + offs[i] * dx_stride + cv * N) = o[i];
}

// ------------------------------------------------------------------------------------------------
// depthwise ConvTranspose2d(k=2f, s=f, p=f/2) backward.
//   dx[b,iy,ix,c]   = sum_{ky,kx} dz[b, iy*f - p + ky, ix*f - p + kx, c] * w[ky][kx][c]
//   dw[ky][kx][c]  += sum_{b,iy,ix} x[b,iy,ix,c] * dz[b, iy*f - p + ky, ix*f - p + kx, c]   (f32 atomics)
// ------------------------------------------------------------------------------------------------
// dw: one tap phase (ky1, kx1) = ((oy+p) % f, (ox+p) % f) per blockIdx.y.  All output pixels of a phase feed the same four
// taps (ky1 + {0,f}, kx1 + {0,f}), so a thread (fixed 8-channel group) keeps its 4 x 8 partial sums in registers while it
// walks the phase's pixels; partials are combined across the workgroup through LDS with plain stores (LDS float atomics
// measured ~170 cycles per wave-instruction here) and leave as 4*C global atomics per workgroup.
template <typename T>
__global__ void __launch_bounds__(256) dwconvT_dw_kernel(const T* __restrict__ x, int x_stride, const T* __restrict__ dz,
                                                         int dz_stride, float* __restrict__ dw, int B, int H, int W, int C,
                                                         int f) {
  typedef typename VecT<T>::type V;
  constexpr int N = VecT<T>::N;
  extern __shared__ float part[];  // [S][4][C]
  const int CV = C / N, S = 256 / CV, k = 2 * f, p = f / 2, Ho = H * f, Wo = W * f;
  const int cv = threadIdx.x % CV, sub = threadIdx.x / CV;
  const int ky1 = blockIdx.y / f, kx1 = blockIdx.y % f;
  float acc[4][N];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int e = 0; e < N; ++e) acc[t][e] = 0.f;
  const long npos = (long)B * (H + 1) * (W + 1);
  if (sub < S) {
    for (long pos = (long)blockIdx.x * S + sub; pos < npos; pos += (long)gridDim.x * S) {
      const int ix1 = (int)(pos % (W + 1));
      const long t2 = pos / (W + 1);
      const int iy1 = (int)(t2 % (H + 1)), b = (int)(t2 / (H + 1));
      const int oy = iy1 * f + ky1 - p, ox = ix1 * f + kx1 - p;
      if (oy < 0 || oy >= Ho || ox < 0 || ox >= Wo) continue;
      const V g = *(const V*)(dz + ((long)(b * Ho + oy) * Wo + ox) * dz_stride + cv * N);
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        const int iy = iy1 - dy;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int dxx = 0; dxx < 2; ++dxx) {
          const int ix = ix1 - dxx;
          if (ix < 0 || ix >= W) continue;
          const V xv = *(const V*)(x + ((long)(b * H + iy) * W + ix) * x_stride + cv * N);
#pragma unroll
          for (int e = 0; e < N; ++e) acc[dy * 2 + dxx][e] += (float)xv[e] * (float)g[e];
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < N; ++e) part[(sub * 4 + t) * C + cv * N + e] = acc[t][e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * C; i += 256) {
    float sum = 0.f;
    for (int s2 = 0; s2 < S; ++s2) sum += part[s2 * 4 * C + i];
    const int t = i / C, c = i - t * C;
    const int ky = ky1 + (t >> 1) * f, kx = kx1 + (t & 1) * f;
    if (sum != 0.f) atomicAdd(dw + (long)(ky * k + kx) * C + c, sum);
  }
}

// dx: gather the k x k output window of every input pixel
template <typename T>
__global__ void __launch_bounds__(256) dwconvT_dx_kernel(const T* __restrict__ dz, int dz_stride, const float* __restrict__ w,
                                                         T* __restrict__ dx, int dx_stride, int B, int H, int W, int C, int f) {
  typedef typename VecT<T>::type V;
  constexpr int N = VecT<T>::N;
  const int CV = C / N, k = 2 * f, p = f / 2, Ho = H * f, Wo = W * f;
  const long nin = (long)B * H * W * CV;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < nin; idx += (long)gridDim.x * 256) {
    const int cv = (int)(idx % CV);
    const long pix = idx / CV;
    const int ix = (int)(pix % W);
    const long t = pix / W;
    const int iy = (int)(t % H), b = (int)(t / H);
    float acc[N];
#pragma unroll
    for (int e = 0; e < N; ++e) acc[e] = 0.f;
    for (int ky = 0; ky < k; ++ky) {
      const int oy = iy * f - p + ky;
      if (oy < 0 || oy >= Ho) continue;
      for (int kx = 0; kx < k; ++kx) {
        const int ox = ix * f - p + kx;
        if (ox < 0 || ox >= Wo) continue;
        const V g = *(const V*)(dz + ((long)(b * Ho + oy) * Wo + ox) * dz_stride + cv * N);
        const float* wt = w + (long)(ky * k + kx) * C + cv * N;
#pragma unroll
        for (int e = 0; e < N; ++e) acc[e] += (float)g[e] * wt[e];
      }
    }
    V o;
#pragma unroll
    for (int e = 0; e < N; ++e) o[e] = (T)acc[e];
    *(V*)(dx + pix * dx_stride + cv * N) = o;
  }
}

// ------------------------------------------------------------------------------------------------
// DCNv2 training pieces (3x3, stride 1, pad 1, dil 1, one deformable group)
// ------------------------------------------------------------------------------------------------
struct DcnGeom {
  float w[4];     // bilinear weights hh*hw, hh*lw, lh*hw, lh*lw
  long off[4];    // corner element offsets or -1
  float hh, hw, lh, lw, mask;
  bool inside;
};
__device__ __forceinline__ DcnGeom dcn_geom(const float* om, int tap, int b, int ho, int wo, int H, int W, int stride_elems,
                                            int mask_is_prob) {
  DcnGeom g;
  const int tr = tap / 3, ts = tap - tr * 3;
  const float h_im = (float)(ho - 1 + tr) + om[2 * tap], w_im = (float)(wo - 1 + ts) + om[2 * tap + 1];
  g.mask = mask_is_prob ? om[18 + tap] : ctdet_sigmoid_exact(om[18 + tap]);   // prob: the reference's functional form
  g.inside = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
#pragma unroll
  for (int q = 0; q < 4; ++q) { g.off[q] = -1; g.w[q] = 0.f; }
  g.hh = g.hw = g.lh = g.lw = 0.f;
  if (!g.inside) return g;
  const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
  const int h_high = h_low + 1, w_high = w_low + 1;
  g.lh = h_im - (float)h_low; g.lw = w_im - (float)w_low; g.hh = 1.f - g.lh; g.hw = 1.f - g.lw;
  g.w[0] = g.hh * g.hw; g.w[1] = g.hh * g.lw; g.w[2] = g.lh * g.hw; g.w[3] = g.lh * g.lw;
  const long base = (long)b * H * W;
  if (h_low >= 0 && w_low >= 0) g.off[0] = (base + (long)h_low * W + w_low) * stride_elems;
  if (h_low >= 0 && w_high <= W - 1) g.off[1] = (base + (long)h_low * W + w_high) * stride_elems;
  if (h_high <= H - 1 && w_low >= 0) g.off[2] = (base + (long)h_high * W + w_low) * stride_elems;
  if (h_high <= H - 1 && w_high <= W - 1) g.off[3] = (base + (long)h_high * W + w_high) * stride_elems;
  return g;
}

// d(offset), d(mask) of one (pixel, tap) into the dom row (f32 or f16, row stride dom_stride); tap 0 also clears the padding
// channels 27 .. dom_stride-1, so the caller does not have to zero the buffer
__device__ __forceinline__ void dom_store(void* dom, int dom_f16, long m, int dom_stride, int tap, float v_h, float v_w, float v_m) {
  if (dom_f16) {
    f16* d = (f16*)dom + m * dom_stride;
    d[2 * tap] = (f16)v_h; d[2 * tap + 1] = (f16)v_w; d[18 + tap] = (f16)v_m;
    if (tap == 0) for (int c = 27; c < dom_stride; ++c) d[c] = (f16)0.f;
  } else {
    float* d = (float*)dom + m * dom_stride;
    d[2 * tap] = v_h; d[2 * tap + 1] = v_w; d[18 + tap] = v_m;
    if (tap == 0) for (int c = 27; c < dom_stride; ++c) d[c] = 0.f;
  }
}

// col[m][tap*Cin + c] = mask * bilinear(x)   (the `columns` of the reference, f16, tap-major)
__global__ void __launch_bounds__(256) dcn_cols_kernel(const f16* __restrict__ x, int x_stride, const float* __restrict__ om,
                                                       int om_stride, f16* __restrict__ col, int B, int H, int W, int Cin, int mask_is_prob) {
  const int CV = Cin >> 3;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)B * H * W * 9 * CV;
  if (idx >= total) return;
  const int cv = (int)(idx % CV);
  long t = idx / CV;
  const int tap = (int)(t % 9);
  const long m = t / 9;
  const int wo = (int)(m % W);
  const long t2 = m / W;
  const int ho = (int)(t2 % H), b = (int)(t2 / H);
  const DcnGeom g = dcn_geom(om + m * om_stride, tap, b, ho, wo, H, W, x_stride, mask_is_prob);
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (g.off[q] >= 0) {
      const f16x8 v = *(const f16x8*)(x + g.off[q] + cv * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += g.w[q] * (float)v[e];
    }
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (f16)(acc[e] * g.mask);
  *(f16x8*)(col + m * (9L * Cin) + (long)tap * Cin + cv * 8) = o;
}

// given dcol (= W^T dY, [M][9*Cin] f16): dx (f32 atomics), d(offset), d(mask logit) -> dom [M][om_stride] f32.
// One wave per (pixel, tap) pass over the channels, one lane per channel: every atomic wave-instruction adds 64
// consecutive floats (256 contiguous bytes, the shape the memory-side atomic units run at full rate with); the
// three per-(pixel,tap) dot products are wave reductions.
__global__ void __launch_bounds__(256) dcn_col2im_coord_kernel(const f16* __restrict__ dcol, const f16* __restrict__ x,
                                                               int x_stride, const float* __restrict__ om, int om_stride,
                                                               float* __restrict__ dx, void* __restrict__ dom, int dom_stride,
                                                               int dom_f16, int B, int H, int W, int Cin, int mask_is_prob,
                                                               int chunked) {
  const int lane = threadIdx.x & 63;
  const long wave_id = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwork = (long)B * H * W * 9;
  for (long wk = wave_id; wk < nwork; wk += (long)gridDim.x * 4) {
    const int tap = (int)(wk % 9);
    const long m = wk / 9;
    const int wo = (int)(m % W);
    const long t2 = m / W;
    const int ho = (int)(t2 % H), b = (int)(t2 / H);
    const DcnGeom g = dcn_geom(om + m * om_stride, tap, b, ho, wo, H, W, x_stride, mask_is_prob);  // wave-uniform
    float val_dot = 0.f, dh = 0.f, dwv = 0.f;
    if (g.inside) {
      // dcol row of a pixel: [tap][Cin], or chunked [Cin/32][tap][32]
      const f16* dcp = dcol + m * (9L * Cin) + (chunked ? tap * 32 : tap * Cin);
      for (int c = lane; c < Cin; c += 64) {
        const float d = (float)dcp[chunked ? (c >> 5) * 288 + (c & 31) : c];
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = g.off[q] >= 0 ? (float)x[g.off[q] + c] : 0.f;
        val_dot += d * (g.w[0] * v[0] + g.w[1] * v[1] + g.w[2] * v[2] + g.w[3] * v[3]);
        dh += d * (-g.hw * v[0] - g.lw * v[1] + g.hw * v[2] + g.lw * v[3]);   // d val / d h  (kernel.cu:754-766)
        dwv += d * (-g.hh * v[0] + g.hh * v[1] - g.lh * v[2] + g.lh * v[3]);  // d val / d w  (kernel.cu:767-779)
        const float dm = d * g.mask;
#pragma unroll
        for (int q = 0; q < 4; ++q)  // input gradient scatter (kernel.cu:871-949)
          if (g.off[q] >= 0) atomicAdd(dx + g.off[q] / x_stride * (long)Cin + c, g.w[q] * dm);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      val_dot += __shfl_down(val_dot, o, 64);
      dh += __shfl_down(dh, o, 64);
      dwv += __shfl_down(dwv, o, 64);
    }
    if (lane == 0)   // the mask logit's gradient goes through its sigmoid
      dom_store(dom, dom_f16, m, dom_stride, tap, dh * g.mask, dwv * g.mask, mask_is_prob ? val_dot : val_dot * g.mask * (1.f - g.mask));
  }
}

// LDS-window form of the kernel above for maps divisible by 8x16 tiles and Cin % 32 == 0.  The scattered f32
// atomics into dx were the whole cost (36*Cin atomic adds per pixel: 2.4 GB of atomic traffic for one 64-channel
// 128x128 layer at batch 16, the chip's atomic rate is ~1.3 TB/s).  Here a workgroup owns an 8x16 pixel tile and, per
// 32-channel chunk, accumulates d(input) in an LDS window (18x26 pixels x 32 ch: the tile, +-1 for the taps, +-4
// for the offsets), then flushes the window with one global f32 atomic per touched element (~10x fewer, each
// wave-instruction 256 contiguous bytes).
// The LDS accumulation is FIXED POINT (ds_add_u32): ds_add_f32 measured ~170 cycles per wave-instruction here (3.2 ms
// for the layer above, slower than the global atomics it replaced), the integer form 0.48 ms.  Scale per (tile, chunk):
// 2^k with max|dcol| * 2^k <= 2^19; at most 128*9 contributions (|weight*mask| <= 1) can meet in one element, so
// the int32 sum cannot overflow, and the quantum is 2^-19 of the tile's largest dcol magnitude (the inputs are f16,
// the result is rounded to f16).  Integer accumulation also makes the in-tile sum order independent.
// The x window (f16) needed by d(offset)/d(mask) is staged in LDS too; lane = (pixel, 8-channel group); the four
// corner dot products use v_dot2_f32_f16.  Samples whose corners leave the window take the global path (per lane).
struct ColGeo {        // staged per (pixel, tap): 32 bytes
  unsigned off;        // bit 30: sample inside the image; bit 31: leaves the window.  Low 30 bits: window pixel index of
                       // corner (h_low, w_low), or (bit 31) its image pixel index as 30-bit two's complement
  unsigned valid;      // bit q: corner q inside the image
  float hh, hw, lh, lw, mask;
  float pad;
};

#ifdef CTDET_DEBUG_COL2IM
__device__ float* g_col2im_dbg = nullptr;
extern "C" int ctdet_debug_col2im_buffer(float* p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_col2im_dbg), &p, sizeof(p));
}
#endif
// 8 consecutive channels as floats (global or LDS)
template <typename T>
__device__ __forceinline__ void load8f(const T* p, float (&o)[8]) {
  if constexpr (sizeof(T) == 2) {
    const f16x8 v = *(const f16x8*)p;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
  } else {
    const f32x4 v0 = *(const f32x4*)p, v1 = *(const f32x4*)(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = v0[e]; o[4 + e] = v1[e]; }
  }
}

// NT = taps per workgroup: 9, or 3 with gridDim.y = 3 (one kernel row each) when the map has fewer tiles than the chip has
// CUs -- every tap's scatter and its d(offset) / d(mask) entries are independent of the other taps'.
// T = f16 (dcol, x f16: the f16 mode) or float (f32 dcol / x: the f16x3 training mode, fixed-point quantum 2^-20 of the tile's
// largest dcol magnitude instead of 2^-19).  157 KB of LDS in both: one workgroup per CU.
// FUSED (T = float, the f16x3 training mode): d(columns) = dY . W is not read from memory but computed here, per tile and
// 32-channel chunk, on the matrix pipe: D[row = column (tap, channel)][col = pixel] = W^T (288 x Cout) . dY^T (Cout x 16 pixels
// of the wave's tile row), f16x3 products (FusedDcol: the packed operand of ctdet_pack_weights_x3 layout 5, dY split in
// registers).  The accumulator layout hands lane (pixel, g) channels 4g..4g+3 and 16+4g..16+4g+3 of every tap -- the lane's
// eight channels of the scatter are those instead of 8g..8g+7 -- so the 302 MB (64 channels, batch 16; 604 MB in f32)
// d(columns) tensor of a layer is never written or read, and the 1x1 GEMM launch that produced it is gone.
struct FusedDcol {
  const float* dy; int dy_stride; int K;     // dY [M][dy_stride] f32, K = its channels used (multiple of 32, zero padded)
  const f16* wpk; const float* wscale;       // rows (Cin/32)*288 + tap*32 + c%32: K/8 groups of {hi[8], lo[8]}; per-row scale
};

template <int NT, typename T, bool FUSED = false>
__global__ void __launch_bounds__(512) dcn_col2im_window_kernel(const T* __restrict__ dcol, const T* __restrict__ x,
                                                                int x_stride, const float* __restrict__ om, int om_stride,
                                                                float* __restrict__ dx, void* __restrict__ dom, int dom_stride,
                                                                int dom_f16, int B, int H, int W, int Cin, int mask_is_prob,
                                                                int chunked, const FusedDcol fz) {
  static_assert(!FUSED || sizeof(T) == 4, "the fused d(columns) GEMM is the f16x3 mode's");
  constexpr int TH = 8, TW = 16, MG = 4, WR = TH + 2 + 2 * MG, WC = TW + 2 + 2 * MG, NPX = WR * WC;  // 18 x 26 = 468
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int FXB = F32 ? 20 : 19;      // fixed-point bits: 9 * 128 contributions * 2^20 < 2^31
  // d(input) window, int32 fixed point, laid out [e = channel % 8][window pixel][q = channel / 8]: the 64 lanes (16
  // pixels x 4 groups) of one ds_add then touch 64 consecutive words
  __shared__ __attribute__((aligned(16))) int dxw[NPX * 32];        // 59,904 B
  __shared__ __attribute__((aligned(16))) float xw[NPX * 32];       // 59,904 B: the x window as f32 in both modes (see the staging loop)
  __shared__ __attribute__((aligned(16))) ColGeo geo[9 * TH * TW];  // 36,864 B
  __shared__ float wmax[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = W / TW, tiles_y = H / TH;
  const int tx0 = (blockIdx.x % tiles_x) * TW, ty0 = ((blockIdx.x / tiles_x) % tiles_y) * TH;
  const int b = blockIdx.x / (tiles_x * tiles_y);
  const int wy0 = ty0 - 1 - MG, wx0 = tx0 - 1 - MG;
  const T* ximg = x + (long)b * H * W * x_stride;
  float* dximg = dx + (long)b * H * W * Cin;
  const int t0 = blockIdx.y * NT;   // first tap of this workgroup

  // ---- geometry once per (pixel, tap) ----
  for (int i = tid; i < NT * TH * TW; i += 512) {
    const int tl = i / (TH * TW), tap = t0 + tl, pl = i % (TH * TW);
    if (tap >= 9) break;               // NT = 5: the second workgroup owns taps 5..8
    const int py = ty0 + (pl >> 4), pxx = tx0 + (pl & 15);
    const float* omr = om + ((long)(b * H + py) * W + pxx) * om_stride;
    const int tr = tap / 3, ts = tap - tr * 3;
    const float h_im = (float)(py - 1 + tr) + omr[2 * tap], w_im = (float)(pxx - 1 + ts) + omr[2 * tap + 1];
    ColGeo g;
    g.mask = mask_is_prob ? omr[18 + tap] : ctdet_sigmoid_exact(omr[18 + tap]);
    g.off = 0; g.valid = 0; g.hh = g.hw = g.lh = g.lw = 0.f; g.pad = 0.f;
    if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
      const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
      g.lh = h_im - (float)h_low; g.lw = w_im - (float)w_low; g.hh = 1.f - g.lh; g.hw = 1.f - g.lw;
      const bool r0 = h_low >= 0, r1 = h_low + 1 <= H - 1, c0 = w_low >= 0, c1 = w_low + 1 <= W - 1;
      g.valid = (r0 && c0 ? 1u : 0u) | (r0 && c1 ? 2u : 0u) | (r1 && c0 ? 4u : 0u) | (r1 && c1 ? 8u : 0u);
      const int wr = h_low - wy0, wc = w_low - wx0;
      const bool inwin = wr >= 0 && wr + 1 < WR && wc >= 0 && wc + 1 < WC;
      g.off = inwin ? (unsigned)(wr * WC + wc) | 0x40000000u : ((unsigned)(h_low * W + w_low) & 0x3FFFFFFFu) | 0xC0000000u;
    }
    geo[tl * (TH * TW) + pl] = g;
  }

  const int fr = lane & 15, q = lane >> 4;
  const int prow = wave, pcol = fr;   // 8 waves = 8 tile rows; lane = (column, 8-channel group)
  const int pl = prow * 16 + pcol;
  const long m = ((long)(b * H + ty0 + prow) * W + tx0 + pcol);
  float s_val[NT], s_dh[NT], s_dw[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) s_val[t] = s_dh[t] = s_dw[t] = 0.f;

  const int nch = Cin / 32;
  for (int chunk = 0; chunk < nch; ++chunk) {
    // ---- this lane's dcol vectors of the chunk, and the tile's largest magnitude (fixed-point scale) ----
    float dv[NT][8];
    float amax = 0.f;
    if constexpr (FUSED) {
      f32x4 accd[2 * NT];
#pragma unroll
      for (int j = 0; j < 2 * NT; ++j) accd[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* dyp = fz.dy + m * fz.dy_stride + q * 8;          // this lane's pixel, k group q of a 32-deep step
      // W fragments come straight from memory (L2): groups of GJ tiles are fetched together, so that a group's 2 GJ loads are
      // in flight at once instead of one tile's pair at a time in front of its three MFMAs
      constexpr int GJ = 6;
      const f16* const wrow = fz.wpk + ((long)chunk * 288 + fr) * fz.K * 2 + q * 16;
      for (int ks = 0; ks < fz.K / 32; ++ks) {
        const f32x4 y0 = *(const f32x4*)(dyp + ks * 32), y1 = *(const f32x4*)(dyp + ks * 32 + 4);
        f16x8 bh, bl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          bh[e] = (f16)y0[e]; bl[e] = (f16)(y0[e] - (float)bh[e]);
          bh[4 + e] = (f16)y1[e]; bl[4 + e] = (f16)(y1[e] - (float)bh[4 + e]);
        }
#pragma unroll
        for (int j0 = 0; j0 < 2 * NT; j0 += GJ) {
          f16x8 ah[GJ], al[GJ];
#pragma unroll
          for (int u = 0; u < GJ; ++u) {
            const int j = j0 + u;
            if (j >= 2 * NT || t0 + (j >> 1) >= 9) continue;
            const f16* wp = wrow + (long)((t0 + (j >> 1)) * 32 + (j & 1) * 16) * fz.K * 2 + ks * 64;
            ah[u] = *(const f16x8*)wp; al[u] = *(const f16x8*)(wp + 8);
          }
#pragma unroll
          for (int u = 0; u < GJ; ++u) {
            const int j = j0 + u;
            if (j >= 2 * NT || t0 + (j >> 1) >= 9) continue;
            accd[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u], bh, accd[j], 0, 0, 0);
            accd[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[u], bh, accd[j], 0, 0, 0);
            accd[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u], bl, accd[j], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 2 * NT; ++j) {
        if (t0 + (j >> 1) >= 9) {
#pragma unroll
          for (int e = 0; e < 4; ++e) dv[j >> 1][(j & 1) * 4 + e] = 0.f;
          continue;
        }
        // the lane's four rows of tile j: columns (tap, 16 * (j & 1) + 4 q + r); the pack scaled every row by a power of two
        const f32x4 sc = *(const f32x4*)(fz.wscale + (long)chunk * 288 + (t0 + (j >> 1)) * 32 + (j & 1) * 16 + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = accd[j][e] * sc[e];
          dv[j >> 1][(j & 1) * 4 + e] = v;
          amax = fmaxf(amax, fabsf(v));
        }
      }
    } else {
      // chunked dcol ([Cin/32][tap][32] per pixel): the nine 32-channel pieces of this chunk are one contiguous run
      const T* dcp = dcol + m * (9L * Cin) + (chunked ? chunk * 288 : chunk * 32) + q * 8;
      const int tstep = chunked ? 32 : Cin;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t0 + t >= 9) {
#pragma unroll
          for (int e = 0; e < 8; ++e) dv[t][e] = 0.f;
          continue;
        }
        load8f<T>(dcp + (t0 + t) * tstep, dv[t]);
#pragma unroll
        for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(dv[t][e]));
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    __syncthreads();   // geometry staged (first chunk) / previous chunk's flush finished, wmax free
    if (lane == 0) wmax[wave] = amax;
    // ---- stage the x window and clear the dx window ----
    // f16 data is widened to f32 here, once per window element: the tap loop below is then the same instruction stream in
    // both modes.  (Round 4: with an f16 window and the conversions inside the tap loop -- v_dot2_f32_f16 in round 3, then
    // v_cvt + FMA chains -- the d(offset) / d(mask) sums of whole waves changed from run to run whenever a second process
    // shared the GPU, and never otherwise; the f32 instantiation never did: tools/probe_contention2.py)
    for (int i = tid; i < NPX * 4; i += 512) {
      const int pw = i >> 2, sl = i & 3;
      const int y = wy0 + pw / WC, xx = wx0 + pw % WC;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
      if (y >= 0 && y < H && xx >= 0 && xx < W) load8f<T>(ximg + ((long)y * W + xx) * x_stride + chunk * 32 + sl * 8, v);
      *(f32x4*)(xw + pw * 32 + sl * 8) = (f32x4){v[0], v[1], v[2], v[3]};
      *(f32x4*)(xw + pw * 32 + sl * 8 + 4) = (f32x4){v[4], v[5], v[6], v[7]};
    }
    for (int i = tid; i < NPX * 8; i += 512) *(int4*)(dxw + i * 4) = make_int4(0, 0, 0, 0);
    __syncthreads();
    float tmax = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) tmax = fmaxf(tmax, wmax[i]);
    int ex = 0;
    (void)frexpf(tmax, &ex);                       // tmax = f * 2^ex, f in [0.5, 1)  =>  tmax * 2^(FXB-ex) < 2^FXB
    const float fscale = ldexpf(1.f, FXB - ex), finv = ldexpf(1.f, ex - FXB);
    // ---- per tap: dots for d(offset)/d(mask), scatter of d(input) ----
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t0 + t >= 9) continue;
      const ColGeo g = geo[t * (TH * TW) + pl];
      if (!(g.off & 0x40000000u)) continue;            // sample outside the image: no contribution
      const bool inwin = !(g.off & 0x80000000u);
      const int base = (int)(g.off & 0x3FFFFFFFu);
      const int pix0 = (base << 2) >> 2;               // image pixel of corner 0 when the sample leaves the window
      float sq[4];
      {
        float v[4][8];
        // the lane's eight channels of the chunk: 8q .. 8q+7, or (FUSED) 4q .. 4q+3 and 16+4q .. 16+4q+3
        constexpr int C1 = FUSED ? 4 : 8, C2 = FUSED ? 16 : 4;     // offsets of the two 4-channel halves: q * C1 and q * C1 + C2
        auto ld8 = [&](const auto* p, float (&o)[8]) {
          typedef std::remove_cv_t<std::remove_reference_t<decltype(*p)>> PT;
          if constexpr (FUSED) {
            const f32x4 a0 = *(const f32x4*)p, a1 = *(const f32x4*)(p + C2);
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = a0[e]; o[4 + e] = a1[e]; }
          } else {
            load8f<PT>(p, o);
          }
        };
        if (inwin) {
          // the window is zero-filled outside the image, so invalid corners read zeros
          const float* c0 = xw + base * 32 + q * C1;
          ld8(c0, v[0]); ld8(c0 + 32, v[1]); ld8(c0 + WC * 32, v[2]); ld8(c0 + WC * 32 + 32, v[3]);
        } else {
          const T* c0 = ximg + (long)pix0 * x_stride + chunk * 32 + q * C1;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (g.valid & (1u << c)) ld8(c0 + ((long)(c >> 1) * W + (c & 1)) * x_stride, v[c]);
            else {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
            }
          }
        }
        // f32 FMAs in both modes.  (Round 3 used v_dot2_f32_f16 here for f16 data; with a second process on the GPU that
        // kernel's d(offset) outputs changed from run to run -- tools/probe_contention*.py --, this form's do not.)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float a2 = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) a2 = fmaf(dv[t][e], v[c][e], a2);
          sq[c] = a2;
        }
      }
      const float w0 = g.hh * g.hw, w1 = g.hh * g.lw, w2 = g.lh * g.hw, w3 = g.lh * g.lw;
#ifdef CTDET_DEBUG_COL2IM
      if (g_col2im_dbg) {      // per (pixel, tap, chunk, q): the four corner dots, the geometry words, corner-0 address
        float* r = g_col2im_dbg + ((((long)m * 9 + (t0 + t)) * nch + chunk) * 4 + q) * 8;
        r[0] = sq[0]; r[1] = sq[1]; r[2] = sq[2]; r[3] = sq[3]; r[4] = g.hw; r[5] = g.lw; r[6] = __uint_as_float(g.off); r[7] = (float)inwin;
      }
#endif
      s_val[t] += w0 * sq[0] + w1 * sq[1] + w2 * sq[2] + w3 * sq[3];
      s_dh[t] += -g.hw * sq[0] - g.lw * sq[1] + g.hw * sq[2] + g.lw * sq[3];    // d val / d h (kernel.cu:754-766)
      s_dw[t] += -g.hh * sq[0] + g.hh * sq[1] - g.lh * sq[2] + g.lh * sq[3];    // d val / d w (kernel.cu:767-779)
      // input-gradient scatter (kernel.cu:871-949)
      const float wq[4] = {w0 * g.mask, w1 * g.mask, w2 * g.mask, w3 * g.mask};
      if (inwin) {
        int* a0 = dxw + base * 4 + q;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (!(g.valid & (1u << c))) continue;
          int* ap = a0 + ((c >> 1) * WC + (c & 1)) * 4;
          const float ws = wq[c] * fscale;
#pragma unroll
          for (int e = 0; e < 8; ++e) atomicAdd(ap + e * (NPX * 4), (int)rintf(ws * dv[t][e]));
        }
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (!(g.valid & (1u << c))) continue;
          float* ap = dximg + (long)(pix0 + (c >> 1) * W + (c & 1)) * Cin + chunk * 32 + q * (FUSED ? 4 : 8);
#pragma unroll
          for (int e = 0; e < 8; ++e) atomicAdd(ap + (FUSED ? (e & 3) + 16 * (e >> 2) : e), wq[c] * dv[t][e]);
        }
      }
    }
    __syncthreads();
    // ---- flush the dx window: one global atomic per touched element, 64 consecutive floats per wave instruction ----
    for (int i = tid; i < NPX * 32; i += 512) {
      const int pw = i >> 5, c = i & 31;
      const int y = wy0 + pw / WC, xx = wx0 + pw % WC;
      // [slot e of the owning lane][window pixel][lane group q]: channel c = 8q + e, or (FUSED) 4q + e % 4 + 16 (e / 4)
      const int vi = FUSED ? dxw[((c & 3) + 4 * (c >> 4)) * (NPX * 4) + pw * 4 + ((c & 15) >> 2)]
                           : dxw[(c & 7) * (NPX * 4) + pw * 4 + (c >> 3)];
      if (vi != 0 && y >= 0 && y < H && xx >= 0 && xx < W)
        atomicAdd(dximg + ((long)y * W + xx) * Cin + chunk * 32 + c, (float)vi * finv);
    }
  }
  // ---- d(offset), d(mask logit): reduce over the 4 channel-group lanes of a pixel ----
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t0 + t >= 9) continue;
    float a = s_val[t], bh = s_dh[t], bw = s_dw[t];
    a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
    bh += __shfl_xor(bh, 16, 64); bh += __shfl_xor(bh, 32, 64);
    bw += __shfl_xor(bw, 16, 64); bw += __shfl_xor(bw, 32, 64);
    if (q == 0) {
      const float mk = geo[t * (TH * TW) + pl].mask;
      dom_store(dom, dom_f16, m, dom_stride, t0 + t, bh * mk, bw * mk, mask_is_prob ? a : a * mk * (1.f - mk));  // through the sigmoid
    }
  }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline unsigned nblk256(long n) { return (unsigned)((n + 255) / 256); }

// 1024 block partials of (sum0, sum1) per channel + the two finalized sums the BN backward apply pass reads
size_t chan_reduce_workspace_bytes(int C) { return (size_t)(1024 * 2 + 2) * C * sizeof(float); }

static int chan_blocks(int M, int C, int N) {
  const int rows = 256 / (C / N);
  long nb = ((long)M + rows * 8 - 1) / (rows * 8);
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  return (int)nb;
}

template <typename T>
static int launch_bn_train_fwd_t(const T* y, int y_stride, const T* res, int res_stride, T* z, int z_stride, int M, int C,
                                 const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                 float* running_var, float* mean, float* invstd, float* scale, float* shift, void* workspace,
                                 int relu, hipStream_t s) {
  constexpr int N = VecT<T>::N;
  CTDET_CHECK(C % N == 0 && C / N <= 256, "bn: unsupported channel count %d", C);
  CTDET_CHECK(y_stride % N == 0 && z_stride % N == 0 && (!res || res_stride % N == 0) && ((((size_t)y | (size_t)z | (size_t)res)) & 15) == 0,
              "bn: tensors must be 16-byte aligned with pixel strides that are multiples of %d", N);
  ChanRedArgs<T> a = {};
  a.y = y; a.y_stride = y_stride; a.M = M; a.C = C; a.mode = 0; a.partial = (float*)workspace;
  const int nb = chan_blocks(M, C, N);
  hipLaunchKernelGGL(chan_reduce_kernel<T>, dim3(nb), dim3(256), 0, s, a);
  hipLaunchKernelGGL(chan_finalize_kernel, dim3(C), dim3(256), 0, s, (const float*)workspace, nb, C, M, 0, eps,
                     momentum, gamma, beta, mean, invstd, scale, shift, running_mean, running_var);
  const int CV = C / N;
  if ((CV & (CV - 1)) == 0) {
    int sh = 0;
    while ((1 << sh) < CV) ++sh;
    const long ppb = (256 >> sh) * AA_ROWS;
    hipLaunchKernelGGL(affine_act_rows_kernel<T>, dim3((unsigned)(((long)M + ppb - 1) / ppb)), dim3(256), 0, s, y, y_stride,
                       (const float*)scale, (const float*)shift, res, res_stride, z, z_stride, (long)M, sh, relu);
  } else {
    hipLaunchKernelGGL(affine_act_kernel<T>, dim3(nblk256((long)M * CV)), dim3(256), 0, s, y, y_stride, (const float*)scale,
                       (const float*)shift, res, res_stride, z, z_stride, (long)M, C, relu);
  }
  CTDET_LAUNCH_CHECK();
  return 0;
}

template <typename T>
static int launch_bn_train_bwd_t(const T* dz, int dz_stride, const T* z, int z_stride, const T* y, int y_stride,
                                 const float* mean, const float* invstd, const float* scale, int M, int C, int relu, T* dy,
                                 int dy_stride, T* dres, int dres_stride, float* dgamma, float* dbeta, float grad_mult,
                                 void* workspace, hipStream_t s) {
  constexpr int N = VecT<T>::N;
  CTDET_CHECK(C % N == 0 && C / N <= 256, "bn_bwd: unsupported channel count %d", C);
  CTDET_CHECK(dz_stride % N == 0 && dy_stride % N == 0 && (!z || z_stride % N == 0) && (!y || y_stride % N == 0) &&
                  (!dres || dres_stride % N == 0) && ((((size_t)dz | (size_t)z | (size_t)y | (size_t)dy | (size_t)dres)) & 15) == 0,
              "bn_bwd: tensors must be 16-byte aligned with pixel strides that are multiples of %d", N);
  ChanRedArgs<T> a = {};
  a.y = y; a.y_stride = y_stride; a.dz = dz; a.dz_stride = dz_stride; a.z = z; a.z_stride = z_stride;
  a.mean = mean; a.invstd = invstd; a.M = M; a.C = C; a.mode = 1; a.relu = relu; a.partial = (float*)workspace;
  const int nb = chan_blocks(M, C, N);
  hipLaunchKernelGGL(chan_reduce_kernel<T>, dim3(nb), dim3(256), 0, s, a);
  float* sums = (float*)workspace + (size_t)1024 * 2 * C;   // [2][C]: sum g, sum g*xhat
  hipLaunchKernelGGL(chan_finalize_kernel, dim3(C), dim3(256), 0, s, (const float*)workspace, nb, C, M, 1,
                     grad_mult, 0.f, (const float*)nullptr, (const float*)nullptr, dbeta, dgamma, sums, sums + C,
                     (float*)nullptr, (float*)nullptr);
  const int CV = C / N;
  if ((CV & (CV - 1)) == 0 && CV <= 256) {
    int sh = 0;
    while ((1 << sh) < CV) ++sh;
    const long ppb = (256 >> sh) * BN_ROWS;
    hipLaunchKernelGGL(bn_bwd_apply_rows_kernel<T>, dim3((unsigned)(((long)M + ppb - 1) / ppb)), dim3(256), 0, s, dz, dz_stride, z,
                       z_stride, y, y_stride, mean, invstd, scale, (const float*)sums, (const float*)(sums + C), dy, dy_stride,
                       dres, dres_stride, (long)M, sh, relu);
  } else {
    hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(nblk256((long)M * CV)), dim3(256), 0, s, dz, dz_stride, z, z_stride, y,
                       y_stride, mean, invstd, scale, (const float*)sums, (const float*)(sums + C), dy, dy_stride, dres,
                       dres_stride, (long)M, C, relu);
  }
  CTDET_LAUNCH_CHECK();
  return 0;
}

int launch_bn_train_fwd(const f16* y, int y_stride, const f16* res, int res_stride, f16* z, int z_stride, int M, int C,
                        const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                        float* running_var, float* mean, float* invstd, float* scale, float* shift, void* workspace,
                        int relu, hipStream_t s) {
  return launch_bn_train_fwd_t<f16>(y, y_stride, res, res_stride, z, z_stride, M, C, gamma, beta, eps, momentum, running_mean,
                                    running_var, mean, invstd, scale, shift, workspace, relu, s);
}
int launch_bn_train_fwd_f32(const float* y, int y_stride, const float* res, int res_stride, float* z, int z_stride, int M, int C,
                            const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                            float* running_var, float* mean, float* invstd, float* scale, float* shift, void* workspace,
                            int relu, hipStream_t s) {
  return launch_bn_train_fwd_t<float>(y, y_stride, res, res_stride, z, z_stride, M, C, gamma, beta, eps, momentum, running_mean,
                                      running_var, mean, invstd, scale, shift, workspace, relu, s);
}
int launch_bn_train_bwd(const f16* dz, int dz_stride, const f16* z, int z_stride, const f16* y, int y_stride,
                        const float* mean, const float* invstd, const float* scale, int M, int C, int relu, f16* dy,
                        int dy_stride, f16* dres, int dres_stride, float* dgamma, float* dbeta, float grad_mult,
                        void* workspace, hipStream_t s) {
  return launch_bn_train_bwd_t<f16>(dz, dz_stride, z, z_stride, y, y_stride, mean, invstd, scale, M, C, relu, dy, dy_stride, dres,
                                    dres_stride, dgamma, dbeta, grad_mult, workspace, s);
}
int launch_bn_train_bwd_f32(const float* dz, int dz_stride, const float* z, int z_stride, const float* y, int y_stride,
                            const float* mean, const float* invstd, const float* scale, int M, int C, int relu, float* dy,
                            int dy_stride, float* dres, int dres_stride, float* dgamma, float* dbeta, float grad_mult,
                            void* workspace, hipStream_t s) {
  return launch_bn_train_bwd_t<float>(dz, dz_stride, z, z_stride, y, y_stride, mean, invstd, scale, M, C, relu, dy, dy_stride,
                                      dres, dres_stride, dgamma, dbeta, grad_mult, workspace, s);
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 weight gradient, window form (maps divisible by 8x32, Cin % 32 == 0).  The generic kernel above
// gathers every tap's pixels separately (nine reads of the input through L2) and ends in 8192 f32 atomics per workgroup on
// ~1000 workgroups.  Here a workgroup owns 32 couts x 32 cin x all 9 taps and walks 8x32-pixel tiles: per tile the dY rows
// (256 x 32) and the (8+2) x (32+2) input window of its 32 channels go to LDS once and the taps read the window at shifted
// rows.  A wave takes two tile rows (two K slabs of 32 pixels), keeps its dY fragments for all nine taps and loads each of
// its four window rows' fragments once (row v serves tap row v of the first tile row and v-1 of the second): 56 transposing
// LDS reads for 72 MFMAs.  Partial sums: the four waves are added through LDS, then one f32 atomic per element at the end
// of the workgroup's tile range -- one workgroup per CU, so a 64->64 layer ends in 2.4 M atomics instead of 8.4 M.
// Measured on the batch-16 step (rocprofv3, us per launch, generic -> window): 64->64 @128^2 68 -> 55, 128->128 @64^2
// 63 -> 44, 256->256 @32^2 58 -> 40, head 64->256 @128^2 179 -> 142, offset convs (Cout 27) 68/43/26 -> 37/27/20.
// ------------------------------------------------------------------------------------------------
#define WW_LD 40   // LDS row pitch in f16 elements (32 channels + 8 pad: 80 bytes, spreads the transposing reads over banks)
// X3: f32 operands split on the way to LDS (hi and lo tiles of both operands: 95 KB, one workgroup per CU -- which is how
// the kernel is launched anyway), three MFMAs per product
// NW waves per workgroup: 4 (f16: two workgroups per CU) or 8 (X3: the 95 KB of split tiles allow one workgroup per CU, so the
// eight waves that keep a SIMD's two slots busy must come from that one workgroup: a wave then owns ONE tile row)
template <bool X3, int NW = X3 ? 8 : 4>
__global__ void __launch_bounds__(64 * NW, 2) conv_wgrad_win_kernel(const WgradArgs a) {
  constexpr int TH = 8, TW = 32, WC = TW + 2, NWIN = (TH + 2) * WC;   // 340 window pixels
  constexpr int NTH = 64 * NW, PPR = 16 * NW;                         // threads; pixels a load round covers (4 pieces per pixel)
  constexpr int NYP = TH * TW / PPR, NXP = (NWIN + PPR - 1) / PPR;    // dY / window pieces per thread (4 + 6, or 2 + 3)
  constexpr int RPW = TH / NW;                                        // tile rows per wave (2 or 1)
  constexpr int NT = X3 ? 2 : 1, ES = X3 ? 4 : 2;
  constexpr int LOY = TH * TW * WW_LD, LOX = NWIN * WW_LD;
  __shared__ __attribute__((aligned(16))) f16 sm[NT * (LOY + LOX)];   // 47680 B per tile set; the first 36864 B hold the f32 sums at the end
  f16* const sY = sm;
  f16* const sX = sm + NT * LOY;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // 1-D grid of gx*gy*split workgroups.  Consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2):
  // the gx*gy workgroups that walk the SAME pixel range (all channel tiles of it) are given ids 8 apart, so that they share
  // an XCD and its L2 serves their common dY / input tiles -- the kernel is bound by operand traffic.
  const int gx = a.Cin / 32, gy = (a.Cout + 31) / 32, nxy = gx * gy, split = a.msplit;
  int xy, bz;
  if ((split & 7) == 0) { const int slot = blockIdx.x >> 3; bz = (slot / nxy) * 8 + (blockIdx.x & 7); xy = slot % nxy; }
  else { xy = blockIdx.x % nxy; bz = blockIdx.x / nxy; }
  const int c0 = (xy % gx) * 32, n0 = (xy / gx) * 32;
  const int tiles_x = a.W / TW, tiles_y = a.H / TH, ntiles = a.B * tiles_y * tiles_x;
  const int per = (ntiles + split - 1) / split;
  const int t_begin = bz * per, t_end = t_begin + per < ntiles ? t_begin + per : ntiles;

  f32x4 acc[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Operand pieces of 8 channels, piece = tid + 256*i: dY 256 pixels x 4 cout groups (i < 4), window 340 pixels x 4 cin groups
  // (i < 6).  The channel group is tid & 3 for every piece; the pixel of a piece inside the tile / window does not depend on
  // the tile, so its offset and its border flags (first / last window row / column, past the window) are computed once.
  // Loads are unconditional from a selected 32-bit byte offset (0 when the pixel is outside the image) and the zeroing
  // happens when the registers go to LDS one tile later: nothing in the loop waits on a load it has just issued.
  const int g = tid & 3, pix = tid >> 2;
  const bool n_ok = n0 + g * 8 < a.Cout;
  const unsigned y_first = (unsigned)((((pix >> 5) * a.W + (pix & 31)) * a.dy_stride + n0 + g * 8) * ES);
  const unsigned y_step = (unsigned)((PPR / TW) * a.W * a.dy_stride * ES);   // a load round further down the tile
  int woff[NXP];
  unsigned wflags = 0;
#pragma unroll
  for (int i = 0; i < NXP; ++i) {
    const int wpx = pix + PPR * i, wr = wpx / WC, wc = wpx - wr * WC;
    woff[i] = wr * a.W + wc;
    const unsigned f = (wr == 0 ? 1u : 0u) | (wr == TH + 1 ? 2u : 0u) | (wc == 0 ? 4u : 0u) | (wc == WC - 1 ? 8u : 0u) |
                       (wpx >= NWIN ? 16u : 0u);
    wflags |= f << (5 * i);
  }
  const char* const xbase = (const char*)a.x + (size_t)(c0 + g * 8) * ES;
  const char* const ybase = (const char*)a.dy;
  Piece<X3> yv[NYP], xv[NXP];
  unsigned okbits = 0;
  auto fetch = [&](int tile) {
    const int txi = tile % tiles_x, tq = tile / tiles_x, tyi = tq % tiles_y, b = tq / tiles_y;
    const int tx0 = txi * TW, ty0 = tyi * TH;
    const unsigned edge = (tyi == 0 ? 1u : 0u) | (tyi == tiles_y - 1 ? 2u : 0u) | (txi == 0 ? 4u : 0u) | (txi == tiles_x - 1 ? 8u : 0u);
    const unsigned bad = wflags & (edge * 0x02108421u | 0x21084210u);
    const int tile_pix = (b * a.H + ty0 - 1) * a.W + tx0 - 1;
    const unsigned ytile = (unsigned)(((b * a.H + ty0) * a.W + tx0) * a.dy_stride * ES);
#pragma unroll
    for (int i = 0; i < NYP; ++i) {
      const unsigned off = n_ok ? ytile + y_first + y_step * i : 0u;
      yv[i] = piece_load<X3>(ybase + off);
    }
    okbits = 0;
#pragma unroll
    for (int i = 0; i < NXP; ++i) {
      const bool ok = ((bad >> (5 * i)) & 31u) == 0u;
      const unsigned off = ok ? (unsigned)((tile_pix + woff[i]) * a.in_stride * ES) : 0u;
      xv[i] = piece_load<X3>(xbase + off);
      okbits |= ok ? 1u << i : 0u;
    }
  };
  const int grp = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  if (t_begin < t_end) fetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();   // the previous tile's fragment reads are done
#pragma unroll
    for (int i = 0; i < NYP; ++i) piece_store<X3>(sY + (pix + PPR * i) * WW_LD + g * 8, LOY, yv[i], n_ok);
#pragma unroll
    for (int i = 0; i < NXP; ++i)
      if (pix + PPR * i < NWIN) piece_store<X3>(sX + (pix + PPR * i) * WW_LD + g * 8, LOX, xv[i], (okbits >> i) & 1u);
    __syncthreads();
    if (tile + 1 < t_end) fetch(tile + 1);
    // dY fragments of this wave's tile rows (K slabs of 32 pixels), both cout tiles: kept for all nine taps
    Frag<X3> fy[RPW][2];
#pragma unroll
    for (int h = 0; h < RPW; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) fy[h][i] = frag_read<X3>(sY + ((RPW * wave + h) * TW + 8 * grp + q) * WW_LD + i * 16 + 4 * p, WW_LD, LOY);
    // window rows RPW*wave .. RPW*wave + RPW + 1: row v serves tap row v - h of the wave's tile row h
#pragma unroll
    for (int v = 0; v < RPW + 2; ++v)
#pragma unroll
      for (int ts = 0; ts < 3; ++ts) {
        Frag<X3> fx[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) fx[j] = frag_read<X3>(sX + ((RPW * wave + v) * WC + ts + 8 * grp + q) * WW_LD + j * 16 + 4 * p, WW_LD, LOX);
#pragma unroll
        for (int h = 0; h < RPW; ++h) {
          const int tr = v - h;
          if (tr < 0 || tr > 2) continue;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[tr * 3 + ts][i][j] = mma_frag<X3>(fy[h][i], fx[j], acc[tr * 3 + ts][i][j]);
        }
      }
  }
  // The waves hold partial sums of the same 32 x 32 x 9 block with the same lane -> element mapping.  f32 atomics to
  // L2 are what this kernel's fixed cost is made of, so the waves are summed through LDS first (wave 3 stores, 2 / 1 / 0
  // add their registers in turn -- no LDS atomics needed) and one atomic per element leaves the workgroup.
  float* const red = (float*)sm;    // [36 tiles][4 r][64 lanes]
  __syncthreads();
#pragma unroll
  for (int w = NW - 1; w >= 0; --w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float* slot = red + (((t * 2 + i) * 2 + j) * 4 + r) * 64 + lane;
              *slot = w == NW - 1 ? acc[t][i][j][r] : *slot + acc[t][i][j][r];
            }
    }
    __syncthreads();
  }
  // D[row = cout][col = cin]: lane held rows 4*(lane>>4)+r, column lane&15; k = tap*Cin + c (tap-major dW)
  for (int e = tid; e < 36 * 256; e += NTH) {
    const int tile = e >> 8, t = tile >> 2, i = (tile >> 1) & 1, j = tile & 1, r = (e >> 6) & 3, ln = e & 63;
    const int n = n0 + i * 16 + 4 * (ln >> 4) + r;
    if (n < a.Cout) wg_add(a, n, t * a.Cin + c0 + j * 16 + (ln & 15), red[e] * a.scale);
  }
}

// ------------------------------------------------------------------------------------------------
// Window-form weight gradient for the narrow layers at full resolution: KS x KS / stride 1 / pad KS/2 with 8 or 16 input
// channels and at most 16 couts -- DLA's 7x7 stem (3 -> 16, input padded to 8 channels) and level0 (16 -> 16), both on
// 512^2 maps.  The generic kernel reads the input once per tap (49 x 67 MB for the stem).  Here the (8+KS-1) x (32+KS-1)
// window of an 8x32-pixel tile sits in LDS as [pixel][CIN] with no padding, so that a 16-column MFMA operand is simply 16
// consecutive halves starting at a pixel: one tap of 16 channels, or two horizontally adjacent taps of 8 (row pitch of the
// transposing read = one pixel, rows overlap).  D = dY^T (16 couts x 32 pixels) x window (32 pixels x 16 columns); a wave
// owns two tile rows and loads each window row's operand once for both.  The odd tap of a 7-wide row pairs with a column
// that does not exist; those eight D columns are dropped in the epilogue.  Epilogue as in conv_wgrad_win_kernel.
// ------------------------------------------------------------------------------------------------
template <int KS, int CIN, bool X3>
__global__ void __launch_bounds__(256, 2) conv_wgrad_narrow_kernel(const WgradArgs a) {
  constexpr int TH = 8, TW = 32, PADK = KS / 2, WR = TH + KS - 1, WCOL = TW + KS - 1, NWIN = WR * WCOL;
  constexpr int SUB = CIN / 8;                             // 8-channel pieces per pixel
  constexpr int NROUND = (NWIN * SUB + 255) / 256;
  constexpr int NP = CIN == 8 ? (KS + 1) / 2 : KS;         // column tiles per kernel row
  constexpr int NT = KS * NP;
  constexpr int LDY = 24;                                  // dY row pitch: 16 couts + 8 pad
  constexpr int Y_ELEMS = TH * TW * LDY, X_ELEMS = (NWIN + 8) * CIN;
  constexpr int NS = X3 ? 2 : 1, ES = X3 ? 4 : 2;
  constexpr int SM_BYTES = NS * (X_ELEMS + Y_ELEMS) * 2 > NT * 1024 ? NS * (X_ELEMS + Y_ELEMS) * 2 : NT * 1024;
  static_assert(NROUND <= 6, "border flags are 5 bits per round in one word");
  __shared__ __attribute__((aligned(16))) unsigned char smraw[SM_BYTES];
  f16* const sY = (f16*)smraw;
  f16* const sX = sY + NS * Y_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = a.W / TW, tiles_y = a.H / TH, ntiles = a.B * tiles_y * tiles_x;
  const int per = (ntiles + gridDim.x - 1) / gridDim.x;
  const int t_begin = blockIdx.x * per, t_end = t_begin + per < ntiles ? t_begin + per : ntiles;

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // 8-channel pieces, piece = tid + 256*i: dY 256 pixels x 2 cout groups (i < 2); window NWIN pixels x SUB channel groups
  const int yg = tid & 1, ypix = tid >> 1;
  const bool n_ok = yg * 8 < a.Cout;
  const unsigned y_first = (unsigned)((((ypix >> 5) * a.W + (ypix & 31)) * a.dy_stride + yg * 8) * ES);
  const unsigned y_step = (unsigned)(4 * a.W * a.dy_stride * ES);     // 128 pixels of the tile = four rows further down
  const int sub = SUB == 2 ? tid & 1 : 0, wpix = SUB == 2 ? tid >> 1 : tid;
  constexpr int WSTEP = 256 / SUB;
  int woff[NROUND];
  unsigned wflags = 0, rep = 0, beyond = 0;
#pragma unroll
  for (int i = 0; i < NROUND; ++i) {
    const int wpx = wpix + WSTEP * i, wr = wpx / WCOL, wc = wpx - wr * WCOL;
    woff[i] = wr * a.W + wc;
    const unsigned f = (wr < PADK ? 1u : 0u) | (wr >= TH + PADK ? 2u : 0u) | (wc < PADK ? 4u : 0u) | (wc >= TW + PADK ? 8u : 0u) |
                       (wpx >= NWIN ? 16u : 0u);
    wflags |= f << (5 * i);
    rep |= 1u << (5 * i);
    beyond |= 16u << (5 * i);
  }
  const char* const xbase = (const char*)a.x + (size_t)(sub * 8) * ES;
  const char* const ybase = (const char*)a.dy;
  Piece<X3> yv[2], xv[NROUND];
  unsigned okbits = 0;
  auto fetch = [&](int tile) {
    const int txi = tile % tiles_x, tq = tile / tiles_x, tyi = tq % tiles_y, b = tq / tiles_y;
    const int tx0 = txi * TW, ty0 = tyi * TH;
    const unsigned edge = (tyi == 0 ? 1u : 0u) | (tyi == tiles_y - 1 ? 2u : 0u) | (txi == 0 ? 4u : 0u) | (txi == tiles_x - 1 ? 8u : 0u);
    const unsigned bad = wflags & (edge * rep | beyond);
    const int tile_pix = (b * a.H + ty0 - PADK) * a.W + tx0 - PADK;
    const unsigned ytile = (unsigned)(((b * a.H + ty0) * a.W + tx0) * a.dy_stride * ES);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned off = n_ok ? ytile + y_first + y_step * i : 0u;
      yv[i] = piece_load<X3>(ybase + off);
    }
    okbits = 0;
#pragma unroll
    for (int i = 0; i < NROUND; ++i) {
      const bool ok = ((bad >> (5 * i)) & 31u) == 0u;
      const unsigned off = ok ? (unsigned)((tile_pix + woff[i]) * a.in_stride * ES) : 0u;
      xv[i] = piece_load<X3>(xbase + off);
      okbits |= ok ? 1u << i : 0u;
    }
  };
  const int grp = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  if (t_begin < t_end) fetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();   // the previous tile's fragment reads are done
#pragma unroll
    for (int i = 0; i < 2; ++i) piece_store<X3>(sY + (ypix + 128 * i) * LDY + yg * 8, Y_ELEMS, yv[i], n_ok);
#pragma unroll
    for (int i = 0; i < NROUND; ++i)
      if (wpix + WSTEP * i < NWIN) piece_store<X3>(sX + (wpix + WSTEP * i) * CIN + sub * 8, X_ELEMS, xv[i], (okbits >> i) & 1u);
    __syncthreads();
    if (tile + 1 < t_end) fetch(tile + 1);
    Frag<X3> fy[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) fy[h] = frag_read<X3>(sY + ((2 * wave + h) * TW + 8 * grp + q) * LDY + 4 * p, LDY, Y_ELEMS);
    // window rows 2*wave .. 2*wave+KS: row v serves tap row v of the first tile row and v-1 of the second
#pragma unroll
    for (int v = 0; v <= KS; ++v)
#pragma unroll
      for (int pr = 0; pr < NP; ++pr) {
        const int s0 = CIN == 8 ? 2 * pr : pr;
        const Frag<X3> fx = frag_read<X3>(sX + ((2 * wave + v) * WCOL + s0 + 8 * grp + q) * CIN + 4 * p, CIN, X_ELEMS);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int tr = v - h;
          if (tr < 0 || tr >= KS) continue;
          acc[tr * NP + pr] = mma_frag<X3>(fy[h], fx, acc[tr * NP + pr]);
        }
      }
  }
  float* const red = (float*)smraw;    // [NT tiles][4 r][64 lanes]
  __syncthreads();
#pragma unroll
  for (int w = 3; w >= 0; --w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* slot = red + (t * 4 + r) * 64 + lane;
          *slot = w == 3 ? acc[t][r] : *slot + acc[t][r];
        }
    }
    __syncthreads();
  }
  // D[row = cout][col]: lane held rows 4*(lane>>4)+r, column lane&15 = (tap parity, channel) or channel; k = tap*CIN + c
  for (int e = tid; e < NT * 256; e += 256) {
    const int t = e >> 8, r = (e >> 6) & 3, ln = e & 63, col = ln & 15;
    const int n = 4 * (ln >> 4) + r, tr = t / NP, pr = t - tr * NP;
    const int ts = CIN == 8 ? 2 * pr + (col >> 3) : pr, c = CIN == 8 ? col & 7 : col;
    if (n < a.Cout && ts < KS) wg_add(a, n, (tr * KS + ts) * CIN + c, red[e] * a.scale);
  }
}

// ------------------------------------------------------------------------------------------------
// Weight gradients into the parameters' layout, all layers of a bucket in one launch.  The weight-gradient kernels accumulate
// tap-major [Cout][tap][Cin] (consecutive lanes = consecutive channels: the split-M partial sums arrive as coalesced
// atomics; the same atomics at OIHW addresses, 4*taps bytes apart, cost the step 6 ms).  This pass adds each finished sum
// to its place in the optimizer's flat OIHW gradient buffer: dst[(n*cin_real + c)*taps + tap] += src[(n*taps + tap)*cin_k + c]
// -- what one autograd add kernel per parameter did before (~230 launches per step).  Up to GS_MAX tensors per launch,
// descriptors by value.
// ------------------------------------------------------------------------------------------------
#define GS_MAX 24
struct GradScatterArgs {
  const float* src[GS_MAX]; float* dst[GS_MAX];
  int cin_real[GS_MAX], cin_k[GS_MAX], taps[GS_MAX], nelem[GS_MAX], blk0[GS_MAX + 1];
  int n;
};
__global__ void __launch_bounds__(256) grad_scatter_oihw_kernel(const GradScatterArgs a) {
  int t = 0;
  while (t + 1 < a.n && (int)blockIdx.x >= a.blk0[t + 1]) ++t;      // wave-uniform, n <= 24
  const int i = ((int)blockIdx.x - a.blk0[t]) * 256 + threadIdx.x;  // element of dst: (n, c, tap)
  if (i >= a.nelem[t]) return;
  const int taps = a.taps[t], cr = a.cin_real[t];
  const int tap = i % taps, nc = i / taps;
  const int c = nc % cr, n = nc / cr;
  a.dst[t][i] += a.src[t][((long)n * taps + tap) * a.cin_k[t] + c];
}

int launch_grad_scatter_oihw(const void* const* src, void* const* dst, const int* cout, const int* cin_real, const int* cin_k,
                             const int* taps, int n, hipStream_t s) {
  for (int base = 0; base < n; base += GS_MAX) {
    GradScatterArgs a = {};
    a.n = n - base < GS_MAX ? n - base : GS_MAX;
    int blocks = 0;
    for (int j = 0; j < a.n; ++j) {
      const int k = base + j;
      CTDET_CHECK(src[k] && dst[k] && cout[k] > 0 && cin_real[k] > 0 && cin_real[k] <= cin_k[k] && taps[k] > 0 &&
                      (long)cout[k] * cin_k[k] * taps[k] < (1L << 31),
                  "grad_scatter: bad descriptor %d", k);
      a.src[j] = (const float*)src[k]; a.dst[j] = (float*)dst[k];
      a.cin_real[j] = cin_real[k]; a.cin_k[j] = cin_k[k]; a.taps[j] = taps[k];
      a.nelem[j] = cout[k] * cin_real[k] * taps[k];
      a.blk0[j] = blocks;
      blocks += (a.nelem[j] + 255) / 256;
    }
    a.blk0[a.n] = blocks;
    if (blocks == 0) continue;
    hipLaunchKernelGGL(grad_scatter_oihw_kernel, dim3(blocks), dim3(256), 0, s, a);
    CTDET_LAUNCH_CHECK();
  }
  return 0;
}

static int device_cu_count() { return ctdet_device_cu_count(); }

template <bool X3>
static int launch_conv_wgrad_t(const WgradArgs& a0, hipStream_t s) {
  WgradArgs a = a0;
  constexpr int ES = X3 ? 4 : 2;
  const bool window_ok = a.stride == 1 && a.dil == 1 && a.R == a.S && a.pad == a.R / 2 && a.H % 8 == 0 && a.W % 32 == 0 &&
                         a.Ho == a.H && a.Wo == a.W && a.in_stride % 8 == 0 && a.dy_stride % 8 == 0 &&
                         (long)a.B * a.H * a.W * (a.in_stride > a.dy_stride ? a.in_stride : a.dy_stride) * ES < (1L << 31) &&
                         !(ctdet_tuning_flags() & CTDET_TUNE_NO_WGRAD_WINDOW);
  CTDET_CHECK((((size_t)a.x | (size_t)a.dy) & 15) == 0, "wgrad: x and dy must be 16-byte aligned");
  if (window_ok && a.Cout <= 16 && ((a.R == 7 && a.Cin == 8) || (a.R == 3 && a.Cin == 16))) {
    const int ntiles = a.B * (a.H / 8) * (a.W / 32);
    int blocks = device_cu_count();   // 1x / 2x / 4x CUs measured the same within noise; fewest atomics wins
    if (blocks > ntiles) blocks = ntiles;
    if (a.R == 7) hipLaunchKernelGGL((conv_wgrad_narrow_kernel<7, 8, X3>), dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_wgrad_narrow_kernel<3, 16, X3>), dim3(blocks), dim3(256), 0, s, a);
    CTDET_LAUNCH_CHECK();
    return 0;
  }
  if (window_ok && a.R == 3 && a.Cin % 32 == 0) {
    const int gx = a.Cin / 32, gy = (a.Cout + 31) / 32;
    const int ntiles = a.B * (a.H / 8) * (a.W / 32);
    // one workgroup per CU: the f32 atomics of the epilogue (9216 per workgroup) are the fixed cost, and a second
    // co-resident workgroup does not speed the tile loop up (measured: 256 / 384 / 512 workgroups -> 25.0 / 25.1 / 25.3 ms steps)
    const int ncu = device_cu_count();
    int split = ncu / (gx * gy);
    if (split < 1) split = 1;
    if (split > ntiles) split = ntiles;
    a.msplit = split;
    hipLaunchKernelGGL(conv_wgrad_win_kernel<X3>, dim3(gx * gy * split), dim3(X3 ? 512 : 256), 0, s, a);
    CTDET_LAUNCH_CHECK();
    return 0;
  }
  CTDET_CHECK(a.Cin % 8 == 0 && a.in_stride % 8 == 0 && a.dy_stride % 8 == 0 && a.Cout % 8 == 0,
              "wgrad: channel counts / strides must be multiples of 8 (Cin=%d Cout=%d)", a.Cin, a.Cout);
  const int gx = (a.K + WG_BK - 1) / WG_BK, gy = (a.Cout + WG_BN - 1) / WG_BN;
  int split = 1024 / (gx * gy);   // ~4 workgroups per CU: fewer, longer pixel ranges (the atomic epilogue is per workgroup)
  if (split < 1) split = 1;
  const int max_split = (a.M + 255) / 256;
  if (split > max_split) split = max_split;
  if (split < 1) split = 1;
  a.msplit = split;
  a.lw = a.lh = -1;
  if ((a.Wo & (a.Wo - 1)) == 0 && (a.Ho & (a.Ho - 1)) == 0) {
    a.lw = a.lh = 0;
    while ((1 << a.lw) < a.Wo) ++a.lw;
    while ((1 << a.lh) < a.Ho) ++a.lh;
  }
  if (a.perm_rs) hipLaunchKernelGGL((conv_wgrad_kernel<true, X3>), dim3(gx * gy * split), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((conv_wgrad_kernel<false, X3>), dim3(gx * gy * split), dim3(256), 0, s, a);
  CTDET_LAUNCH_CHECK();
  return 0;
}
int launch_conv_wgrad(const WgradArgs& a, hipStream_t s) { return launch_conv_wgrad_t<false>(a, s); }
int launch_conv_wgrad_x3(const WgradArgs& a, hipStream_t s) { return launch_conv_wgrad_t<true>(a, s); }

// ------------------------------------------------------------------------------------------------
// depth-to-space for the input gradient of a stride-2 3x3 convolution.  dx of such a conv splits into four output phases
// (row parity, column parity) that use 1, 2, 2 and 4 of the 9 taps; ops_train.conv_dgrad computes all four as ONE 2x2
// convolution over dY with 4*C output channels (16 tap-products per output quad instead of the 36 of the zero-stuffed
// form) and this kernel interleaves them:  dst[b, y, x, c] = src[b, (y+1)/2, (x+1)/2, ((y&1)*2 + (x&1))*C + c].
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) depth_to_space2_kernel(const T* __restrict__ src, int src_stride, T* __restrict__ dst,
                                                              int dst_stride, int B, int H, int W, int C, int Hs, int Ws) {
  typedef typename VecT<T>::type V;
  constexpr int N = VecT<T>::N;
  const int CV = C / N;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * H * W * CV) return;
  const int cv = (int)(idx % CV);
  long t = idx / CV;
  const int x = (int)(t % W); t /= W;
  const int y = (int)(t % H);
  const int b = (int)(t / H);
  const int phase = (y & 1) * 2 + (x & 1);
  const long sp = ((long)b * Hs + ((y + 1) >> 1)) * Ws + ((x + 1) >> 1);
  *(V*)(dst + ((long)(b * H + y) * W + x) * dst_stride + cv * N) = *(const V*)(src + sp * src_stride + phase * C + cv * N);
}

template <typename T>
static int launch_depth_to_space2_t(const T* src, int src_stride, T* dst, int dst_stride, int B, int H, int W, int C, int Hs, int Ws,
                                    hipStream_t s) {
  constexpr int N = VecT<T>::N;
  CTDET_CHECK(C % N == 0 && src_stride % N == 0 && dst_stride % N == 0 && src_stride >= 4 * C,
              "depth_to_space2: C=%d / strides must be multiples of %d, src_stride >= 4*C", C, N);
  CTDET_CHECK(Hs >= (H + 1) / 2 + ((H & 1) ? 0 : 1) && Ws >= (W + 1) / 2 + ((W & 1) ? 0 : 1), "depth_to_space2: source map %dx%d too small for %dx%d", Hs, Ws, H, W);
  const long total = (long)B * H * W * (C / N);
  if (total == 0) return 0;
  hipLaunchKernelGGL(depth_to_space2_kernel<T>, dim3(nblk256(total)), dim3(256), 0, s, src, src_stride, dst, dst_stride, B, H, W, C,
                     Hs, Ws);
  CTDET_LAUNCH_CHECK();
  return 0;
}
int launch_depth_to_space2(const void* src, int src_stride, void* dst, int dst_stride, int B, int H, int W, int C, int Hs, int Ws,
                           int dtype, hipStream_t s) {
  if (dtype == CTDET_F32)
    return launch_depth_to_space2_t<float>((const float*)src, src_stride, (float*)dst, dst_stride, B, H, W, C, Hs, Ws, s);
  return launch_depth_to_space2_t<f16>((const f16*)src, src_stride, (f16*)dst, dst_stride, B, H, W, C, Hs, Ws, s);
}

template <typename T>
static int launch_maxpool2x2_bwd_t(const T* x, int x_stride, const T* dz, int dz_stride, T* dx, int dx_stride, int B, int H,
                                   int W, int C, hipStream_t s) {
  constexpr int N = VecT<T>::N;
  CTDET_CHECK(C % N == 0 && H % 2 == 0 && W % 2 == 0 && x_stride % N == 0 && dz_stride % N == 0 && dx_stride % N == 0,
              "maxpool_bwd: bad shape");
  const long total = (long)B * (H / 2) * (W / 2) * (C / N);
  if (total == 0) return 0;
  hipLaunchKernelGGL(maxpool2x2_bwd_kernel<T>, dim3(nblk256(total)), dim3(256), 0, s, x, x_stride, dz, dz_stride, dx, dx_stride, B,
                     H, W, C);
  CTDET_LAUNCH_CHECK();
  return 0;
}
int launch_maxpool2x2_bwd(const f16* x, int x_stride, const f16* dz, int dz_stride, f16* dx, int dx_stride, int B, int H,
                          int W, int C, hipStream_t s) {
  return launch_maxpool2x2_bwd_t<f16>(x, x_stride, dz, dz_stride, dx, dx_stride, B, H, W, C, s);
}
int launch_maxpool2x2_bwd_f32(const float* x, int x_stride, const float* dz, int dz_stride, float* dx, int dx_stride, int B,
                              int H, int W, int C, hipStream_t s) {
  return launch_maxpool2x2_bwd_t<float>(x, x_stride, dz, dz_stride, dx, dx_stride, B, H, W, C, s);
}

template <typename T>
static int launch_dwconvT_bwd_t(const T* x, int x_stride, const T* dz, int dz_stride, const float* w, T* dx, int dx_stride,
                                float* dw, int B, int H, int W, int C, int f, hipStream_t s) {
  constexpr int N = VecT<T>::N;
  CTDET_CHECK(C % N == 0 && f % 2 == 0 && C / N <= 256 && x_stride % N == 0 && dz_stride % N == 0 && dx_stride % N == 0,
              "dwconvT_bwd: bad shape C=%d f=%d", C, f);
  const int S = 256 / (C / N);
  const size_t lds = (size_t)S * 4 * C * sizeof(float);   // <= 32 KB for every C
  long nb = ((long)B * (H + 1) * (W + 1) + S * 16 - 1) / (S * 16);
  if (nb > 512) nb = 512;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(dwconvT_dw_kernel<T>, dim3((unsigned)nb, (unsigned)(f * f)), dim3(256), lds, s, x, x_stride, dz, dz_stride,
                     dw, B, H, W, C, f);
  long nbx = ((long)B * H * W * (C / N) + 255) / 256;
  if (nbx > 4096) nbx = 4096;
  hipLaunchKernelGGL(dwconvT_dx_kernel<T>, dim3((unsigned)nbx), dim3(256), 0, s, dz, dz_stride, w, dx, dx_stride, B, H, W, C, f);
  CTDET_LAUNCH_CHECK();
  return 0;
}
int launch_dwconvT_bwd(const f16* x, int x_stride, const f16* dz, int dz_stride, const float* w, f16* dx, int dx_stride,
                       float* dw, int B, int H, int W, int C, int f, hipStream_t s) {
  return launch_dwconvT_bwd_t<f16>(x, x_stride, dz, dz_stride, w, dx, dx_stride, dw, B, H, W, C, f, s);
}
int launch_dwconvT_bwd_f32(const float* x, int x_stride, const float* dz, int dz_stride, const float* w, float* dx,
                           int dx_stride, float* dw, int B, int H, int W, int C, int f, hipStream_t s) {
  return launch_dwconvT_bwd_t<float>(x, x_stride, dz, dz_stride, w, dx, dx_stride, dw, B, H, W, C, f, s);
}

int launch_dcn_cols_window(const f16* x, int x_stride, const float* om, int om_stride, f16* col, int B, int H, int W, int Cin,
                           int mask_is_prob, hipStream_t s);   // conv_igemm.hip: the sampling kernel's LDS window
int launch_dcn_cols(const f16* x, int x_stride, const float* om, int om_stride, f16* col, int B, int H, int W, int Cin,
                    int mask_is_prob, hipStream_t s) {
  CTDET_CHECK(Cin % 8 == 0 && om_stride >= 27, "dcn_cols: bad shape");
  const long total = (long)B * H * W * 9 * (Cin / 8);
  if (total == 0) return 0;
  if (H % 8 == 0 && W % 16 == 0 && Cin % 32 == 0 && x_stride % 8 == 0 && om_stride % 4 == 0 && H <= 65534 && W <= 65534 &&
      (((size_t)x | (size_t)col | (size_t)om) & 15) == 0 && !(ctdet_tuning_flags() & CTDET_TUNE_NO_COL2IM_WINDOW))
    return launch_dcn_cols_window(x, x_stride, om, om_stride, col, B, H, W, Cin, mask_is_prob, s);
  hipLaunchKernelGGL(dcn_cols_kernel, dim3(nblk256(total)), dim3(256), 0, s, x, x_stride, om, om_stride, col, B, H, W, Cin,
                     mask_is_prob);
  CTDET_LAUNCH_CHECK();
  return 0;
}

template <typename T, bool FUSED = false>
static void launch_col2im_window(const T* dcol, const T* x, int x_stride, const float* om, int om_stride, float* dx, void* dom,
                                 int dom_stride, int dom_f16, int B, int H, int W, int Cin, int mask_is_prob, int chunked,
                                 hipStream_t s, const FusedDcol fz = FusedDcol()) {
  const unsigned tiles = (unsigned)(B * (H / 8) * (W / 16));
  const int ncu = ctdet_device_cu_count();    // one workgroup per CU at a time: with fewer tiles, split a tile's taps
  if ((int)tiles * 3 <= ncu)
    hipLaunchKernelGGL((dcn_col2im_window_kernel<3, T, FUSED>), dim3(tiles, 3), dim3(512), 0, s, dcol, x, x_stride, om, om_stride, dx, dom,
                       dom_stride, dom_f16, B, H, W, Cin, mask_is_prob, chunked, fz);
  else if ((int)tiles * 2 <= ncu)
    hipLaunchKernelGGL((dcn_col2im_window_kernel<5, T, FUSED>), dim3(tiles, 2), dim3(512), 0, s, dcol, x, x_stride, om, om_stride, dx, dom,
                       dom_stride, dom_f16, B, H, W, Cin, mask_is_prob, chunked, fz);
  else
    hipLaunchKernelGGL((dcn_col2im_window_kernel<9, T, FUSED>), dim3(tiles), dim3(512), 0, s, dcol, x, x_stride, om, om_stride, dx, dom,
                       dom_stride, dom_f16, B, H, W, Cin, mask_is_prob, chunked, fz);
}

// d(columns) GEMM + scatter in one kernel (f16x3 training mode): dy f32 [M][dy_stride] with K channels (multiple of 32; channels
// beyond the layer's couts zero), wpk / wscale from ctdet_pack_weights_x3 (layout 5, transposed 3).  0 if launched, 1 if the
// shape does not qualify (the caller then produces d(columns) and calls ctdet_dcn_col2im_coord).
int launch_dcn_col2im_fused(const float* dy, int dy_stride, int K, const void* wpk, const float* wscale, const float* x, int x_stride,
                            const float* om, int om_stride, float* dx, float* dom, int dom_stride, int B, int H, int W, int Cin,
                            int mask_is_prob, hipStream_t s) {
  CTDET_CHECK(dom_stride >= 27 && dom_stride <= 64, "dcn_col2im: dom_stride=%d", dom_stride);
  if ((long)B * H * W == 0) return 0;
  if (!(H % 8 == 0 && W % 16 == 0 && Cin % 32 == 0 && x_stride % 4 == 0 && K % 32 == 0 && K > 0 && dy_stride >= K && dy_stride % 4 == 0 &&
        ((((size_t)x | (size_t)dy | (size_t)wpk | (size_t)wscale)) & 15) == 0) || (ctdet_tuning_flags() & CTDET_TUNE_NO_COL2IM_WINDOW))
    return 1;
  FusedDcol fz;
  fz.dy = dy; fz.dy_stride = dy_stride; fz.K = K; fz.wpk = (const f16*)wpk; fz.wscale = wscale;
  launch_col2im_window<float, true>(nullptr, x, x_stride, om, om_stride, dx, (void*)dom, dom_stride, 0, B, H, W, Cin, mask_is_prob, 1, s, fz);
  CTDET_LAUNCH_CHECK();
  return 0;
}

int launch_dcn_col2im_coord(const f16* dcol, const f16* x, int x_stride, const float* om, int om_stride, float* dx,
                            void* dom, int dom_stride, int dom_f16, int B, int H, int W, int Cin, int mask_is_prob, int chunked,
                            hipStream_t s) {
  CTDET_CHECK(dom_stride >= 27 && dom_stride <= 64, "dcn_col2im: dom_stride=%d", dom_stride);
  CTDET_CHECK(!chunked || Cin % 32 == 0, "dcn_col2im: the chunked dcol layout needs Cin %% 32 == 0 (Cin=%d)", Cin);
  CTDET_CHECK(Cin % 8 == 0, "dcn_col2im: Cin=%d must be a multiple of 8", Cin);
  const long nwork = (long)B * H * W * 9;
  if (nwork == 0) return 0;
  if (H % 8 == 0 && W % 16 == 0 && Cin % 32 == 0 && x_stride % 8 == 0 && !(ctdet_tuning_flags() & CTDET_TUNE_NO_COL2IM_WINDOW)) {
    launch_col2im_window<f16>(dcol, x, x_stride, om, om_stride, dx, dom, dom_stride, dom_f16, B, H, W, Cin, mask_is_prob, chunked, s);
    CTDET_LAUNCH_CHECK();
    return 0;
  }
  long nb = (nwork + 3) / 4;
  if (nb > 256 * 32) nb = 256 * 32;
  hipLaunchKernelGGL(dcn_col2im_coord_kernel, dim3((unsigned)nb), dim3(256), 0, s, dcol, x, x_stride, om, om_stride, dx, dom,
                     dom_stride, dom_f16, B, H, W, Cin, mask_is_prob, chunked);
  CTDET_LAUNCH_CHECK();
  return 0;
}

// ================================================================================================================
// f32-tensor training kernels.  BatchNorm, pooling and the depthwise up-convolution are the templates above instantiated
// for float (16-byte vectors of 4 channels): they serve both modes that keep f32 tensors -- f32 (the reference's own
// arithmetic: contractions as f32 FMA chains) and f16x3 (contractions as three f16 MFMA products, f32 accumulation).
// What is specific to the f32 mode is below: the weight gradient as plain f32 FMAs and the DCNv2 scatter with f32 atomics.
// ================================================================================================================
// dW[n][k] += scale * sum_m dY[m][n] * im2col(x)[m][k], k = tap*Cin + c.  Block = 16 couts x 16 k, 16 pixels per step
// staged in LDS, pixel range split over blockIdx.z, f32 atomics at the end.
struct WgradArgsF {
  const float* x; const float* dy; float* dw;
  int B, H, W, Cin, in_stride, Cout, Ho, Wo, dy_stride, R, S, stride, pad, dil, K, M, msplit;
  float scale;
  int perm_rs, perm_cin, cin_real, cout_real;
};
__global__ void __launch_bounds__(256) conv_wgrad_f32_kernel(const WgradArgsF a) {
  __shared__ float sdy[16][17], sa[16][17];
  const int tk = threadIdx.x & 15, tn = threadIdx.x >> 4;
  const int k0 = blockIdx.x * 16, n0 = blockIdx.y * 16;
  const long per = ((long)a.M + a.msplit - 1) / a.msplit;
  const long mb = (long)blockIdx.z * per;
  const long me = mb + per < a.M ? mb + per : a.M;
  // the k this thread LOADS (column tk of the staged A tile): tap and channel
  const int kl = k0 + tk;
  const bool k_ok = kl < a.K;
  const int tap = k_ok ? kl / a.Cin : 0, ch = k_ok ? kl - tap * a.Cin : 0;
  const int tr = tap / a.S, ts = tap - tr * a.S;
  float acc = 0.f;
  for (long m0 = mb; m0 < me; m0 += 16) {
    const long m = m0 + tn;                 // pixel this thread loads (row tn of both staged tiles)
    float dyv = 0.f, av = 0.f;
    if (m < me) {
      if (n0 + tk < a.Cout) dyv = a.dy[m * a.dy_stride + n0 + tk];
      if (k_ok) {
        const int wo = (int)(m % a.Wo);
        const long t = m / a.Wo;
        const int ho = (int)(t % a.Ho), b = (int)(t / a.Ho);
        const int hi = ho * a.stride - a.pad + tr * a.dil, wi = wo * a.stride - a.pad + ts * a.dil;
        if (hi >= 0 && hi < a.H && wi >= 0 && wi < a.W) av = a.x[((long)(b * a.H + hi) * a.W + wi) * a.in_stride + ch];
      }
    }
    sdy[tn][tk] = dyv; sa[tn][tk] = av;
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 16; ++p) acc = fmaf(sdy[p][tn], sa[p][tk], acc);   // output element (n0 + tn, k0 + tk)
    __syncthreads();
  }
  if (n0 + tn < a.Cout && k_ok && acc != 0.f) wg_add(a, n0 + tn, kl, acc * a.scale);
}

// col[m][tap*Cin + c] = mask * bilinear(x), f32.  A workgroup walks (pixel, tap) pairs, 256 / (Cin / 4) at a time: the sampling
// geometry of a pair (floor, the four weights, the sigmoid of the mask logit) is computed ONCE, by one thread, and handed to the
// Cin / 4 threads that each blend 4 channels of the four corners (float4 loads; consecutive threads = consecutive channels: every
// corner is one contiguous run, every store of a pair one contiguous Cin * 4 bytes) -- round 4: the first f32 form had every
// one of those threads re-derive the geometry (16 times for 64 channels).  The blend in the reference's operation order
// (kernel.cu:666-699: v1 w1 + v2 w2 + v3 w3 + v4 w4, then * mask :854-861).
struct ColsGeoS { float w[4]; int off[4]; float mask; };
constexpr int COLS_PASSES = 8;
__global__ void __launch_bounds__(256) dcn_cols_f32_kernel(const float* __restrict__ x, int x_stride,
                                                           const float* __restrict__ om, int om_stride, float* __restrict__ col,
                                                           int B, int H, int W, int Cin, int mask_is_prob) {
  __shared__ ColsGeoS geo[2][64];
  const int CV = Cin >> 2;                       // <= 256
  const int PP = 256 / CV;                       // pairs per pass (>= 1), <= 64 (Cin >= 16)
  const long npairs = (long)B * H * W * 9;
  const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
  for (int pass = 0; pass < COLS_PASSES; ++pass) {
    const long pair0 = ((long)blockIdx.x * COLS_PASSES + pass) * PP;
    if (pair0 >= npairs) break;                  // block-uniform
    ColsGeoS* gs = geo[pass & 1];
    if (threadIdx.x < PP && pair0 + threadIdx.x < npairs) {
      const long pr = pair0 + threadIdx.x;
      const int tap = (int)(pr % 9);
      const long m = pr / 9;
      const int wo = (int)(m % W);
      const long t2 = m / W;
      const int ho = (int)(t2 % H), b = (int)(t2 / H);
      const DcnGeom g = dcn_geom(om + m * om_stride, tap, b, ho, wo, H, W, x_stride, mask_is_prob);
      ColsGeoS o;
#pragma unroll
      for (int q = 0; q < 4; ++q) { o.w[q] = g.w[q]; o.off[q] = (int)g.off[q]; }
      o.mask = g.mask;
      gs[threadIdx.x] = o;
    }
    __syncthreads();                             // (two geometry buffers: the next pass writes the other one)
    const long pr = pair0 + pl;
    if (pl < PP && pr < npairs) {
      const ColsGeoS g = gs[pl];
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      f32x4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = g.off[q] >= 0 ? *(const f32x4*)(x + g.off[q] + cv * 4) : z4;
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (g.w[0] * v[0][e] + g.w[1] * v[1][e] + g.w[2] * v[2][e] + g.w[3] * v[3][e]) * g.mask;
      const int tap = (int)(pr % 9);
      const long m = pr / 9;
      *(f32x4*)(col + m * (9L * Cin) + (long)tap * Cin + cv * 4) = o;
    }
  }
}

// the generic coordinate / col2im kernel above for f32 columns and inputs (same wave-per-(pixel, tap) structure)
__global__ void __launch_bounds__(256) dcn_col2im_coord_f32_kernel(const float* __restrict__ dcol, const float* __restrict__ x,
                                                                   int x_stride, const float* __restrict__ om, int om_stride,
                                                                   float* __restrict__ dx, float* __restrict__ dom, int dom_stride, int B,
                                                                   int H, int W, int Cin, int mask_is_prob, int chunked) {
  const int lane = threadIdx.x & 63;
  const long wave_id = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwork = (long)B * H * W * 9;
  for (long wk = wave_id; wk < nwork; wk += (long)gridDim.x * 4) {
    const int tap = (int)(wk % 9);
    const long m = wk / 9;
    const int wo = (int)(m % W);
    const long t2 = m / W;
    const int ho = (int)(t2 % H), b = (int)(t2 / H);
    const DcnGeom g = dcn_geom(om + m * om_stride, tap, b, ho, wo, H, W, x_stride, mask_is_prob);
    float val_dot = 0.f, dh = 0.f, dwv = 0.f;
    if (g.inside) {
      // dcol row of a pixel: [tap][Cin], or chunked [Cin/32][tap][32]
      const float* dcp = dcol + m * (9L * Cin) + (chunked ? tap * 32 : tap * Cin);
      for (int c = lane; c < Cin; c += 64) {
        const float d = dcp[chunked ? (c >> 5) * 288 + (c & 31) : c];
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = g.off[q] >= 0 ? x[g.off[q] + c] : 0.f;
        val_dot += d * (g.w[0] * v[0] + g.w[1] * v[1] + g.w[2] * v[2] + g.w[3] * v[3]);
        dh += d * (-g.hw * v[0] - g.lw * v[1] + g.hw * v[2] + g.lw * v[3]);
        dwv += d * (-g.hh * v[0] + g.hh * v[1] - g.lh * v[2] + g.lh * v[3]);
        const float dm = d * g.mask;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (g.off[q] >= 0) atomicAdd(dx + g.off[q] / x_stride * (long)Cin + c, g.w[q] * dm);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      val_dot += __shfl_down(val_dot, o, 64);
      dh += __shfl_down(dh, o, 64);
      dwv += __shfl_down(dwv, o, 64);
    }
    if (lane == 0)
      dom_store(dom, 0, m, dom_stride, tap, dh * g.mask, dwv * g.mask, mask_is_prob ? val_dot : val_dot * g.mask * (1.f - g.mask));
  }
}

int launch_conv_wgrad_f32(const WgradArgs& h, hipStream_t s) {
  WgradArgsF a;
  a.x = (const float*)h.x; a.dy = (const float*)h.dy; a.dw = h.dw;
  a.B = h.B; a.H = h.H; a.W = h.W; a.Cin = h.Cin; a.in_stride = h.in_stride; a.Cout = h.Cout; a.Ho = h.Ho; a.Wo = h.Wo;
  a.dy_stride = h.dy_stride; a.R = h.R; a.S = h.S; a.stride = h.stride; a.pad = h.pad; a.dil = h.dil; a.K = h.K; a.M = h.M;
  a.scale = h.scale; a.perm_rs = h.perm_rs; a.perm_cin = h.perm_cin; a.cin_real = h.cin_real; a.cout_real = h.cout_real;
  const int nkb = (a.K + 15) / 16, nnb = (a.Cout + 15) / 16;
  long ms = 4096 / ((long)nkb * nnb);
  if (ms < 1) ms = 1;
  const long maxs = ((long)a.M + 63) / 64;
  if (ms > maxs) ms = maxs;
  if (ms > 65535) ms = 65535;
  a.msplit = (int)ms;
  if (a.M == 0) return 0;
  hipLaunchKernelGGL(conv_wgrad_f32_kernel, dim3(nkb, nnb, (unsigned)ms), dim3(256), 0, s, a);
  CTDET_LAUNCH_CHECK();
  return 0;
}

int launch_dcn_cols_f32(const float* x, int x_stride, const float* om, int om_stride, float* col, int B, int H, int W, int Cin,
                        int mask_is_prob, hipStream_t s) {
  CTDET_CHECK(om_stride >= 27 && Cin % 4 == 0 && Cin >= 16 && Cin <= 1024 && x_stride % 4 == 0 && ((((size_t)x | (size_t)col)) & 15) == 0,
              "dcn_cols(f32): Cin=%d (16..1024) / x_stride=%d must be multiples of 4, tensors 16-byte aligned", Cin, x_stride);
  CTDET_CHECK((long)B * H * W * x_stride < (1L << 31), "dcn_cols(f32): input too large for 32-bit element offsets");
  const long npairs = (long)B * H * W * 9;
  if (npairs == 0) return 0;
  const int PP = 256 / (Cin / 4);
  const long per_block = (long)PP * COLS_PASSES;
  hipLaunchKernelGGL(dcn_cols_f32_kernel, dim3((unsigned)((npairs + per_block - 1) / per_block)), dim3(256), 0, s, x, x_stride, om, om_stride,
                     col, B, H, W, Cin, mask_is_prob);
  CTDET_LAUNCH_CHECK();
  return 0;
}

// window = 1 (the f16x3 training mode): the LDS-window scatter with fixed-point accumulation where the shape allows;
// window = 0 (the f32 mode): the generic kernel, plain f32 atomics
int launch_dcn_col2im_coord_f32(const float* dcol, const float* x, int x_stride, const float* om, int om_stride, float* dx,
                                float* dom, int dom_stride, int B, int H, int W, int Cin, int mask_is_prob, int chunked, int window,
                                hipStream_t s) {
  CTDET_CHECK(dom_stride >= 27 && dom_stride <= 64, "dcn_col2im: dom_stride=%d", dom_stride);
  const long nwork = (long)B * H * W * 9;
  if (nwork == 0) return 0;
  if (window && H % 8 == 0 && W % 16 == 0 && Cin % 32 == 0 && x_stride % 4 == 0 && ((((size_t)x | (size_t)dcol)) & 15) == 0 &&
      !(ctdet_tuning_flags() & CTDET_TUNE_NO_COL2IM_WINDOW)) {
    launch_col2im_window<float>(dcol, x, x_stride, om, om_stride, dx, (void*)dom, dom_stride, 0, B, H, W, Cin, mask_is_prob, chunked, s);
    CTDET_LAUNCH_CHECK();
    return 0;
  }
  CTDET_CHECK(!chunked || Cin % 32 == 0, "dcn_col2im: the chunked dcol layout needs Cin %% 32 == 0 (Cin=%d)", Cin);
  long nb = (nwork + 3) / 4;
  if (nb > 256 * 32) nb = 256 * 32;
  hipLaunchKernelGGL(dcn_col2im_coord_f32_kernel, dim3((unsigned)nb), dim3(256), 0, s, dcol, x, x_stride, om, om_stride, dx,
                     dom, dom_stride, B, H, W, Cin, mask_is_prob, chunked);
  CTDET_LAUNCH_CHECK();
  return 0;
}
