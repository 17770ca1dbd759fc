// DLA base layers fused for inference: image normalisation -> 7x7 stem (3->16) -> level0 3x3 (16->16) -> level1 3x3
// stride 2 (16->32), each conv with its folded BatchNorm + ReLU (reference dla.py:204-215 base_layer/level0/level1, called
// from centernet.py:206-207 after the (x/255 - mean)/std of centernet.py:193-200).
//
// Unfused, these layers are HBM streams: 0.27 GB in / 0.54 GB out for the stem, 0.54 + 0.54 for level0, 0.54 + 0.27 for
// level1 at batch 64 x 512^2 (f16, 16 channels at full resolution) -- 0.74 ms of the 8.6 ms step, plus 0.08 ms of
// preprocessing.  Here a workgroup owns an 8x16 tile of level1 outputs and walks back up the receptive field: 17x33 level0
// outputs, 19x35 stem outputs, 25x41 input pixels.  The intermediate maps live in LDS only (f16, the same rounding points as
// the unfused path); HBM sees the uint8 image once (50 MB) and the level1 map once (0.27 GB).  Recompute overhead from the
// halos: 1.30x for the stem, 1.10x for level0.
//
// MFMA forms (v_mfma_f32_16x16x32_f16, A = weights [cout][k], B = pixels [k][pixel], D[cout][pixel]):
//   stem    one K step per kernel row: k = (s, c) with s = 0..7 (column 7 zero weight), c = 0..3 (channel 3 zero) -- the input
//           window is stored 4 channels (8 bytes) per pixel, so lane (pixel, k-group q) reads the 16 bytes of window pixels
//           (x + 2q, x + 2q + 1); 7 K steps instead of the 13 of an 8-channel layout.
//   level0  k = tap * 16 + c (5 K steps, the last half zero); maps are stored as two planes of 8 channels (16 bytes per
//           pixel and plane), so the 16 pixel-lanes of a fragment read consecutive 16-byte slots.
//   level1  same, stride 2, two cout tiles paired so a lane stores 8 consecutive couts (16 bytes).
// Regions are walked as flat runs of 16 pixels (pixel index / region width by constant division), so odd region widths
// waste only the tail of the last tile.
#include "common.h"

namespace {
constexpr int T1H = 8, T1W = 16;                   // level1 output tile
constexpr int L0H = 2 * T1H + 1, L0W = 2 * T1W + 1;  // level0 outputs under it (3x3 stride 2 pad 1)
constexpr int STH = L0H + 2, STW = L0W + 2;          // stem outputs under those (3x3 pad 1)
constexpr int INH = STH + 6, INW = STW + 6;          // input pixels under those (7x7 pad 3)
constexpr int NIN = INH * INW, NST = STH * STW, NL0 = L0H * L0W;
constexpr int IN_PAD = 8;                          // zeroed pixels behind the window: tap column 7 of the last row reads them
constexpr int IN_ROUNDS = (NIN + IN_PAD + 255) / 256;
constexpr int K0 = 7 * 32, K1 = 160;               // packed K of the stem / of the two 3x3 layers

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f16x8 lds_read16_align8(const char* p) {
  const u32x2 lo = *(const u32x2*)p, hi = *(const u32x2*)(p + 8);
  const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(f16x8, v);
}

__device__ __forceinline__ f16x4 bn_relu_f16(f32x4 acc, f32x4 sc, f32x4 bi, bool keep) {
  f16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float v = fmaxf(acc[j] * sc[j] + bi[j], 0.f);
    o[j] = keep ? (f16)v : (f16)0.f;
  }
  return o;
}
}  // namespace

template <typename TIn>
__global__ void __launch_bounds__(256, 2) dla_base_fused_kernel(const BaseArgs a) {
  __shared__ __attribute__((aligned(16))) char inb[(NIN + IN_PAD) * 8];   // [pixel][4 ch] normalised input window
  __shared__ __attribute__((aligned(16))) char stb[2 * NST * 16];         // [plane][pixel][8 ch] stem outputs
  __shared__ __attribute__((aligned(16))) char l0b[2 * NL0 * 16];         // [plane][pixel][8 ch] level0 outputs

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, q = lane >> 4;
  const int H1 = a.Hp >> 1, W1 = a.Wp >> 1;
  const int tiles_x = W1 / T1W, tiles_y = H1 / T1H;
  const int ox = (blockIdx.x % tiles_x) * T1W;
  const int oy = ((blockIdx.x / tiles_x) % tiles_y) * T1H;
  const int b = blockIdx.x / (tiles_x * tiles_y);

  // ---- input window: (x / 255 - mean) / std in f32, the reference's operation order, rounded to f16; zero outside the image
  // (the stem's padding and the batch padding up to Hp x Wp) ----
  {
    const TIn* img = (const TIn*)a.img + (long)b * a.img_batch_stride;
    const long plane = (long)a.H * a.W;
    const int y0 = 2 * oy - 5, x0 = 2 * ox - 5;
    float v[IN_ROUNDS][3];
#pragma unroll
    for (int i = 0; i < IN_ROUNDS; ++i) {
      const int pid = tid + 256 * i;
      const int wr = pid / INW, wc = pid - wr * INW;
      const int Y = y0 + wr, X = x0 + wc;
      v[i][0] = v[i][1] = v[i][2] = 0.f;
      if (pid < NIN && Y >= 0 && Y < a.H && X >= 0 && X < a.W) {
        const TIn* p = img + (long)Y * a.W + X;
        v[i][0] = ((float)p[0] / 255.f - a.mean[0]) / a.stdv[0];
        v[i][1] = ((float)p[plane] / 255.f - a.mean[1]) / a.stdv[1];
        v[i][2] = ((float)p[2 * plane] / 255.f - a.mean[2]) / a.stdv[2];
      }
    }
#pragma unroll
    for (int i = 0; i < IN_ROUNDS; ++i) {
      const int pid = tid + 256 * i;
      if (pid < NIN + IN_PAD) {
        const f16x4 o = {(f16)v[i][0], (f16)v[i][1], (f16)v[i][2], (f16)0.f};
        *(f16x4*)(inb + pid * 8) = o;
      }
    }
  }
  __syncthreads();

  // ---- stem: 7x7, 3 -> 16 ----
  {
    f16x8 wf[7];
    const f16* wr = (const f16*)a.w0 + fr * K0 + q * 8;
#pragma unroll
    for (int r = 0; r < 7; ++r) wf[r] = *(const f16x8*)(wr + r * 32);
    const f32x4 sc = *(const f32x4*)(a.s0 + q * 4), bi = *(const f32x4*)(a.b0 + q * 4);
    const int sy0 = 2 * oy - 2, sx0 = 2 * ox - 2;
    for (int t = wave; t < (NST + 15) / 16; t += 4) {
      const int p = t * 16 + fr;
      const int pc = p < NST ? p : NST - 1;
      const int py = pc / STW, px = pc - py * STW;
      const char* base = inb + (py * INW + px) * 8 + q * 16;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 7; ++r)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[r], lds_read16_align8(base + r * INW * 8), acc, 0, 0, 0);
      const int Y = sy0 + py, X = sx0 + px;
      const bool inside = Y >= 0 && Y < a.Hp && X >= 0 && X < a.Wp;   // outside: level0's zero padding
      const f16x4 o = bn_relu_f16(acc, sc, bi, inside);
      if (p < NST) *(f16x4*)(stb + (q >> 1) * (NST * 16) + p * 16 + (q & 1) * 8) = o;
    }
  }
  __syncthreads();

  // ---- level0: 3x3, 16 -> 16 ----
  {
    f16x8 wf[5];
    const f16* wr = (const f16*)a.w1 + fr * K1 + q * 8;
    int kaddr[5];
#pragma unroll
    for (int kt = 0; kt < 5; ++kt) {
      wf[kt] = *(const f16x8*)(wr + kt * 32);
      const int G = kt * 4 + q;
      const int tap = (G >> 1) < 9 ? (G >> 1) : 0;     // K tail: zero weights, read tap 0
      const int tr = tap / 3, ts = tap - tr * 3;
      kaddr[kt] = (G & 1) * (NST * 16) + (tr * STW + ts) * 16;
    }
    const f32x4 sc = *(const f32x4*)(a.s1 + q * 4), bi = *(const f32x4*)(a.b1 + q * 4);
    const int ly0 = 2 * oy - 1, lx0 = 2 * ox - 1;
    for (int t = wave; t < (NL0 + 15) / 16; t += 4) {
      const int p = t * 16 + fr;
      const int pc = p < NL0 ? p : NL0 - 1;
      const int py = pc / L0W, px = pc - py * L0W;
      const char* base = stb + (py * STW + px) * 16;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < 5; ++kt)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kt], *(const f16x8*)(base + kaddr[kt]), acc, 0, 0, 0);
      const int Y = ly0 + py, X = lx0 + px;
      const bool inside = Y >= 0 && Y < a.Hp && X >= 0 && X < a.Wp;   // outside: level1's zero padding
      const f16x4 o = bn_relu_f16(acc, sc, bi, inside);
      if (p < NL0) *(f16x4*)(l0b + (q >> 1) * (NL0 * 16) + p * 16 + (q & 1) * 8) = o;
    }
  }
  __syncthreads();

  // ---- level1: 3x3 stride 2, 16 -> 32; lane = (pixel column fr, 8 consecutive couts q*8..q*8+7) ----
  {
    f16x8 wf[5][2];
    int kaddr[5];
#pragma unroll
    for (int kt = 0; kt < 5; ++kt) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int cout = (fr >> 2) * 8 + c * 4 + (fr & 3);   // tile pair layout: tile c row r <-> cout (r/4)*8 + c*4 + r%4
        wf[kt][c] = *(const f16x8*)((const f16*)a.w2 + cout * K1 + kt * 32 + q * 8);
      }
      const int G = kt * 4 + q;
      const int tap = (G >> 1) < 9 ? (G >> 1) : 0;
      const int tr = tap / 3, ts = tap - tr * 3;
      kaddr[kt] = (G & 1) * (NL0 * 16) + (tr * L0W + ts + 2 * fr) * 16;
    }
    const f32x4 sc0 = *(const f32x4*)(a.s2 + q * 8), sc1 = *(const f32x4*)(a.s2 + q * 8 + 4);
    const f32x4 bi0 = *(const f32x4*)(a.b2 + q * 8), bi1 = *(const f32x4*)(a.b2 + q * 8 + 4);
    for (int row = wave; row < T1H; row += 4) {
      const char* base = l0b + (2 * row * L0W) * 16;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < 5; ++kt) {
        const f16x8 pf = *(const f16x8*)(base + kaddr[kt]);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kt][0], pf, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kt][1], pf, acc1, 0, 0, 0);
      }
      const f16x4 o0 = bn_relu_f16(acc0, sc0, bi0, true), o1 = bn_relu_f16(acc1, sc1, bi1, true);
      const f16x8 o = {o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o1[2], o1[3]};
      f16* yp = (f16*)a.y + ((long)(b * H1 + oy + row) * W1 + ox + fr) * a.out_stride + q * 8;
      *(f16x8*)yp = o;
    }
  }
}

int launch_dla_base(const BaseArgs& a, hipStream_t s) {
  CTDET_CHECK(a.Hp % (2 * T1H) == 0 && a.Wp % (2 * T1W) == 0, "dla_base: padded size %dx%d must be a multiple of %dx%d",
              a.Hp, a.Wp, 2 * T1H, 2 * T1W);
  CTDET_CHECK(a.H <= a.Hp && a.W <= a.Wp && a.H > 0 && a.W > 0, "dla_base: image %dx%d larger than padded %dx%d", a.H, a.W,
              a.Hp, a.Wp);
  CTDET_CHECK(a.out_stride >= 32 && a.out_stride % 8 == 0 && (((size_t)a.y) & 15) == 0, "dla_base: output rows must be 16-byte aligned");
  const long blocks = (long)a.B * (a.Hp / (2 * T1H)) * (a.Wp / (2 * T1W));
  if (blocks == 0) return 0;
  CTDET_CHECK(blocks < (1L << 31), "dla_base: too many tiles");
  if (a.img_dtype == CTDET_U8)
    hipLaunchKernelGGL((dla_base_fused_kernel<uint8_t>), dim3((unsigned)blocks), dim3(256), 0, s, a);
  else if (a.img_dtype == CTDET_F32)
    hipLaunchKernelGGL((dla_base_fused_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, s, a);
  else
    CTDET_CHECK(false, "dla_base: image dtype %d (want u8 or f32)", a.img_dtype);
  CTDET_LAUNCH_CHECK();
  return 0;
}
