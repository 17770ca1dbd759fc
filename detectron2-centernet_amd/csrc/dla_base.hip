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
#include <type_traits>

namespace {
constexpr int T1H = 8, T1W = 16;                   // level1 output tile
constexpr int L0H = 2 * T1H + 1, L0W = 2 * T1W + 1;  // level0 outputs under it (3x3 stride 2 pad 1)
constexpr int STH = L0H + 2, STW = L0W + 2;          // stem outputs under those (3x3 pad 1)
constexpr int INH = STH + 6, INW = STW + 6;          // input pixels under those (7x7 pad 3)
constexpr int NIN = INH * INW, NST = STH * STW, NL0 = L0H * L0W;
constexpr int IN_PAD = 8;                          // zeroed pixels behind the window: tap column 7 of the last row reads them
constexpr int IN_ROUNDS = (NIN + IN_PAD + 255) / 256;
constexpr int K0 = 7 * 32, K1 = 160;
constexpr int STP = (NST * 16 + 255) / 256 * 256, L0P = (NL0 * 16 + 255) / 256 * 256;   // plane strides: multiples of 256 B, so
                                                                // the two planes of a pixel sit on the same LDS banks               // packed K of the stem / of the two 3x3 layers

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f16x8 lds_read16_align8(const char* p) {
  const u32x2 lo = *(const u32x2*)p, hi = *(const u32x2*)(p + 8);
  const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(f16x8, v);
}

typedef f16 f16x2 __attribute__((ext_vector_type(2)));

// relu(acc * scale + bias) -> 4 f16 (round to nearest even like every other f16 store on the path, packed max after the
// conversion); MASKED: zero when the pixel lies outside the map
template <bool MASKED>
__device__ __forceinline__ f16x4 bn_relu_f16(f32x4 acc, f32x4 sc, f32x4 bi, bool keep) {
  const f32x4 v = acc * sc + bi;
  f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
  const f16x4 z = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
  o = __builtin_elementwise_max(o, z);
  if constexpr (MASKED) o = keep ? o : z;
  return o;
}

// 4 bytes at any byte address
__device__ __forceinline__ unsigned load_u32_unaligned(const void* p) {
  typedef unsigned u32_a1 __attribute__((aligned(1)));
  return *(const u32_a1*)p;
}
}  // namespace

template <typename TIn>
__global__ void __launch_bounds__(256, 2) dla_base_fused_kernel(const BaseArgs a, int ntiles) {
  __shared__ __attribute__((aligned(16))) char inb[(NIN + IN_PAD) * 8];   // [pixel][4 ch] normalised input window
  __shared__ __attribute__((aligned(16))) char stb[2 * STP];              // [plane][pixel][8 ch] stem outputs
  __shared__ __attribute__((aligned(16))) char l0b[2 * L0P];              // [plane][pixel][8 ch] level0 outputs
  __shared__ f16 lut[3 * 256];                                            // uint8 images: normalised value per (channel, byte)
  __shared__ __attribute__((aligned(16))) float sbv[128];                 // scale/bias: stem 0/16, level0 32/48, level1 64/96

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, q = lane >> 4;
  const int H1 = a.Hp >> 1, W1 = a.Wp >> 1;
  const int tiles_x = W1 / T1W, tiles_y = H1 / T1H;
  const long plane = (long)a.H * a.W;
  constexpr bool BYTES = sizeof(TIn) == 1;

  // (x / 255 - mean) / std in f32, the reference's operation order, rounded to f16.  A byte has 256 values: tabulate them
  // once per workgroup instead of two f32 divisions per sample.
  if constexpr (BYTES) {
#pragma unroll
    for (int c = 0; c < 3; ++c) lut[c * 256 + tid] = (f16)(((float)tid / 255.f - a.mean[c]) / a.stdv[c]);
  }
  // the window buffer's 4th channel and the pixels behind the window stay zero (the fast conversion path writes channels 0-2)
  for (int i = tid; i < (NIN + IN_PAD) * 2; i += 256) ((unsigned*)inb)[i] = 0u;

  // ---- the three layers' weight fragments, LDS tap offsets and BatchNorm vectors stay in registers across tiles ----
  f16x8 wf0[7], wf1[5], wf2[5][2];
  int kaddr1[5], kaddr2[5];
#pragma unroll
  for (int r = 0; r < 7; ++r) wf0[r] = *(const f16x8*)((const f16*)a.w0 + fr * K0 + r * 32 + q * 8);
#pragma unroll
  for (int kt = 0; kt < 5; ++kt) {
    wf1[kt] = *(const f16x8*)((const f16*)a.w1 + fr * K1 + kt * 32 + q * 8);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int cout = (fr >> 2) * 8 + c * 4 + (fr & 3);   // tile pair layout: tile c row r <-> cout (r/4)*8 + c*4 + r%4
      wf2[kt][c] = *(const f16x8*)((const f16*)a.w2 + cout * K1 + kt * 32 + q * 8);
    }
    const int G = kt * 4 + q;
    const int tap = (G >> 1) < 9 ? (G >> 1) : 0;           // K tail: zero weights, read tap 0
    const int tr = tap / 3, ts = tap - tr * 3;
    kaddr1[kt] = (G & 1) * STP + (tr * STW + ts) * 16;
    kaddr2[kt] = (G & 1) * L0P + (tr * L0W + ts + 2 * fr) * 16;
  }
  // folded BatchNorm vectors: LDS (read back 16 bytes at a time in the epilogues; 32 VGPRs otherwise)
  if (tid < 16) { sbv[tid] = a.s0[tid]; sbv[16 + tid] = a.b0[tid]; sbv[32 + tid] = a.s1[tid]; sbv[48 + tid] = a.b1[tid]; }
  if (tid < 32) { sbv[64 + tid] = a.s2[tid]; sbv[96 + tid] = a.b2[tid]; }

  // level0 walks its 17 x 33 region as flat runs of 16 pixels, 9 runs per wave: the lane's pixel in each run, as LDS byte
  // offsets (read side: position in the 35-wide stem map; write side: position in the 33-wide level0 map; 0xffff = beyond
  // the region), packed 2 x 16 bits.  Fixed for the whole launch.
  unsigned l0pos[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int p = (wave * 9 + t) * 16 + fr;
    const int pc = p < NL0 ? p : NL0 - 1;
    const int py = pc / L0W, px = pc - py * L0W;
    l0pos[t] = (unsigned)((py * STW + px) * 16) | ((p < NL0 ? (unsigned)(p * 16) : 0xffffu) << 16);
  }

  // ---- window fetch for the NEXT tile, issued while the current tile computes.  Nothing touches the registers before the
  // conversion one tile later (a conditional load or an early pack would put an s_waitcnt vmcnt right behind each load).
  // Interior tiles of byte images: a thread loads 4 consecutive bytes of one plane row (825 dwords per window, 4 per
  // thread).  Other tiles, and f32 images: one sample per load at clamped coordinates, validity kept as a bit mask. ----
  constexpr int FAST_PER_ROW = (INW + 3) / 4;                    // 11 dwords cover the 41 pixels of a window row
  constexpr int FAST_N = 3 * INH * FAST_PER_ROW;                 // 825
  constexpr int FAST_ROUNDS = (FAST_N + 255) / 256;              // 4
  // (byte images, border tiles: the three channels of a pixel packed into one register -- that path waits for its loads)
  constexpr int raw1C = BYTES ? 1 : 3;
  unsigned raw[IN_ROUNDS * raw1C];
  unsigned okmask = 0;
  bool raw_fast = false;
  auto tile_origin = [&](int tile, int& ox, int& oy, int& b) {
    ox = (tile % tiles_x) * T1W;
    oy = ((tile / tiles_x) % tiles_y) * T1H;
    b = tile / (tiles_x * tiles_y);
  };
  auto fetch = [&](int tile) {
    int ox, oy, b;
    tile_origin(tile, ox, oy, b);
    const TIn* img = (const TIn*)a.img + (long)b * a.img_batch_stride;
    const int y0 = 2 * oy - 5, x0 = 2 * ox - 5;
    // every byte the fast path touches lies inside the image rows (it reads 3 bytes past the window's last column)
    raw_fast = BYTES && y0 >= 0 && y0 + INH <= a.H && x0 >= 0 && x0 + 4 * FAST_PER_ROW <= a.W;
    if (raw_fast) {
#pragma unroll
      for (int i = 0; i < FAST_ROUNDS; ++i) {
        const int e = min(tid + 256 * i, FAST_N - 1);
        const int c = e / (INH * FAST_PER_ROW), rem = e - c * (INH * FAST_PER_ROW);
        const int wr = rem / FAST_PER_ROW, g = rem - wr * FAST_PER_ROW;
        raw[i] = load_u32_unaligned((const char*)img + c * plane + (long)(y0 + wr) * a.W + x0 + 4 * g);
      }
    } else {
      okmask = 0;
#pragma unroll
      for (int i = 0; i < IN_ROUNDS; ++i) {
        const int pid = tid + 256 * i;
        const int wr = pid / INW, wc = pid - wr * INW;
        const int Y = y0 + wr, X = x0 + wc;
        const bool ok = pid < NIN && Y >= 0 && Y < a.H && X >= 0 && X < a.W;
        const int Yc = Y < 0 ? 0 : (Y >= a.H ? a.H - 1 : Y), Xc = X < 0 ? 0 : (X >= a.W ? a.W - 1 : X);
        const TIn* p = img + (long)Yc * a.W + Xc;
        if constexpr (BYTES) {
          raw[i] = (unsigned)p[0] | ((unsigned)p[plane] << 8) | ((unsigned)p[2 * plane] << 16);
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c) raw[i * 3 + c] = __builtin_bit_cast(unsigned, p[c * plane]);
        }
        okmask |= ok ? 1u << i : 0u;
      }
    }
  };
  auto convert = [&]() {
    if (raw_fast) {
#pragma unroll
      for (int i = 0; i < FAST_ROUNDS; ++i) {
        const int e = tid + 256 * i;
        const int c = e / (INH * FAST_PER_ROW), rem = e - c * (INH * FAST_PER_ROW);
        const int wr = rem / FAST_PER_ROW, g = rem - wr * FAST_PER_ROW;
        char* dst = inb + (wr * INW + 4 * g) * 8 + c * 2;
        const int npx = e < FAST_N ? (g == FAST_PER_ROW - 1 ? INW - 4 * (FAST_PER_ROW - 1) : 4) : 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k < npx) *(f16*)(dst + k * 8) = lut[c * 256 + ((raw[i] >> (8 * k)) & 255u)];
      }
    } else {
#pragma unroll
      for (int i = 0; i < IN_ROUNDS; ++i) {
        const int pid = tid + 256 * i;
        if (pid < NIN) {
          const bool ok = okmask & (1u << i);
          f16x4 o = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            f16 v;
            if constexpr (BYTES) v = lut[c * 256 + ((raw[i] >> (8 * c)) & 255u)];
            else v = (f16)((__builtin_bit_cast(float, raw[i * 3 + c]) / 255.f - a.mean[c]) / a.stdv[c]);
            o[c] = ok ? v : (f16)0.f;
          }
          *(f16x4*)(inb + pid * 8) = o;
        }
      }
    }
  };

  // ---- the three conv phases of one tile; MASKED: the tile touches the border of the Hp x Wp map, so stem / level0 outputs
  // outside it are forced to zero (they are the next layer's zero padding) ----
  auto compute = [&](auto masked_c, int ox, int oy, int b) {
    constexpr bool MASKED = decltype(masked_c)::value;
    // stem: 7x7, 3 -> 16.  Wave w owns stem rows [5w, 5w+5) (the last wave 4) in three 16-column strips (the third overlaps
    // the second: columns 19..34; both write the same values).  A strip's 11 window rows are read once -- output row i,
    // kernel row r uses fragment i + r -- and its five accumulator chains are independent.
    {
      const int sy0 = 2 * oy - 2, sx0 = 2 * ox - 2;
      const int r0 = wave * 5;
      const bool five = wave < 3;
      const f32x4 sc = *(const f32x4*)(sbv + q * 4), bi = *(const f32x4*)(sbv + 16 + q * 4);
#pragma unroll
      for (int strip = 0; strip < 3; ++strip) {
        const int px = (strip == 0 ? 0 : strip == 1 ? 16 : STW - 16) + fr;
        const char* base = inb + (r0 * INW + px) * 8 + q * 16;
        f16x8 f[11];
#pragma unroll
        for (int j = 0; j < 10; ++j) f[j] = lds_read16_align8(base + j * INW * 8);
        f[10] = lds_read16_align8(base + (five ? 10 : 9) * INW * 8);
        f32x4 acc[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 7; ++r)
#pragma unroll
          for (int i = 0; i < 5; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0[r], f[i + r], acc[i], 0, 0, 0);
        const int X = sx0 + px;
        const bool xin = X >= 0 && X < a.Wp;
        char* wbase = stb + (q >> 1) * STP + (r0 * STW + px) * 16 + (q & 1) * 8;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          const int Y = sy0 + r0 + i;
          const f16x4 o = bn_relu_f16<MASKED>(acc[i], sc, bi, xin && Y >= 0 && Y < a.Hp);
          if (i < 4 || five) *(f16x4*)(wbase + i * STW * 16) = o;   // the last wave has four rows
        }
      }
    }
    __syncthreads();

    // level0: 3x3, 16 -> 16: 36 flat runs of 16 pixels, 9 per wave, three at a time (fifteen fragment reads in flight, three
    // independent accumulator chains)
    {
      const int ly0 = 2 * oy - 1, lx0 = 2 * ox - 1;
      const f32x4 sc = *(const f32x4*)(sbv + 32 + q * 4), bi = *(const f32x4*)(sbv + 48 + q * 4);
#pragma unroll
      for (int it = 0; it < 3; ++it) {
        f16x8 f[3][5];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const char* base = stb + (l0pos[it * 3 + u] & 0xffffu);
#pragma unroll
          for (int kt = 0; kt < 5; ++kt) f[u][kt] = *(const f16x8*)(base + kaddr1[kt]);
        }
        f32x4 acc[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 5; ++kt)
#pragma unroll
          for (int u = 0; u < 3; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1[kt], f[u][kt], acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const unsigned wofs = l0pos[it * 3 + u] >> 16;
          bool inside = true;
          if constexpr (MASKED) {
            const int p = (int)(wofs >> 4), py = p / L0W, px = p - py * L0W;
            const int Y = ly0 + py, X = lx0 + px;
            inside = Y >= 0 && Y < a.Hp && X >= 0 && X < a.Wp;
          }
          const f16x4 o = bn_relu_f16<MASKED>(acc[u], sc, bi, inside);
          if (wofs != 0xffffu) *(f16x4*)(l0b + (q >> 1) * L0P + wofs + (q & 1) * 8) = o;
        }
      }
    }
    __syncthreads();

    // level1: 3x3 stride 2, 16 -> 32; lane = (pixel column fr, 8 consecutive couts q*8..q*8+7); rows 2*wave and 2*wave+1,
    // so the 2x2 max-pool of the tile (Tree.downsample of level2, dla.py:128-129) is one packed max across the wave's two
    // rows and one with the neighbouring lane
    {
      f16x8 pf[2][5];
#pragma unroll
      for (int rw = 0; rw < 2; ++rw)
#pragma unroll
        for (int kt = 0; kt < 5; ++kt) pf[rw][kt] = *(const f16x8*)(l0b + (2 * (2 * wave + rw) * L0W) * 16 + kaddr2[kt]);
      f32x4 acc[2][2];
#pragma unroll
      for (int rw = 0; rw < 2; ++rw) acc[rw][0] = acc[rw][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < 5; ++kt)
#pragma unroll
        for (int rw = 0; rw < 2; ++rw)
#pragma unroll
          for (int c = 0; c < 2; ++c)
            acc[rw][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf2[kt][c], pf[rw][kt], acc[rw][c], 0, 0, 0);
      f16* yp = (f16*)a.y + ((long)(b * H1 + oy + 2 * wave) * W1 + ox + fr) * a.out_stride + q * 8;
      f16x8 o[2];
#pragma unroll
      for (int rw = 0; rw < 2; ++rw) {
        const f16x4 o0 = bn_relu_f16<false>(acc[rw][0], *(const f32x4*)(sbv + 64 + q * 8), *(const f32x4*)(sbv + 96 + q * 8), true);
        const f16x4 o1 = bn_relu_f16<false>(acc[rw][1], *(const f32x4*)(sbv + 68 + q * 8), *(const f32x4*)(sbv + 100 + q * 8), true);
        o[rw] = (f16x8){o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o1[2], o1[3]};
        *(f16x8*)(yp + (long)rw * W1 * a.out_stride) = o[rw];
      }
      if (a.pool) {
        const f16x8 v = __builtin_elementwise_max(o[0], o[1]);
        u32x4 vb = __builtin_bit_cast(u32x4, v), nb;
#pragma unroll
        for (int e = 0; e < 4; ++e)   // the pixel column fr ^ 1: quad_perm [1,0,3,2]
          nb[e] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)vb[e], 0xB1, 0xF, 0xF, true);
        const f16x8 m = __builtin_elementwise_max(v, __builtin_bit_cast(f16x8, nb));
        if (!(fr & 1))
          *(f16x8*)((f16*)a.pool + ((long)(b * (H1 >> 1) + (oy >> 1) + wave) * (W1 >> 1) + ((ox + fr) >> 1)) * a.pool_stride +
                    q * 8) = m;
      }
    }
  };

  // weights have landed before the first window fetch is issued: the waitcnt pass then never waits for them inside the tile
  // loop (vmcnt counts in order, so a wait for a weight register would also drain the younger window loads)
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  __syncthreads();                                   // lut, sbv, cleared inb
  for (; tile < ntiles; tile += gridDim.x) {
    int ox, oy, b;
    tile_origin(tile, ox, oy, b);
    convert();                                       // inb was last read in the previous tile's stem phase, two barriers ago
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
    // stem outputs of this tile span rows 2oy-2 .. 2oy+16 and columns 2ox-2 .. 2ox+32 of the Hp x Wp map
    const bool interior = oy > 0 && 2 * oy + 16 < a.Hp && ox > 0 && 2 * ox + 32 < a.Wp;
    if (interior) compute(std::false_type{}, ox, oy, b);
    else compute(std::true_type{}, ox, oy, b);
  }
}

int launch_dla_base(const BaseArgs& a, hipStream_t s) {
  CTDET_CHECK(a.Hp % (2 * T1H) == 0 && a.Wp % (2 * T1W) == 0, "dla_base: padded size %dx%d must be a multiple of %dx%d",
              a.Hp, a.Wp, 2 * T1H, 2 * T1W);
  CTDET_CHECK(a.H <= a.Hp && a.W <= a.Wp && a.H > 0 && a.W > 0, "dla_base: image %dx%d larger than padded %dx%d", a.H, a.W,
              a.Hp, a.Wp);
  CTDET_CHECK(a.out_stride >= 32 && a.out_stride % 8 == 0 && (((size_t)a.y) & 15) == 0, "dla_base: output rows must be 16-byte aligned");
  CTDET_CHECK(!a.pool || (a.pool_stride >= 32 && a.pool_stride % 8 == 0 && (((size_t)a.pool) & 15) == 0),
              "dla_base: pooled output rows must be 16-byte aligned");
  const long tiles = (long)a.B * (a.Hp / (2 * T1H)) * (a.Wp / (2 * T1W));
  if (tiles == 0) return 0;
  CTDET_CHECK(tiles < (1L << 31), "dla_base: too many tiles");
  // persistent workgroups: 2 per CU, each walks tiles blockIdx.x, + gridDim.x, ... with its weights in
  // registers and the next tile's image window in flight
  const int ncu = ctdet_device_cu_count();
  const long want = 2L * ncu;
  const unsigned blocks = (unsigned)(tiles < want ? tiles : want);
  if (a.img_dtype == CTDET_U8)
    hipLaunchKernelGGL((dla_base_fused_kernel<uint8_t>), dim3(blocks), dim3(256), 0, s, a, (int)tiles);
  else if (a.img_dtype == CTDET_F32)
    hipLaunchKernelGGL((dla_base_fused_kernel<float>), dim3(blocks), dim3(256), 0, s, a, (int)tiles);
  else
    CTDET_CHECK(false, "dla_base: image dtype %d (want u8 or f32)", a.img_dtype);
  CTDET_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// f16x3 form of the fused base (f32 tensors, every product as hi*hi + lo*hi + hi*lo on the f16 matrix pipe, f32 accumulate:
// the arithmetic of conv_f32_win_kernel<..., SP>, which runs these three layers one by one in that mode -- 0.63 + 0.70 +
// 0.48 ms at batch 64 x 512^2, each a 1-2 GB f32 stream, plus 0.2 ms of preprocessing and pooling).  Same tile and
// receptive-field walk as the f16 kernel above (8x16 level1 outputs <- 17x33 level0 <- 19x35 stem <- 25x41 input pixels), one
// tile per workgroup, two workgroups per CU:
//   * the window is normalised in f32 with the reference's operation order ((x / 255 - mean) / std; byte images through a
//     3 x 256 table built per workgroup in the stem planes' bytes, free until the stem phase) and stored split, 16 bytes per
//     pixel = {hi[4], lo[4]} (channel 3 zero);
//   * the stem's k order is the f16 kernel's (kernel row, 8 taps of which the eighth has zero weights, 4 channels): a K step is
//     four consecutive taps of a row, lane group q takes tap 4 half + q, so every LDS address of the kernel is
//     (per-run lane base) + (compile-time offset) -- no vector address arithmetic inside the K loops;
//   * a fragment read ({hi[4], lo[4]} of one tap / 4-channel group) IS the B operand of both products of the generic scheme
//     (conv_f32_win_kernel): wa = {w_lo, w_hi}, wb = {w_hi, 0} from the {w_hi[4], w_lo[4]} groups of ctdet_pack_weights_x3
//     (layout 0, k = tap * channels + c, rows scaled by a power of two that the folded BatchNorm scale carries back);
//   * stem and level0 outputs (f32 after scale / bias / ReLU, zero outside the map) are split once, when they are written to
//     LDS: four planes of 4 channels, 16 bytes per pixel and plane, so the 16 pixel-lanes of a fragment read consecutive
//     slots.  The level0 planes reuse the bytes of the input window (dead after the stem phase): 42.8 + 36.1 KB per workgroup;
//   * a phase's weight groups are fetched from L2 one phase ahead of their use (all three sets do not fit in 256 registers).
// HBM sees the image once (50 MB of bytes at batch 64) and the level1 map (+ its 2x2 max-pool) once (0.54 + 0.13 GB).
// Measured alternatives (profiles/r04_base_x3_ablation.txt): persistent workgroups with the next window prefetched (no gain:
// the time is in the three K-loop phases, not in the window stage), all weights resident in registers with the second operand
// assembled next to its MFMA (spills; slower).
// ------------------------------------------------------------------------------------------
namespace {
constexpr int X3_STP = (NST * 16 + 63) / 64 * 64, X3_L0P = (NL0 * 16 + 63) / 64 * 64;    // plane strides (bytes)
constexpr int X3_NK0 = 14;                          // stem K steps: 7 kernel rows x 8 taps (the eighth zero) x 4 channels = 224
static_assert((NIN + IN_PAD) * 16 <= 4 * X3_L0P, "the input window fits under the level0 planes");
static_assert(3 * 256 * 4 <= 4 * X3_STP, "the byte table fits under the stem planes");
static_assert(4 * X3_STP + 4 * X3_L0P <= 81920, "two workgroups per CU");

__device__ __forceinline__ f16x8 split4(const f32x4 v) {
  const f16x4 hi = __builtin_convertvector(v, f16x4);
  f32x4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = v[j] - (float)hi[j];
  const f16x4 lo = __builtin_convertvector(r, f16x4);
  return __builtin_shufflevector(hi, lo, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ f32x4 bn_relu_f32(const f32x4 acc, const f32x4 sc, const f32x4 bi) {
  f32x4 v = acc * sc;
  v = v + bi;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
  return v;
}
// one phase's raw weight groups of a lane ({w_hi[4], w_lo[4]} each), fetched ahead of the phase that uses them
template <int NKT>
__device__ __forceinline__ void x3_fetch(const void* w, int row, int kpad, int q, f16x8 (&raw)[NKT]) {
#if defined(CTDET_BASE_ABLATE) && (CTDET_BASE_ABLATE & 64)
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) raw[kt] = (f16x8){(f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f};
  return;
#endif
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) raw[kt] = *(const f16x8*)((const char*)w + ((long)row * kpad + kt * 16 + q * 4) * 4);
}
// the two A operands of the generic f16x3 scheme from the packed groups: wa = {w_lo, w_hi}, wb = {w_hi, 0}
template <int NKT>
__device__ __forceinline__ void x3_operands(const f16x8 (&raw)[NKT], f16x8 (&wa)[NKT], f16x8 (&wb)[NKT]) {
  const f16x4 z4 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    wa[kt] = __builtin_shufflevector(raw[kt], raw[kt], 4, 5, 6, 7, 0, 1, 2, 3);
    wb[kt] = __builtin_shufflevector(__builtin_shufflevector(raw[kt], raw[kt], 0, 1, 2, 3), z4, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}
}  // namespace

// measurement builds only (tools/ablate_base.sh): CTDET_BASE_ABLATE == 1 drops the fragment reads of the K loops (one read per
// run instead), == 2 drops their MFMAs; bits 4 / 8 / 16 / 32 / 64 skip the window conversion / the stem loop / the level0 loop /
// the level1 phase / the weight fetches
#ifndef CTDET_BASE_ABLATE
#define CTDET_BASE_ABLATE 0
#endif
#if CTDET_BASE_ABLATE == 1
#define X3_FRAG(ptr, off) (*(const f16x8*)(ptr))
#else
#define X3_FRAG(ptr, off) (*(const f16x8*)((ptr) + (off)))
#endif
__device__ __forceinline__ f32x4 x3_mma(const f16x8 w, const f16x8 h, f32x4 acc) {
#if CTDET_BASE_ABLATE == 2
  acc[0] += (float)h[0] * (float)w[0];
  return acc;
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(w, h, acc, 0, 0, 0);
#endif
}

template <typename TIn>
__global__ void __launch_bounds__(256, 2) dla_base_x3_kernel(const BaseArgs a, int ntiles) {
  __shared__ __attribute__((aligned(16))) char stb[4 * X3_STP];    // [plane][pixel]{hi[4], lo[4]} stem outputs
  __shared__ __attribute__((aligned(16))) char l0b[4 * X3_L0P];    // the input window, then the level0 outputs
  char* const inb = l0b;
  constexpr bool BYTES = sizeof(TIn) == 1;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, q = lane >> 4;
  const int H1 = a.Hp >> 1, W1 = a.Wp >> 1;
  const int tiles_x = W1 / T1W, tiles_y = H1 / T1H;
  // consecutive workgroups go to different XCDs: give each XCD a contiguous run of tiles (neighbours share window pixels in L2)
  int tile = blockIdx.x;
  if ((ntiles & 7) == 0) tile = (tile & 7) * (ntiles >> 3) + (tile >> 3);
  const int ox = (tile % tiles_x) * T1W;
  const int oy = ((tile / tiles_x) % tiles_y) * T1H;
  const int b = tile / (tiles_x * tiles_y);

  f16x8 raw0[X3_NK0];
  x3_fetch<X3_NK0>(a.w0, fr, X3_NK0 * 16, q, raw0);                // in flight during the window stage

  // ---- input window: rows 2oy-5 .. 2oy+19, columns 2ox-5 .. 2ox+35 of the image; zero outside it (also inside the padded
  // map: the reference pads the NORMALISED batch with zeros, centernet.py:193-200 + ImageList) ----
  {
    const TIn* img = (const TIn*)a.img + (long)b * a.img_batch_stride;
    const long plane = (long)a.H * a.W;
    const int iy0 = 2 * oy - 5, ix0 = 2 * ox - 5;
    float* lut = (float*)stb;
    constexpr int ROUNDS = (NIN + 255) / 256;
    // tap column 7 (zero weights) of the window's last row reads the pixels behind it: finite values wanted
    if (tid < IN_PAD) *(f16x8*)(inb + (NIN + tid) * 16) = split4((f32x4){0.f, 0.f, 0.f, 0.f});
    if constexpr (BYTES) {
      unsigned rawpx[ROUNDS];                                      // the three channels of a pixel, 0xffffffff = outside
#pragma unroll
      for (int c = 0; c < 3; ++c) lut[c * 256 + tid] = ((float)tid / 255.f - a.mean[c]) / a.stdv[c];
#pragma unroll
      for (int i = 0; i < ROUNDS; ++i) {
        const int p = tid + 256 * i;
        const int wy = p / INW, wx = p - wy * INW;
        const int y = iy0 + wy, x = ix0 + wx;
        rawpx[i] = 0xffffffffu;
        if (p < NIN && y >= 0 && y < a.H && x >= 0 && x < a.W) {
          const TIn* s = img + (long)y * a.W + x;
          rawpx[i] = (unsigned)s[0] | ((unsigned)s[plane] << 8) | ((unsigned)s[2 * plane] << 16);
        }
      }
      __syncthreads();
      if (!(CTDET_BASE_ABLATE & 4)) {
#pragma unroll
      for (int i = 0; i < ROUNDS; ++i) {
        const int p = tid + 256 * i;
        if (p < NIN) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (rawpx[i] != 0xffffffffu) {
            v[0] = lut[rawpx[i] & 255]; v[1] = lut[256 + ((rawpx[i] >> 8) & 255)]; v[2] = lut[512 + ((rawpx[i] >> 16) & 255)];
          }
          *(f16x8*)(inb + p * 16) = split4(v);
        }
      }
      }
    } else {
#pragma unroll
      for (int i = 0; i < ROUNDS; ++i) {
        const int p = tid + 256 * i;
        if (p < NIN) {
          const int wy = p / INW, wx = p - wy * INW;
          const int y = iy0 + wy, x = ix0 + wx;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (y >= 0 && y < a.H && x >= 0 && x < a.W) {
            const TIn* s = img + (long)y * a.W + x;
            v[0] = ((float)s[0] / 255.f - a.mean[0]) / a.stdv[0];
            v[1] = ((float)s[plane] / 255.f - a.mean[1]) / a.stdv[1];
            v[2] = ((float)s[2 * plane] / 255.f - a.mean[2]) / a.stdv[2];
          }
          *(f16x8*)(inb + p * 16) = split4(v);
        }
      }
    }
  }

  f16x8 raw1[9];
  x3_fetch<9>(a.w1, fr, 144, q, raw1);                             // level0's weights: in flight during the stem phase

  // ---- stem: 7x7 on 4-channel pixels, 14 K steps (kernel row, half row of 4 taps).  A wave walks its share of the 42 runs of
  // 16 pixels two at a time (two independent accumulator chains), a last odd run alone ----
  {
    f16x8 wa[X3_NK0], wb[X3_NK0];
    x3_operands<X3_NK0>(raw0, wa, wb);
    const f32x4 sc = *(const f32x4*)(a.s0 + 4 * q), bi = *(const f32x4*)(a.b0 + 4 * q);
    __syncthreads();                                               // the window is complete (and the table no longer needed)
    constexpr int RUNS = (NST + 15) / 16;                          // 42: waves 0, 1 take 11, waves 2, 3 take 10
    const int r0 = wave * (RUNS / 4) + (wave < RUNS % 4 ? wave : RUNS % 4), nr = RUNS / 4 + (wave < RUNS % 4 ? 1 : 0);
    auto run = [&](int t, auto pairc) {
      constexpr int NU = decltype(pairc)::value;
      int py[NU], px[NU], pp[NU];
      const char* rp[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int p = (t + u) * 16 + fr;
        pp[u] = p;
        const int pc = p < NST ? p : NST - 1;
        py[u] = pc / STW; px[u] = pc - py[u] * STW;
        rp[u] = inb + (py[u] * INW + px[u] + q) * 16;
      }
      f32x4 acc[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < X3_NK0; ++kt) {
        f16x8 h[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) h[u] = X3_FRAG(rp[u], ((kt >> 1) * INW + (kt & 1) * 4) * 16);
#pragma unroll
        for (int u = 0; u < NU; ++u) acc[u] = x3_mma(wa[kt], h[u], acc[u]);
#pragma unroll
        for (int u = 0; u < NU; ++u) acc[u] = x3_mma(wb[kt], h[u], acc[u]);
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (pp[u] < NST) {
          const int sy = 2 * oy - 2 + py[u], sx = 2 * ox - 2 + px[u];
          f32x4 v = bn_relu_f32(acc[u], sc, bi);
          if (!(sy >= 0 && sy < a.Hp && sx >= 0 && sx < a.Wp)) v = (f32x4){0.f, 0.f, 0.f, 0.f};
          *(f16x8*)(stb + q * X3_STP + pp[u] * 16) = split4(v);
        }
      }
    };
    if (!(CTDET_BASE_ABLATE & 8)) {
      int t = r0;
      for (; t + 1 < r0 + nr; t += 2) run(t, std::integral_constant<int, 2>{});
      if (t < r0 + nr) run(t, std::integral_constant<int, 1>{});
    }
  }

  f16x8 raw2[9];
  x3_fetch<9>(a.w2, fr, 144, q, raw2);                             // level1's first cout tile: in flight during level0
  __syncthreads();                                                 // stem planes complete; the window is dead

  // ---- level0: 3x3, 16 -> 16: a K step = one tap, lane group q reads plane q ----
  {
    f16x8 wa[9], wb[9];
    x3_operands<9>(raw1, wa, wb);
    const f32x4 sc = *(const f32x4*)(a.s1 + 4 * q), bi = *(const f32x4*)(a.b1 + 4 * q);
    constexpr int RUNS = (NL0 + 15) / 16;                          // 36: 9 per wave
    const int r0 = wave * (RUNS / 4) + (wave < RUNS % 4 ? wave : RUNS % 4), nr = RUNS / 4 + (wave < RUNS % 4 ? 1 : 0);
    auto run = [&](int t, auto pairc) {
      constexpr int NU = decltype(pairc)::value;
      int py[NU], px[NU], pp[NU];
      const char* rp[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int p = (t + u) * 16 + fr;
        pp[u] = p;
        const int pc = p < NL0 ? p : NL0 - 1;
        py[u] = pc / L0W; px[u] = pc - py[u] * L0W;
        rp[u] = stb + q * X3_STP + (py[u] * STW + px[u]) * 16;
      }
      f32x4 acc[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < 9; ++kt) {
        f16x8 h[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) h[u] = X3_FRAG(rp[u], ((kt / 3) * STW + kt % 3) * 16);
#pragma unroll
        for (int u = 0; u < NU; ++u) acc[u] = x3_mma(wa[kt], h[u], acc[u]);
#pragma unroll
        for (int u = 0; u < NU; ++u) acc[u] = x3_mma(wb[kt], h[u], acc[u]);
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (pp[u] < NL0) {
          const int ly = 2 * oy - 1 + py[u], lx = 2 * ox - 1 + px[u];
          f32x4 v = bn_relu_f32(acc[u], sc, bi);
          if (!(ly >= 0 && ly < a.Hp && lx >= 0 && lx < a.Wp)) v = (f32x4){0.f, 0.f, 0.f, 0.f};
          *(f16x8*)(l0b + q * X3_L0P + pp[u] * 16) = split4(v);
        }
      }
    };
    if (!(CTDET_BASE_ABLATE & 16)) {
      int t = r0;
      for (; t + 1 < r0 + nr; t += 2) run(t, std::integral_constant<int, 2>{});
      if (t < r0 + nr) run(t, std::integral_constant<int, 1>{});
    }
  }
  f16x8 raw3[9];
  x3_fetch<9>(a.w2, 16 + fr, 144, q, raw3);                        // level1's second cout tile
  __syncthreads();

  // ---- level1: 3x3 stride 2, 16 -> 32: wave w owns tile rows 2w, 2w + 1; the two cout tiles one after the other ----
  {
    const char* rp = l0b + q * X3_L0P + (2 * (2 * wave) * L0W + 2 * fr) * 16;   // level0 row 2r of tile row r = 2 wave
    float* yp = (float*)a.y + ((long)(b * H1 + oy + 2 * wave) * W1 + ox + fr) * a.out_stride + 4 * q;
    auto tile_c = [&](const f16x8 (&raw)[9], int c) {
      f16x8 wa[9], wb[9];
      x3_operands<9>(raw, wa, wb);
      const f32x4 sc = *(const f32x4*)(a.s2 + c * 16 + 4 * q), bi = *(const f32x4*)(a.b2 + c * 16 + 4 * q);
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int kt = 0; kt < 9; ++kt) {
        const f16x8 h0 = X3_FRAG(rp, ((kt / 3) * L0W + kt % 3) * 16);
        const f16x8 h1 = X3_FRAG(rp, ((kt / 3 + 2) * L0W + kt % 3) * 16);
        acc[0] = x3_mma(wa[kt], h0, acc[0]);
        acc[1] = x3_mma(wa[kt], h1, acc[1]);
        acc[0] = x3_mma(wb[kt], h0, acc[0]);
        acc[1] = x3_mma(wb[kt], h1, acc[1]);
      }
      const f32x4 v0 = bn_relu_f32(acc[0], sc, bi), v1 = bn_relu_f32(acc[1], sc, bi);
      *(f32x4*)(yp + c * 16) = v0;
      *(f32x4*)(yp + (long)W1 * a.out_stride + c * 16) = v1;
      if (a.pool) {
        f32x4 m;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = fmaxf(v0[j], v1[j]);
          // the pixel column fr ^ 1: quad_perm [1,0,3,2]
          const float n = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
          m[j] = fmaxf(v, n);
        }
        if (!(fr & 1))
          *(f32x4*)((float*)a.pool + ((long)(b * (H1 >> 1) + (oy >> 1) + wave) * (W1 >> 1) + ((ox + fr) >> 1)) * a.pool_stride +
                    c * 16 + 4 * q) = m;
      }
    };
    if (!(CTDET_BASE_ABLATE & 32)) {
      tile_c(raw2, 0);
      tile_c(raw3, 1);
    }
  }
}

int launch_dla_base_x3(const BaseArgs& a, hipStream_t s) {
  CTDET_CHECK(a.Hp % (2 * T1H) == 0 && a.Wp % (2 * T1W) == 0, "dla_base(f16x3): padded size %dx%d must be a multiple of %dx%d",
              a.Hp, a.Wp, 2 * T1H, 2 * T1W);
  CTDET_CHECK(a.H <= a.Hp && a.W <= a.Wp && a.H > 0 && a.W > 0, "dla_base(f16x3): image %dx%d larger than padded %dx%d", a.H,
              a.W, a.Hp, a.Wp);
  CTDET_CHECK(a.out_stride >= 32 && a.out_stride % 4 == 0 && (((size_t)a.y) & 15) == 0,
              "dla_base(f16x3): output rows must be 16-byte aligned");
  CTDET_CHECK(!a.pool || (a.pool_stride >= 32 && a.pool_stride % 4 == 0 && (((size_t)a.pool) & 15) == 0),
              "dla_base(f16x3): pooled output rows must be 16-byte aligned");
  CTDET_CHECK(((((size_t)a.w0) | ((size_t)a.w1) | ((size_t)a.w2) | ((size_t)a.s0) | ((size_t)a.b0) | ((size_t)a.s1) | ((size_t)a.b1) |
                ((size_t)a.s2) | ((size_t)a.b2)) & 15) == 0, "dla_base(f16x3): operands must be 16-byte aligned");
  const long tiles = (long)a.B * (a.Hp / (2 * T1H)) * (a.Wp / (2 * T1W));
  if (tiles == 0) return 0;
  CTDET_CHECK(tiles < (1L << 31), "dla_base(f16x3): too many tiles");
  if (a.img_dtype == CTDET_U8)
    hipLaunchKernelGGL((dla_base_x3_kernel<uint8_t>), dim3((unsigned)tiles), dim3(256), 0, s, a, (int)tiles);
  else if (a.img_dtype == CTDET_F32)
    hipLaunchKernelGGL((dla_base_x3_kernel<float>), dim3((unsigned)tiles), dim3(256), 0, s, a, (int)tiles);
  else
    CTDET_CHECK(false, "dla_base(f16x3): image dtype %d (want u8 or f32)", a.img_dtype);
  CTDET_LAUNCH_CHECK();
  return 0;
}
