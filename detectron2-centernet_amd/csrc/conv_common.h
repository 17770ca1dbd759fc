// Device helpers shared by the conv-shaped kernels (conv_igemm.hip: f16 MFMA path, conv_f32.hip: f32 MFMA path).
#pragma once
#include "common.h"
#include <type_traits>
// slot permutation for 64-byte LDS rows read as 16-row fragments by ds_read_b128:
// rows r and r+4 share banks, so the 4 rows {r, r+4, r+8, r+12} get distinct slot XORs.
__device__ __forceinline__ int swz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

// XCD-aware tile mapping (1-D grid of 8*mchunk*nby workgroups).  Workgroups are dealt round-robin over the 8
// XCDs, each with a private L2: XCD x gets the contiguous pixel-tile range [x*mchunk, (x+1)*mchunk) and walks it
// with the cout tile innermost, so workgroups that share an input tile (other cout tiles) or halo rows
// (neighbouring pixel tiles) run on the same L2 close in time.  Placement only affects speed, never results.
__device__ __forceinline__ bool tile_of_block(int nbx, int nby, int& m_tile, int& n_tile) {
  const int lin = blockIdx.x;
  const int xcd = lin & 7, sq = lin >> 3;
  const int mchunk = (nbx + 7) >> 3;
  n_tile = sq % nby;
  const int m_local = sq / nby;
  m_tile = xcd * mchunk + m_local;
  return m_local < mchunk && m_tile < nbx;
}

struct DcnSample {
  int off[4];    // element offsets of the 4 corner pixels (already * in_stride), -1 = contributes 0
  float wt[4];   // bilinear weights
  float mask;    // sigmoid(mask logit)
  f16 wm[4];     // f16(wt[q] * mask): the MFMA path blends in packed f16 (v_pk_fma_f16)
};

// Sampling geometry of one (pixel, tap); follows deform_conv_cuda_kernel.cu:836-861 and :666-699.
__device__ __forceinline__ void dcn_setup(const ConvArgs& a, bool row_ok, int pix_base, int hb, int wb,
                                          int tr, int ts, const float* omrow, DcnSample& sp) {
  sp.off[0] = sp.off[1] = sp.off[2] = sp.off[3] = -1;
  sp.wt[0] = sp.wt[1] = sp.wt[2] = sp.wt[3] = 0.f;
  sp.wm[0] = sp.wm[1] = sp.wm[2] = sp.wm[3] = (f16)0.f;
  sp.mask = 0.f;
  if (!row_ok || tr >= a.R) return;
  const int tap = tr * a.S + ts;
  const float oh = omrow[2 * tap], ow = omrow[2 * tap + 1];
  const float mraw = omrow[2 * a.R * a.S + tap];
  sp.mask = a.mask_is_prob ? mraw : ctdet_sigmoid_exact(mraw);
  const float h_im = (float)(hb + tr * a.dil) + oh;
  const float w_im = (float)(wb + ts * a.dil) + ow;
  if (!(h_im > -1.f && w_im > -1.f && h_im < (float)a.H && w_im < (float)a.W)) return;
  const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
  const float hh = 1.f - lh, hw = 1.f - lw;
  sp.wt[0] = hh * hw; sp.wt[1] = hh * lw; sp.wt[2] = lh * hw; sp.wt[3] = lh * lw;
#pragma unroll
  for (int q = 0; q < 4; ++q) sp.wm[q] = (f16)(sp.wt[q] * sp.mask);
  if (h_low >= 0 && w_low >= 0) sp.off[0] = (pix_base + h_low * a.W + w_low) * a.in_stride;
  if (h_low >= 0 && w_high <= a.W - 1) sp.off[1] = (pix_base + h_low * a.W + w_high) * a.in_stride;
  if (h_high <= a.H - 1 && w_low >= 0) sp.off[2] = (pix_base + h_high * a.W + w_low) * a.in_stride;
  if (h_high <= a.H - 1 && w_high <= a.W - 1) sp.off[3] = (pix_base + h_high * a.W + w_high) * a.in_stride;
}

template <typename TOut>
__device__ __forceinline__ void epilogue_store4(const ConvArgs& a, int m, int c, f32x4 v) {
  // c is a multiple of 4; Cout is a multiple of 4 (host guarantees), so a group is all-in or all-out
  if (c >= a.Cout) return;
  if (a.scale) { const f32x4 s = *(const f32x4*)(a.scale + c); v = v * s; }
  if (a.bias) { const f32x4 b = *(const f32x4*)(a.bias + c); v = v + b; }
  if (a.res) {
    const TOut* rp = (const TOut*)a.res + (long)m * a.res_stride + c;
    if constexpr (sizeof(TOut) == 2) {
      const f16x4 r = *(const f16x4*)rp;
      v[0] += (float)r[0]; v[1] += (float)r[1]; v[2] += (float)r[2]; v[3] += (float)r[3];
    } else {
      v = v + *(const f32x4*)rp;
    }
  }
  if (a.act == CTDET_ACT_RELU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
  } else if (a.act == CTDET_ACT_SIGMOID_CLAMP) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fminf(fmaxf(ctdet_sigmoid_exact(v[j]), a.clamp_lo), a.clamp_hi);
  }
  TOut* yp = (TOut*)a.y + (long)m * a.out_stride + c;
  if constexpr (sizeof(TOut) == 2) {
    f16x4 o; o[0] = (f16)v[0]; o[1] = (f16)v[1]; o[2] = (f16)v[2]; o[3] = (f16)v[3];
    *(f16x4*)yp = o;
  } else {
    *(f32x4*)yp = v;
  }
}

// Cout (relative to the wave's first one) held in accumulator element i of cout tile c by the lanes of quad-group q
// (q = lane / 16).  Tiles pair up: a lane owns 8 consecutive couts per pair, so one 16-byte store per lane lets the
// four q-lanes of a pixel write 64 contiguous bytes (whole 32-byte sectors; the 8-byte pieces of the unpaired layout
// left every sector to be completed by four separate store instructions, and the store tail of a tile cost as much
// as a dozen K steps).  The weight loaders place LDS row (tile tt, row r) = cout_of(tt, r / 4, r % 4) to match.
template <int TC>
__device__ __forceinline__ constexpr int cout_of(int c, int q, int i) {
  if (TC % 2 == 0) return (c >> 1) * 32 + q * 8 + (c & 1) * 4 + i;
  return 4 * TC * q + 4 * c + i;
}

// scale/bias/residual/activation + store of the TC accumulator tiles one lane holds for output pixel m;
// cbase = first cout of the wave.
template <typename TOut, int TC>
__device__ __forceinline__ void epilogue_tiles(const ConvArgs& a, int m, int cbase, int q, const f32x4 (&acc)[TC]) {
  if constexpr (TC % 2 != 0) {
#pragma unroll
    for (int c = 0; c < TC; ++c) epilogue_store4<TOut>(a, m, cbase + cout_of<TC>(c, q, 0), acc[c]);
  } else {
    constexpr int VEC = 16 / sizeof(TOut);  // elements per 16-byte store
    const bool wide = (a.out_stride % VEC) == 0 && (((size_t)a.y) & 15) == 0 &&
                      (!a.res || ((a.res_stride % VEC) == 0 && (((size_t)a.res) & 15) == 0));
#pragma unroll
    for (int h = 0; h < TC / 2; ++h) {
      const int c0 = cbase + h * 32 + q * 8;
      if (!wide || c0 + 8 > a.Cout) {
        epilogue_store4<TOut>(a, m, c0, acc[2 * h]);
        epilogue_store4<TOut>(a, m, c0 + 4, acc[2 * h + 1]);
        continue;
      }
      f32x4 v0 = acc[2 * h], v1 = acc[2 * h + 1];
      if (a.scale) { v0 = v0 * *(const f32x4*)(a.scale + c0); v1 = v1 * *(const f32x4*)(a.scale + c0 + 4); }
      if (a.bias) { v0 = v0 + *(const f32x4*)(a.bias + c0); v1 = v1 + *(const f32x4*)(a.bias + c0 + 4); }
      if (a.res) {
        const TOut* rp = (const TOut*)a.res + (long)m * a.res_stride + c0;
        if constexpr (sizeof(TOut) == 2) {
          const f16x8 r = *(const f16x8*)rp;
#pragma unroll
          for (int j = 0; j < 4; ++j) { v0[j] += (float)r[j]; v1[j] += (float)r[4 + j]; }
        } else {
          v0 = v0 + *(const f32x4*)rp; v1 = v1 + *(const f32x4*)(rp + 4);
        }
      }
      if (a.act == CTDET_ACT_RELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { v0[j] = fmaxf(v0[j], 0.f); v1[j] = fmaxf(v1[j], 0.f); }
      } else if (a.act == CTDET_ACT_SIGMOID_CLAMP) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v0[j] = fminf(fmaxf(ctdet_sigmoid_exact(v0[j]), a.clamp_lo), a.clamp_hi);
          v1[j] = fminf(fmaxf(ctdet_sigmoid_exact(v1[j]), a.clamp_lo), a.clamp_hi);
        }
      }
      TOut* yp = (TOut*)a.y + (long)m * a.out_stride + c0;
      if constexpr (sizeof(TOut) == 2) {
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[j] = (f16)v0[j]; o[4 + j] = (f16)v1[j]; }
        *(f16x8*)yp = o;
      } else {
        *(f32x4*)yp = v0;
        *(f32x4*)(yp + 4) = v1;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// LDS-DMA variant for plain convolutions (everything except the DCNv2 sampler): tiles go HBM/L2 -> LDS
// with global_load_lds_dwordx4 (no staging registers, no ds_write), 3-stage LDS ring, loads two K steps
// ahead kept in flight across the barrier with a counted s_waitcnt vmcnt (never 0 in the main loop),
// one raw s_barrier per K step.  Padding / K-tail / rows beyond M read a 16-byte zero page instead of
// being zero-filled in registers.  The LDS image is the same swizzled [row][32 k] layout as above: the
// DMA writes lane-linear (wave base + lane*16), so the swizzle lives in which k-group a lane *fetches*.
// ------------------------------------------------------------------------------------------
static __device__ __attribute__((aligned(16))) unsigned int g_zero_page[64];

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// All of this wave's LDS operations have completed.  Through the builtin, not inline asm: the compiler's own waitcnt
// bookkeeping then KNOWS the counter is zero.  With an asm wait it still counts fragments that were read before a barrier
// for use behind it as in flight, and puts s_waitcnt lgkmcnt(n) in front of the MFMAs that consume them -- n chosen so
// that they wait for the reads issued AFTER the barrier (the next step's operands): the register double buffer is undone.
// Encoding (gfx9): vmcnt 63 and expcnt 7 = "don't wait", lgkmcnt 0.
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xC07F); }

// Global loads of the rare slow paths (a DCNv2 sample outside the LDS window), issued AND awaited inside one asm block.  A
// plain C++ load there leaves "a VMEM load may be pending" in the compiler's waitcnt bookkeeping at the join with the fast
// path, and the pass then puts s_waitcnt vmcnt(0) into the fast path's MFMA stream -- every tap -- which also waits for the
// LDS-DMAs this file issues through asm (weights of the next kernel row, the next window): their latency, meant to be
// hidden until the next barrier, is exposed right behind the issue.
__device__ __forceinline__ void gload2_sync(const float* p, float& v0, float& v1) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  f32x2_ v;
  asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  v0 = v[0]; v1 = v[1];
}
// the active lanes replace v by the 16 bytes at p (the others keep theirs: call it inside the divergent branch)
__device__ __forceinline__ void gload4_sync_into(const float* p, f32x4& v) {
  asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "+v"(v) : "v"(p) : "memory");
}

// One 16-byte-per-lane global->LDS DMA (global_load_lds_dwordx4: LDS address = M0 + lane*16).  Issued through inline
// asm on purpose: the compiler's waitcnt pass treats every LDS read as possibly aliasing every outstanding
// __builtin_amdgcn_global_load_lds and puts s_waitcnt vmcnt(0) in front of it, which serialises the multi-stage
// rings below.  All consumers here order DMA -> LDS read themselves (wait_vmcnt<N>() + s_barrier before the first
// read of a stage, lgkmcnt(0) + s_barrier before a stage is overwritten).
__device__ __forceinline__ void dma16(const void* g, char* lds_wave_base) {
  const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(g) : "memory", "m0");
}

// ------------------------------------------------------------------------------------------
// f16x3 ("split") arithmetic of the f32-activation kernels (conv_f32.hip, SP = true): an f32 operand x is carried as
// hi + lo, hi = f16(x), lo = f16(x - hi) (x - hi is exact in f32, so hi + lo reproduces x to ~2^-22 relative, 3e-8
// absolute once lo reaches the f16 subnormals) and a product a*w runs on the f16 matrix pipe as
// a_hi*w_hi + a_lo*w_hi + a_hi*w_lo with f32 accumulation; the dropped a_lo*w_lo is ~2^-22 of the product.  Exact f16
// products, f32 sums: the error of a K-long dot product measures 6e-8 * sum|a w| (tools/split_probe.hip), the f32 MFMA
// chain 1e-7.  Activations stay f32 in HBM and LDS (the f32 kernels' images unchanged); weights are split once at pack
// time (ctdet_split_weights): each group of 4 consecutive k of a packed [Cout_pad][Kpad] f32 row becomes 16 bytes
// {w_hi[4], w_lo[4]}, which IS the A fragment a lane reads with ds_read_b128.  Per 16-k step and (cout tile, pixel tile):
//   acc += A . {a_hi[4], 0}         -> w_hi a_hi
//   acc += A . {a_lo[4], a_hi[4]}   -> w_hi a_lo + w_lo a_hi
// two v_mfma_f32_16x16x32_f16 of the same opcode on one accumulator (dependent issue needs no wait states; a 16x16x16
// second product would cost 87 % of a 16x16x32 and mixes opcodes on one accumulator).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void split_b(const f32x4 x, f16x8& b1, f16x8& b2) {
  const f16x4 hi = __builtin_convertvector(x, f16x4);
  f32x4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = x[j] - (float)hi[j];
  const f16x4 lo = __builtin_convertvector(r, f16x4);
  const f16x4 z = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
  b1 = __builtin_shufflevector(hi, z, 0, 1, 2, 3, 4, 5, 6, 7);
  b2 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// one 16-k step of one pixel tile against TC cout tiles: wf = the lanes' 16-byte weight fragments (4 f32 k, or the split
// group), pf = 4 f32 k of the lane's pixel
template <bool SP, int TC>
__device__ __forceinline__ void mma_px(const f32x4 (&wf)[TC], const f32x4 pf, f32x4 (&acc)[TC]) {
  if constexpr (SP) {
    f16x8 b1, b2;
    split_b(pf, b1, b2);
#pragma unroll
    for (int c = 0; c < TC; ++c)
      acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[c]), b1, acc[c], 0, 0, 0);
#pragma unroll
    for (int c = 0; c < TC; ++c)
      acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[c]), b2, acc[c], 0, 0, 0);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < TC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[c][e], pf[e], acc[c], 0, 0, 0);
  }
}

// k-th weight of a packed row as f32 (scalar fallback kernels)
template <bool SP>
__device__ __forceinline__ float packed_w(const float* row, int k) {
  if constexpr (SP) {
    const f16* g = (const f16*)(row + (k & ~3));
    return (float)g[k & 3] + (float)g[4 + (k & 3)];
  } else {
    return row[k];
  }
}

// cout tile of a conv (= ctdet_conv_cout_tile): packed weight rows are padded to a multiple of it
static inline int pick_bc(int cout) {
  if (cout <= 16) return 16;
  if (cout <= 32) return 32;
  if (cout <= 64) return 64;
  if (cout <= 128) return 128;          // 80 classes: one padded 128-cout tile reads the input once, three 32-cout tiles three times
  if (cout % 128 == 0) return 128;
  if (cout % 64 == 0) return 64;
  return 32;
}
