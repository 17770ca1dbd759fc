// Conv-shaped contractions of the CenterNet path for gfx950 (CDNA4): NHWC f16 activations, packed KRSC weights,
// v_mfma_f32_16x16x32_f16 with the weights as the A operand and the pixels as the B operand
// (D[cout][pixel] = sum_k W[cout][k] * A[pixel][k]), so a lane ends up with consecutive output channels of one pixel.
// Kernels in this file (launch_conv_f16 picks one):
//   conv3x3_halo_kernel   3x3/s1/p1, Cin % 32 == 0: input window of an 8x32 tile in LDS once per 32-channel chunk
//   head_fused_kernel     CenterNet heads: 3x3 + ReLU + 1x1 per head, hidden map in registers
//   conv_igemm_uk_kernel  1x1 / strided / Root (multi-source) convs: uniform-K im2col-on-the-fly tiles
//   conv_igemm_dma_kernel generic fallback (odd channel counts, input dilation for strided input gradients)
//   conv_win_kernel, conv_smallc_kernel   the 3/16-channel DLA base layers
//   dcn_window_kernel     DCNv2 (deform_conv_cuda_kernel.cu:666-868 + deform_conv_cuda.cu:874-927): bilinear gathers from
//                         an LDS window, blended straight into MFMA operand registers; no `columns` buffer, batch in M
//   (the f32 mode of all of the above lives in conv_f32.hip)
// LDS tiles are [row][32 k] f16 (64-byte rows) with a 16-byte-slot XOR swizzle (swz()) that makes the ds_read_b128
// fragment reads conflict free; global->LDS goes through LDS-DMA with counted vmcnt waits and one s_barrier per K step.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

#include "conv_common.h"

template <int BP, int BC, int WP, int WC_, typename TOut>
__global__ void __launch_bounds__(256) conv_igemm_dma_kernel(const ConvArgs a) {
  constexpr int TP = BP / WP / 16;
  constexpr int TC = BC / WC_ / 16;
  constexpr int A_LD = BP / 64;
  constexpr int BCL = BC < 64 ? 64 : BC;  // weight rows staged (>= 64 so every wave issues the same DMA count)
  constexpr int B_LD = BCL / 64;
  constexpr int NLOAD = A_LD + B_LD;
  constexpr int STAGE = (BP + BCL) * 64;
  constexpr int NST = 3;
  static_assert(WP * WC_ == 4, "4 waves per workgroup");
  __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave / WC_, wc = wave % WC_;
  int m_tile, n_tile;
  if (!tile_of_block((a.M + BP - 1) / BP, a.Cout_pad / BC, m_tile, n_tile)) return;
  const int m0 = m_tile * BP, n0 = n_tile * BC;
  const f16* __restrict__ x = (const f16*)a.x;
  const f16* __restrict__ w = (const f16*)a.w;
  // the zero page pointer is made opaque (lives in a VGPR pair) so that `cond ? real : zero` compiles to a
  // v_cndmask and ONE global_load_lds per staged row; if the compiler sees a uniform address on one side it
  // emits two exec-masked DMAs instead, which would break the counted vmcnt below.
  const f16* zero = (const f16*)g_zero_page;
  asm volatile("" : "+v"(zero));

  const int lrow = tid >> 2, slot = tid & 3;
  const int g = slot ^ swz(lrow);
  // per staged pixel row: element offset of tap (0,0) and a bit mask of the taps that fall inside the image
  int rb[A_LD], rhb[A_LD], rwb[A_LD];
  unsigned long long tapmask[A_LD];
  long rowm[A_LD];
  const int idl = a.in_dil;
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const int m = m0 + lrow + 64 * i;
    const bool ok = m < a.M;
    const int mm = ok ? m : 0;
    const int wo = mm % a.Wo, t = mm / a.Wo;
    const int ho = t % a.Ho, b = t / a.Ho;
    const int hb = ho * a.stride - a.pad, wb = wo * a.stride - a.pad;
    rb[i] = b; rhb[i] = hb; rwb[i] = wb;
    rowm[i] = ok ? (long)m : -1;
    unsigned long long mk = 0;
    if (ok)
      for (int r = 0; r < a.R; ++r)
        for (int s2 = 0; s2 < a.S; ++s2) {
          const int hn = hb + r * a.dil, wn = wb + s2 * a.dil;  // position in the (zero-stuffed) input
          if (hn >= 0 && wn >= 0 && hn % idl == 0 && wn % idl == 0 && hn / idl < a.H && wn / idl < a.W)
            mk |= 1ull << (r * a.S + s2);
        }
    tapmask[i] = mk;
  }
  long b_off[B_LD];
  bool b_ok[B_LD];
#pragma unroll
  for (int j = 0; j < B_LD; ++j) {
    const int L = lrow + 64 * j;
    const int Lw = L % (16 * TC), wv = L / (16 * TC);
    const int tt = Lw >> 4, r = Lw & 15;
    const int cl = wv * 16 * TC + cout_of<TC>(tt, r >> 2, r & 3);
    b_ok[j] = L < BC;
    b_off[j] = (long)(n0 + (b_ok[j] ? cl : 0)) * a.Kpad + g * 8;
  }

  int c0, tr, ts;
  {
    const int kc = g * 8, tap = kc / a.Cin;
    c0 = kc - tap * a.Cin;
    tr = tap / a.S;
    ts = tap - tr * a.S;
  }
  const bool chunk_major = a.korder == 1;
  auto advance_k = [&]() {
    if (chunk_major) {
      // next tap of the same 32-channel chunk; after the last tap move to the next chunk
      if (++ts == a.S) { ts = 0; if (++tr == a.R) { tr = 0; c0 += 32; } }
    } else {
      c0 += 32;
      while (c0 >= a.Cin) {
        c0 -= a.Cin;
        if (++ts == a.S) { ts = 0; ++tr; }
      }
    }
  };

  auto issue = [&](int kt, int stage) {
    char* sb = smem + stage * STAGE + wave * 1024;
    if (a.nsrc > 1) {
      const int kk = kt * 32 + g * 8;
      const f16* src = (const f16*)a.xs[0];
      int st = a.xs_stride[0], cb0 = 0;
      if (kk >= a.xs_cend[0]) { src = (const f16*)a.xs[1]; st = a.xs_stride[1]; cb0 = a.xs_cend[0]; }
      if (a.nsrc > 2 && kk >= a.xs_cend[1]) { src = (const f16*)a.xs[2]; st = a.xs_stride[2]; cb0 = a.xs_cend[1]; }
      if (a.nsrc > 3 && kk >= a.xs_cend[2]) { src = (const f16*)a.xs[3]; st = a.xs_stride[3]; cb0 = a.xs_cend[2]; }
      const bool kin = kk < a.Cin;
#pragma unroll
      for (int i = 0; i < A_LD; ++i)
        dma16((kin && rowm[i] >= 0) ? src + rowm[i] * st + (kk - cb0) : zero, sb + i * 4096);
    } else {
      const int tap = tr * a.S + ts;
      const bool kin = tr < a.R && c0 < a.Cin;
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        const int hi = (rhb[i] + tr * a.dil) / idl, wi = (rwb[i] + ts * a.dil) / idl;
        dma16((kin && ((tapmask[i] >> tap) & 1ull)) ? x + ((long)(rb[i] * a.H + hi) * a.W + wi) * a.in_stride + c0 : zero,
              sb + i * 4096);
      }
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) dma16(b_ok[j] ? w + b_off[j] + kt * 32 : zero, sb + BP * 64 + j * 4096);
  };

  f32x4 acc[TP][TC];
#pragma unroll
  for (int p = 0; p < TP; ++p)
#pragma unroll
    for (int c = 0; c < TC; ++c) acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15;
  const int frag_off = fr * 64 + (((lane >> 4) ^ swz(fr)) << 4);
  const int nk = a.Kpad / 32;

  // prologue: two tiles in flight
  issue(0, 0);
  advance_k();
  if (nk > 1) { issue(1, 1); advance_k(); }

  int st_c = 0, st_l = 2;  // stage being computed / stage the next prefetch goes to
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed once at most one younger tile's DMAs are still outstanding
    if (kt + 1 < nk) wait_vmcnt<NLOAD>(); else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of the stage about to be refilled are done
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk) { issue(kt + 2, st_l); advance_k(); }
    const char* base = smem + st_c * STAGE;
    f16x8 wf[TC];
#pragma unroll
    for (int c = 0; c < TC; ++c) wf[c] = *(const f16x8*)(base + BP * 64 + (wc * 16 * TC + 16 * c) * 64 + frag_off);
#pragma unroll
    for (int p = 0; p < TP; ++p) {
      const f16x8 pf = *(const f16x8*)(base + (wp * 16 * TP + 16 * p) * 64 + frag_off);
#pragma unroll
      for (int c = 0; c < TC; ++c) acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c], pf, acc[p][c], 0, 0, 0);
    }
    st_c = st_c == NST - 1 ? 0 : st_c + 1;
    st_l = st_l == NST - 1 ? 0 : st_l + 1;
  }

  const int q = lane >> 4;
  const int cb = n0 + wc * 16 * TC;
#pragma unroll
  for (int p = 0; p < TP; ++p) {
    const int m = m0 + wp * 16 * TP + 16 * p + fr;
    if (m >= a.M) continue;
    epilogue_tiles<TOut, TC>(a, m, cb, q, acc[p]);
  }
}

// ------------------------------------------------------------------------------------------
// Uniform-K variant of the LDS-DMA kernel: when Cin (and every concat source) is a multiple of 32, the tap /
// channel-chunk state of a K step is the same for every lane, so it lives in SGPRs; per lane only a fixed row
// pointer (+ the lane's k-group offset) and a 32-bit tap-validity mask remain.  This removes ~80 per-lane
// 64-bit VALU ops per K step and ~70 VGPRs compared with the generic kernel above (which stays for odd channel
// counts), which is what lets two workgroups share a CU (arch VGPRs + 128 accumulators <= 256).
// ------------------------------------------------------------------------------------------
template <int BP, int BC, int WP, int WC_, bool CAT, typename TOut>
__global__ void __launch_bounds__(256, 2) conv_igemm_uk_kernel(const ConvArgs a) {
  constexpr int TP = BP / WP / 16;
  constexpr int TC = BC / WC_ / 16;
  constexpr int A_LD = BP / 64;
  constexpr int BCL = BC < 64 ? 64 : BC;
  constexpr int B_LD = BCL / 64;
  constexpr int NLOAD = A_LD + B_LD;
  constexpr int STAGE = (BP + BCL) * 64;
  constexpr int NST = 3;
  static_assert(WP * WC_ == 4, "4 waves per workgroup");
  __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave / WC_, wc = wave % WC_;
  int m_tile, n_tile;
  if (!tile_of_block((a.M + BP - 1) / BP, a.Cout_pad / BC, m_tile, n_tile)) return;
  const int m0 = m_tile * BP, n0 = n_tile * BC;
  const f16* zero = (const f16*)g_zero_page;
  asm volatile("" : "+v"(zero));

  const int lrow = tid >> 2, slot = tid & 3;
  const int g = slot ^ swz(lrow);

  const f16* rowp[A_LD];   // conv: pixel of tap (0,0) + g*8 channels; cat: row of the current source + g*8
  unsigned tapmask[A_LD];  // conv: bit t = tap t inside the image; cat: bit 0 = row < M
  int rowm[A_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const int m = m0 + lrow + 64 * i;
    const bool ok = m < a.M;
    const int mm = ok ? m : 0;
    rowm[i] = mm;
    unsigned mk = 0;
    if constexpr (CAT) {
      mk = ok ? 1u : 0u;
      rowp[i] = (const f16*)a.xs[0] + (long)mm * a.xs_stride[0] + g * 8;
    } else {
      const int wo = mm % a.Wo, t = mm / a.Wo;
      const int ho = t % a.Ho, b = t / a.Ho;
      const int hb = ho * a.stride - a.pad, wb = wo * a.stride - a.pad;
      if (ok)
        for (int r = 0; r < a.R; ++r)
          for (int s2 = 0; s2 < a.S; ++s2) {
            const int hi = hb + r * a.dil, wi = wb + s2 * a.dil;
            if (hi >= 0 && hi < a.H && wi >= 0 && wi < a.W) mk |= 1u << (r * a.S + s2);
          }
      rowp[i] = (const f16*)a.x + ((long)b * a.H * a.W + (long)hb * a.W + wb) * a.in_stride + g * 8;
    }
    tapmask[i] = mk;
  }
  const f16* wptr[B_LD];
#pragma unroll
  for (int j = 0; j < B_LD; ++j) {
    const int L = lrow + 64 * j;
    const int Lw = L % (16 * TC), wv = L / (16 * TC);
    const int tt = Lw >> 4, r = Lw & 15;
    const int cl = wv * 16 * TC + cout_of<TC>(tt, r >> 2, r & 3);
    // rows beyond BC (only when BC < 64) re-read packed row 0: harmless, their LDS rows are never consumed
    wptr[j] = (const f16*)a.w + (long)(n0 + (L < BC ? cl : 0)) * a.Kpad + g * 8;
  }

  // ---- uniform (scalar) K-step state, advanced incrementally ----
  // conv (chunk-major k): walk the S taps of a row (+step_s), then the next row (+step_r), then the next
  // 32-channel chunk (+step_c); `dlt` is the element offset added to every row pointer, `tbit` the tap's mask bit.
  const long step_s = (long)a.dil * a.in_stride;
  const long step_r = ((long)a.dil * a.W - (long)(a.S - 1) * a.dil) * a.in_stride;
  const long step_c = 32 - ((long)(a.R - 1) * a.dil * a.W + (long)(a.S - 1) * a.dil) * a.in_stride;
  long dlt = 0;
  unsigned tbit = 1u;
  int ts = 0, tr = 0;
  int sj = 0, sleft = CAT ? a.xs_cend[0] : 0;  // cat: current source and channels left in it
  long woff = 0;                               // element offset into the packed weight rows
  auto advance_k = [&]() {
    woff += 32;
    if constexpr (CAT) {
      dlt += 32;
      sleft -= 32;
      if (sleft == 0 && sj + 1 < a.nsrc) {
        ++sj;
        sleft = a.xs_cend[sj] - a.xs_cend[sj - 1];
        dlt = 0;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) rowp[i] = (const f16*)a.xs[sj] + (long)rowm[i] * a.xs_stride[sj] + g * 8;
      }
    } else {
      if (++ts < a.S) { dlt += step_s; tbit <<= 1; }
      else {
        ts = 0;
        if (++tr < a.R) { dlt += step_r; tbit <<= 1; }
        else { tr = 0; dlt += step_c; tbit = 1u; }
      }
    }
  };
  auto issue = [&](char* sb) {  // sb: this wave's slice of the target stage
#pragma unroll
    for (int i = 0; i < A_LD; ++i) dma16((tapmask[i] & tbit) ? rowp[i] + dlt : zero, sb + i * 4096);
#pragma unroll
    for (int j = 0; j < B_LD; ++j) dma16(wptr[j] + woff, sb + BP * 64 + j * 4096);
  };

  f32x4 acc[TP][TC];
#pragma unroll
  for (int p = 0; p < TP; ++p)
#pragma unroll
    for (int c = 0; c < TC; ++c) acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15;
  const int frag_off = fr * 64 + (((lane >> 4) ^ swz(fr)) << 4);
  const char* fragA = smem + (wp * 16 * TP) * 64 + frag_off;            // pixel fragments of this wave
  const char* fragB = smem + BP * 64 + (wc * 16 * TC) * 64 + frag_off;  // weight fragments of this wave
  char* dmab = smem + wave * 1024;
  const int nk = a.Kpad / 32;

  issue(dmab);
  advance_k();
  if (nk > 1) { issue(dmab + STAGE); advance_k(); }

  // one K step on compile-time stage ST (compute) / SL (prefetch target): LDS offsets become immediates
  auto kstep = [&](int kt, auto st_c, auto st_l) {
    constexpr int ST = decltype(st_c)::value, SL = decltype(st_l)::value;
    if (kt + 1 < nk) wait_vmcnt<NLOAD>(); else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk) { issue(dmab + SL * STAGE); advance_k(); }
    f16x8 wf[TC];
#pragma unroll
    for (int c = 0; c < TC; ++c) wf[c] = *(const f16x8*)(fragB + ST * STAGE + c * 1024);
#pragma unroll
    for (int p = 0; p < TP; ++p) {
      const f16x8 pf = *(const f16x8*)(fragA + ST * STAGE + p * 1024);
#pragma unroll
      for (int c = 0; c < TC; ++c) acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c], pf, acc[p][c], 0, 0, 0);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  int kt = 0;
  for (; kt + 2 < nk; kt += 3) {
    kstep(kt, I0{}, I2{});
    kstep(kt + 1, I1{}, I0{});
    kstep(kt + 2, I2{}, I1{});
  }
  if (kt < nk) { kstep(kt, I0{}, I2{}); ++kt; }
  if (kt < nk) { kstep(kt, I1{}, I0{}); ++kt; }

  const int q = lane >> 4;
  const int cb = n0 + wc * 16 * TC;
#pragma unroll
  for (int p = 0; p < TP; ++p) {
    const int m = m0 + wp * 16 * TP + 16 * p + fr;
    if (m >= a.M) continue;
    epilogue_tiles<TOut, TC>(a, m, cb, q, acc[p]);
  }
}

// ------------------------------------------------------------------------------------------
// Halo-resident 3x3 / stride 1 / pad 1 convolution (every DLABasicBlock conv, the DCN offset/mask convs, the fused
// head conv, and their input gradients): a workgroup owns an 8x32-pixel output tile of one image and walks K
// chunk-major.  For each 32-channel chunk the (8+2) x (32+2) input window is brought into LDS ONCE (6 DMA
// instructions per thread) and all nine taps read their pixel fragments from it at shifted addresses; only the
// 32-deep weight tile of each (chunk, tap) streams through a 3-stage ring.  Compared with the per-tap staging of
// the kernels above this issues 2-3x fewer LDS-DMA instructions per MFMA (the measured limiter: the plain kernel
// tops out at 44% / 23% of MFMA peak even with all loads served from one L1 line) and fetches each input byte from
// L2 once per cout tile instead of nine times.
// LDS: halo[2] x {main [10 rows][32 px][64 B] swizzled like the tiles above, side [10][left,right][64 B]} + ring[3].
// ------------------------------------------------------------------------------------------
// SP (the f16x3 mode of conv_f32.hip, TOut = float): the same kernel on f32 activations -- a chunk is 16 channels, again 64
// bytes per window pixel, so the LDS images, the DMA schedule and every wait count are unchanged; the weights are the split
// image of the tap-major f32 pack (the K loop fetches the 64-byte piece (tap, chunk) of every cout row, as the f32 DCNv2
// window kernel does) and a K step is mma_px<true> (two f16 MFMAs per tile on hi / lo halves, conv_common.h).
template <int BC, int WP, int WC_, typename TOut, bool SP = false>
__global__ void __launch_bounds__(256, 2) conv3x3_halo_kernel(const ConvArgs a) {
  using TIn = std::conditional_t<SP, float, f16>;
  constexpr int CH = SP ? 16 : 32, EPV = SP ? 4 : 8;   // channels per chunk (64 bytes), elements per 16-byte vector
  static_assert(!SP || std::is_same<TOut, float>::value, "split mode stores f32");
  constexpr int TH = 8, TW = 32, BP = TH * TW;
  constexpr int TP = BP / WP / 16;      // 16-pixel tiles per wave
  constexpr int TC = BC / WC_ / 16;
  constexpr int ROWS_W = TH / WP;       // tile rows per wave
  constexpr int BCL = BC < 64 ? 64 : BC;
  constexpr int B_LD = BCL / 64;
  constexpr int HMAIN = 10 * 32 * 64, HSIDE = 4096, HBUF = HMAIN + HSIDE;
  constexpr int WST = BCL * 64;
  static_assert(WP * WC_ == 4 && TP == 2 * ROWS_W, "wave layout");
  __shared__ __attribute__((aligned(16))) char smem[2 * HBUF + 3 * WST];
  char* const ring = smem + 2 * HBUF;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave / WC_, wc = wave % WC_;
  const int tiles_x = a.W / TW, tiles_y = a.H / TH;
  int m_tile, n_tile;
  if (!tile_of_block(a.B * tiles_y * tiles_x, a.Cout_pad / BC, m_tile, n_tile)) return;
  const int tx0 = (m_tile % tiles_x) * TW;
  const int ty0 = ((m_tile / tiles_x) % tiles_y) * TH;
  const int b = m_tile / (tiles_x * tiles_y);
  const int n0 = n_tile * BC;
  const TIn* zero = (const TIn*)g_zero_page;
  asm volatile("" : "+v"(zero));
  const TIn* ximg = (const TIn*)a.x + (long)b * a.H * a.W * a.in_stride;

  // ---- halo loader: 5 main pieces + 1 side piece per thread and chunk ----
  // main piece i of thread t: pid = t + 256 i -> halo row (t>>7) + 2i, pixel (t>>2)&31, slot t&3: rows two apart,
  // so one base pointer + a uniform row-pair stride suffice
  const int hslot = tid & 3, hpx = (tid >> 2) & 31, hr0 = tid >> 7;
  const int y0 = ty0 - 1 + hr0;
  const TIn* hp0 = ximg + ((long)y0 * a.W + tx0 + hpx) * a.in_stride + (hslot ^ swz(hpx)) * EPV;
  const long row2 = 2L * a.W * a.in_stride;
  unsigned hmask = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) hmask |= (y0 + 2 * i >= 0 && y0 + 2 * i < a.H) ? (1u << i) : 0u;
  const TIn* hps;
  {
    const int side = (tid >> 2) & 1, hr = tid >> 3;   // [hr 0..9][side][slot], tid < 80
    const int y = ty0 - 1 + hr, x = side ? tx0 + TW : tx0 - 1;
    const bool ok = tid < 80 && y >= 0 && y < a.H && x >= 0 && x < a.W;
    hps = ximg + ((long)(ok ? y : 0) * a.W + (ok ? x : 0)) * a.in_stride + hslot * EPV;
    hmask |= ok ? 32u : 0u;
  }
  const int lrow = tid >> 2;
  const int gw = hslot ^ swz(lrow);
  const TIn* wptr[B_LD];
#pragma unroll
  for (int j = 0; j < B_LD; ++j) {
    const int L = lrow + 64 * j;
    const int Lw = L % (16 * TC), wv = L / (16 * TC);
    const int tt = Lw >> 4, r = Lw & 15;
    const int cl = wv * 16 * TC + cout_of<TC>(tt, r >> 2, r & 3);
    wptr[j] = (const TIn*)a.w + (long)(n0 + (L < BC ? cl : 0)) * a.Kpad + gw * EPV;
  }
  auto issue_halo = [&](int chunk, int hb) {
    char* dst = smem + hb * HBUF + wave * 1024;
    const long coff = (long)chunk * CH;
#pragma unroll
    for (int i = 0; i < 5; ++i) dma16((hmask & (1u << i)) ? hp0 + i * row2 + coff : zero, dst + i * 4096);
    dma16((hmask & 32u) ? hps + coff : zero, smem + hb * HBUF + HMAIN + wave * 1024);
  };
  // K step kt = chunk * 9 + tap.  f16: packed chunk-major, step kt is the kt-th 32-k piece of a row; SP: packed tap-major
  auto issue_w = [&](int kt, int st) {
    long koff;
    if constexpr (SP) { const int ch = kt / 9, tp = kt - 9 * ch; koff = (long)tp * a.Cin + ch * CH; }
    else koff = (long)kt * 32;
#pragma unroll
    for (int j = 0; j < B_LD; ++j) dma16(wptr[j] + koff, ring + st * WST + wave * 1024 + j * 4096);
  };

  // ---- fragment addressing: per lane one LDS base per (px-tile half e, tap column s) with the wave's first tile
  // row folded in; interior lanes then use immediates for (tile row + tap row) * 2048.  The single edge lane of
  // (e=0,s=0) [left halo column] and (e=1,s=2) [right halo column] reads the side region (row stride 128). ----
  const int l15 = lane & 15, kg = lane >> 4;
  const int row0 = wp * ROWS_W;
  int abase[2][3];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2) {
      const int X = 16 * e + l15 + s2 - 1;
      if (X < 0) abase[e][s2] = HMAIN + kg * 16 + row0 * 128;
      else if (X > 31) abase[e][s2] = HMAIN + 64 + kg * 16 + row0 * 128;
      else abase[e][s2] = X * 64 + ((kg ^ swz(X)) << 4) + row0 * 2048;
    }
  const int estride0 = (l15 == 0) ? 128 : 2048;    // row stride of this lane for (e=0, s=0)
  const int estride1 = (l15 == 15) ? 128 : 2048;   // ... for (e=1, s=2)
  const int fr_off = l15 * 64 + ((kg ^ swz(l15)) << 4);
  const char* fragB = ring + (wc * 16 * TC) * 64 + fr_off;

  f32x4 acc[TP][TC];
#pragma unroll
  for (int p = 0; p < TP; ++p)
#pragma unroll
    for (int c = 0; c < TC; ++c) acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nch = a.Cin / CH, nk = nch * 9;
  issue_halo(0, 0);
  issue_w(0, 0);
  issue_w(1, 1);

  // one K step = tap T of the current chunk; tap, ring stage (9 % 3 == 0) and halo buffer are compile time
  auto kstep = [&](int kt, int chunk, auto tapc, auto hbc) {
    constexpr int T = decltype(tapc)::value, HB = decltype(hbc)::value;
    constexpr int R_ = T / 3, S_ = T % 3, ST = T % 3, SL = (T + 2) % 3;
    // outstanding DMAs allowed while waiting for weights(kt): the younger weight tile (B_LD) and, right after a
    // halo prefetch was queued behind it (T == 1), those 6 as well
    // (only when a prefetch WAS queued: in the last chunk the larger allowance would let the wait pass with
    // weights(kt) itself still in flight -- wrong tiles under memory load, found by the full-size determinism test)
    if (kt + 1 < nk) { if (T == 1 && chunk + 1 < nch) wait_vmcnt<B_LD + 6>(); else wait_vmcnt<B_LD>(); }
    else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (T == 0 && chunk + 1 < nch) issue_halo(chunk + 1, HB ^ 1);
    if (kt + 2 < nk) issue_w(kt + 2, SL);
    const char* hbuf = smem + HB * HBUF;
    f32x4 wf[TC];    // 16 bytes: 8 f16 k (f16 mode) / the split group of 4 k (SP)
#pragma unroll
    for (int c = 0; c < TC; ++c) wf[c] = *(const f32x4*)(fragB + ST * WST + c * 1024);
#pragma unroll
    for (int p = 0; p < TP; ++p) {
      const int e = p & 1, lr = p >> 1;
      int off;
      if (e == 0 && S_ == 0) off = abase[0][0] + (lr + R_) * estride0;
      else if (e == 1 && S_ == 2) off = abase[1][2] + (lr + R_) * estride1;
      else off = abase[e][S_] + (lr + R_) * 2048;
      const f32x4 pf = *(const f32x4*)(hbuf + off);
      if constexpr (SP) {
        mma_px<true, TC>(wf, pf, acc[p]);
      } else {
#pragma unroll
        for (int c = 0; c < TC; ++c)
          acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[c]), __builtin_bit_cast(f16x8, pf),
                                                             acc[p][c], 0, 0, 0);
      }
    }
  };
  auto chunk_steps = [&](int kt, int chunk, auto hbc) {
    kstep(kt + 0, chunk, std::integral_constant<int, 0>{}, hbc);
    kstep(kt + 1, chunk, std::integral_constant<int, 1>{}, hbc);
    kstep(kt + 2, chunk, std::integral_constant<int, 2>{}, hbc);
    kstep(kt + 3, chunk, std::integral_constant<int, 3>{}, hbc);
    kstep(kt + 4, chunk, std::integral_constant<int, 4>{}, hbc);
    kstep(kt + 5, chunk, std::integral_constant<int, 5>{}, hbc);
    kstep(kt + 6, chunk, std::integral_constant<int, 6>{}, hbc);
    kstep(kt + 7, chunk, std::integral_constant<int, 7>{}, hbc);
    kstep(kt + 8, chunk, std::integral_constant<int, 8>{}, hbc);
  };
  int chunk = 0;
  for (; chunk + 1 < nch; chunk += 2) {
    chunk_steps(chunk * 9, chunk, std::integral_constant<int, 0>{});
    chunk_steps(chunk * 9 + 9, chunk + 1, std::integral_constant<int, 1>{});
  }
  if (chunk < nch) chunk_steps(chunk * 9, chunk, std::integral_constant<int, 0>{});

  const int q = lane >> 4;
  const int cb = n0 + wc * 16 * TC;
#pragma unroll
  for (int p = 0; p < TP; ++p) {
    const int y = ty0 + row0 + (p >> 1), x = tx0 + 16 * (p & 1) + l15;
    const int m = (b * a.H + y) * a.W + x;
    epilogue_tiles<TOut, TC>(a, m, cb, q, acc[p]);
  }
}

template <int BC, int WP, int WC_, typename TOut, bool SP = false>
static int launch_halo(const ConvArgs& a, hipStream_t s) {
  const int nbx = a.B * (a.H / 8) * (a.W / 32), nby = a.Cout_pad / BC;
  dim3 grid(8 * ((nbx + 7) / 8) * nby);
  hipLaunchKernelGGL((conv3x3_halo_kernel<BC, WP, WC_, TOut, SP>), grid, dim3(256), 0, s, a);
  CTDET_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// f16x3 mode of the halo-resident 3x3 convolution, second form ("pair" weights, ctdet_conv_desc.korder 2).  The SP
// instantiation above splits every pixel fragment in registers once per tap (nine times per window pixel and chunk: 40 VALU
// per 32 MFMAs) and spends four MFMAs per 32 k-products' worth of work.  Here
//   * the f32 window of a 16-channel chunk is converted IN PLACE, once, to the split form: every 16-byte piece (4 f32
//     channels of a pixel) becomes {hi[4], lo[4]} f16.  A thread converts exactly the pieces its own LDS-DMAs fetched, two
//     steps before they are first read, so the pass needs no barrier of its own;
//   * two taps share an MFMA: the B operands are H = {hi(tap t), hi(tap t+1)} and L = {lo(t), lo(t+1)} -- four ds_read_b64
//     straight from the converted window, no VALU -- and the weights come packed per tap pair as X = {w_hi(t), w_hi(t+1)},
//     Y = {w_lo(t), w_lo(t+1)} (ops.PackedConv._pack_pairs).  acc += X.H + Y.H + X.L: three 16x16x32 MFMAs per 32
//     k-products instead of four.  The nine taps are five pairs with a zero tenth tap (15 MFMAs per tile and chunk, was 18);
//   * a step's MFMAs start right behind its barrier: their operands are read DURING the previous step -- the weight
//     fragments into a second register set (the barrier of step k vouches for the weights of step k+1), the pixel fragments
//     of tile p into the registers the MFMAs of tile p have just released -- and the step's LDS-DMAs are issued between its
//     MFMAs.  With the reads and DMA issue in front of the MFMAs the kernel ran at the SUM of its skeleton (DMA issue,
//     barrier, LDS reads: 132 us on 128->128 @64^2, batch 64, measured with the MFMAs removed) and its MFMA time (137 us):
//     a wave parks ~500 cycles per step on LDS latency and the barrier, and its SIMD partner is often parked with it.
// LDS: halo[2] as above + a 3-stage ring of {X image, Y image} (BCL rows x 64 B each).
// ------------------------------------------------------------------------------------------
template <int BC, int WP, int WC_>
__global__ void __launch_bounds__(256, 2) conv3x3_halo_pair_kernel(const ConvArgs a) {
  constexpr int TH = 8, TW = 32, BP = TH * TW;
  constexpr int TP = BP / WP / 16;      // 16-pixel tiles per wave
  constexpr int TC = BC / WC_ / 16;
  constexpr int ROWS_W = TH / WP;       // tile rows per wave
  constexpr int BCL = BC < 64 ? 64 : BC;
  constexpr int W_LD = BCL / 64;        // DMA rounds per image (X or Y) and stage
  constexpr int HMAIN = 10 * 32 * 64, HSIDE = 4096, HBUF = HMAIN + HSIDE;
  constexpr int WIMG = BCL * 64, WST = 2 * WIMG, NST = 3;
  static_assert(WP * WC_ == 4 && TP == 2 * ROWS_W, "wave layout");
  static_assert(2 * HBUF + NST * WST <= 81920, "two workgroups per CU");
  __shared__ __attribute__((aligned(16))) char smem[2 * HBUF + NST * WST];
  char* const ring = smem + 2 * HBUF;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave / WC_, wc = wave % WC_;
  const int tiles_x = a.W / TW, tiles_y = a.H / TH;
  int m_tile, n_tile;
  if (!tile_of_block(a.B * tiles_y * tiles_x, a.Cout_pad / BC, m_tile, n_tile)) return;
  const int tx0 = (m_tile % tiles_x) * TW;
  const int ty0 = ((m_tile / tiles_x) % tiles_y) * TH;
  const int b = m_tile / (tiles_x * tiles_y);
  const int n0 = n_tile * BC;
  const float* zero = (const float*)g_zero_page;
  asm volatile("" : "+v"(zero));
  const float* ximg = (const float*)a.x + (long)b * a.H * a.W * a.in_stride;

  // ---- halo loader (as conv3x3_halo_kernel): 5 main pieces + 1 side piece per thread and chunk ----
  const int hslot = tid & 3, hpx = (tid >> 2) & 31, hr0 = tid >> 7;
  const int y0 = ty0 - 1 + hr0;
  const float* hp0 = ximg + ((long)y0 * a.W + tx0 + hpx) * a.in_stride + (hslot ^ swz(hpx)) * 4;
  const long row2 = 2L * a.W * a.in_stride;
  unsigned hmask = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) hmask |= (y0 + 2 * i >= 0 && y0 + 2 * i < a.H) ? (1u << i) : 0u;
  const float* hps;
  {
    const int side = (tid >> 2) & 1, hr = tid >> 3;   // [hr 0..9][side][slot], tid < 80
    const int y = ty0 - 1 + hr, x = side ? tx0 + TW : tx0 - 1;
    const bool ok = tid < 80 && y >= 0 && y < a.H && x >= 0 && x < a.W;
    hps = ximg + ((long)(ok ? y : 0) * a.W + (ok ? x : 0)) * a.in_stride + hslot * 4;
    hmask |= ok ? 32u : 0u;
  }
  const int lrow = tid >> 2;
  const int gw = hslot ^ swz(lrow);
  const float* wptr[W_LD];             // packed row (X image of step 0) of the cout this thread stages, + its k group
#pragma unroll
  for (int j = 0; j < W_LD; ++j) {
    const int L = lrow + 64 * j;
    const int Lw = L % (16 * TC), wv = L / (16 * TC);
    const int tt = Lw >> 4, r = Lw & 15;
    const int cl = wv * 16 * TC + cout_of<TC>(tt, r >> 2, r & 3);
    wptr[j] = (const float*)a.w + (long)(n0 + (L < BC ? cl : 0)) * a.Kpad + gw * 4;
  }
  auto issue_halo = [&](int chunk, int hb) {
    char* dst = smem + hb * HBUF + wave * 1024;
    const long coff = (long)chunk * 16;
#pragma unroll
    for (int i = 0; i < 5; ++i) dma16((hmask & (1u << i)) ? hp0 + i * row2 + coff : zero, dst + i * 4096);
    dma16((hmask & 32u) ? hps + coff : zero, smem + hb * HBUF + HMAIN + wave * 1024);
  };
  // K step kt = chunk * 5 + pair: 32 floats of every packed row = X (16) then Y (16)
  auto issue_w = [&](int kt, int st) {
#pragma unroll
    for (int j = 0; j < W_LD; ++j) {
      dma16(wptr[j] + (long)kt * 32, ring + st * WST + wave * 1024 + j * 4096);
      dma16(wptr[j] + (long)kt * 32 + 16, ring + st * WST + WIMG + wave * 1024 + j * 4096);
    }
  };
  // the thread's own six pieces of halo buffer hb: 4 f32 -> {hi[4], lo[4]} f16, in place
  auto convert = [&](int hb) {
    char* base = smem + hb * HBUF + wave * 1024 + lane * 16;
    f32x4 v[6];
#pragma unroll
    for (int i = 0; i < 5; ++i) v[i] = *(const f32x4*)(base + i * 4096);
    v[5] = *(const f32x4*)(base + HMAIN);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const f16x4 hi = __builtin_convertvector(v[i], f16x4);
      f32x4 r;
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = v[i][j] - (float)hi[j];
      const f16x4 lo = __builtin_convertvector(r, f16x4);
      *(f16x8*)(base + (i < 5 ? i * 4096 : HMAIN)) = __builtin_shufflevector(hi, lo, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };

  // ---- fragment addressing (as conv3x3_halo_kernel) ----
  const int l15 = lane & 15, kg = lane >> 4;
  const int row0 = wp * ROWS_W;
  int abase[2][3];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2) {
      const int X = 16 * e + l15 + s2 - 1;
      if (X < 0) abase[e][s2] = HMAIN + kg * 16 + row0 * 128;
      else if (X > 31) abase[e][s2] = HMAIN + 64 + kg * 16 + row0 * 128;
      else abase[e][s2] = X * 64 + ((kg ^ swz(X)) << 4) + row0 * 2048;
    }
  const int estride0 = (l15 == 0) ? 128 : 2048;    // row stride of this lane for (e=0, s=0)
  const int estride1 = (l15 == 15) ? 128 : 2048;   // ... for (e=1, s=2)
  const int fr_off = l15 * 64 + ((kg ^ swz(l15)) << 4);
  const char* fragB = ring + (wc * 16 * TC) * 64 + fr_off;

  f32x4 acc[TP][TC];
#pragma unroll
  for (int p = 0; p < TP; ++p)
#pragma unroll
    for (int c = 0; c < TC; ++c) acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  typedef unsigned long long u64;
  typedef u64 u64x2 __attribute__((ext_vector_type(2)));
  // window offset of this lane's fragment of pixel tile p for tap T
  auto tap_off = [&](int p, int T) {
    const int R_ = T / 3, S_ = T % 3;
    const int e = p & 1, lr = p >> 1;
    if (e == 0 && S_ == 0) return abase[0][0] + (lr + R_) * estride0;
    if (e == 1 && S_ == 2) return abase[1][2] + (lr + R_) * estride1;
    return abase[e][S_] + (lr + R_) * 2048;
  };
  // operands of a K step: X / Y weight fragments (two register sets) and H / L pixel fragments
  static_assert(TP == 4, "two pixel-tile pairs per wave");
  typedef __attribute__((address_space(3))) const volatile u64 lds_u64;
  f16x8 xf[2][TC], yf[2][TC];
  u64x2 hq[TP], lq[TP];
  // weight fragment j of ring stage st into register set PAR: j = 2 * c + (0: X, 1: Y)
  auto load_w1 = [&](auto parc, int st, int j) {
    constexpr int PAR = decltype(parc)::value;
    const int c = j >> 1;
    if (j & 1) yf[PAR][c] = *(const f16x8*)(fragB + st * WST + WIMG + c * 1024);
    else xf[PAR][c] = *(const f16x8*)(fragB + st * WST + c * 1024);
  };
  // piece q of the pixel fragments of tile p for pair PR (taps 2*PR, 2*PR+1; tap 9: zero weights, tap 8's pixels) from halo
  // buffer hb: q = 0, 1 the hi halves of the two taps (H), q = 2, 3 the lo halves (L).  Separate 8-byte reads (volatile: not
  // merged into one 16-byte read of a tap whose halves then have to be moved apart, with a wait for the data in the
  // middle of the MFMA stream)
  auto load_px1 = [&](int p, auto prc, int hb, int q) {
    constexpr int PR = decltype(prc)::value;
    constexpr int T0 = 2 * PR, T1 = PR == 4 ? 8 : 2 * PR + 1;
    const char* src = smem + hb * HBUF + tap_off(p, (q & 1) ? T1 : T0) + (q >> 1) * 8;
    if (q >> 1) lq[p][q & 1] = *(lds_u64*)src;
    else hq[p][q & 1] = *(lds_u64*)src;
  };
  // MFMA i (0 .. 6 * TC) of the tile pair (p, p + 1): three groups of 2 * TC -- kinds[g] = 0: X.H, 1: Y.H, 2: X.L -- each
  // over (tile u, cout tile c); consecutive MFMAs write different accumulators, the three that accumulate into one tile are
  // 2 * TC instructions apart
  auto mfma1 = [&](auto parc, int p, int i, int k0, int k1, int k2) {
    constexpr int PAR = decltype(parc)::value;
    const int g = i / (2 * TC), w = i % (2 * TC), u = w / TC, c = w % TC;
    const int kind = g == 0 ? k0 : (g == 1 ? k1 : k2);
    const f16x8 wfrag = kind == 1 ? yf[PAR][c] : xf[PAR][c];
    const f16x8 pfrag = __builtin_bit_cast(f16x8, kind == 2 ? lq[p + u] : hq[p + u]);
    acc[p + u][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfrag, pfrag, acc[p + u][c], 0, 0, 0);
  };

  const int nch = a.Cin / 16, nk = nch * 5;      // nk >= 5: the three prologue stages exist
  issue_halo(0, 0);
  issue_w(0, 0);
  issue_w(1, 1);
  issue_w(2, 2);
  wait_vmcnt<6 * W_LD>();               // the window of chunk 0 (older than the weight stages)
  convert(0);
  wait_vmcnt<4 * W_LD>();               // weights(0)
  wait_lgkm0();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int j = 0; j < 2 * TC; ++j) load_w1(std::integral_constant<int, 0>{}, 0, j);
#pragma unroll
  for (int p = 0; p < TP; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) load_px1(p, std::integral_constant<int, 0>{}, 0, q);

  // K step kt (pair PR of `chunk`, ring stage st = kt % 3, weight fragments in register set PAR).  Behind one barrier, the
  // step's 12 * TC MFMAs in 4 * TC slots of three; in front of every slot ONE or two LDS reads of the next step's operands,
  // pinned there with sched_barriers -- the eight weight reads of a step issued in one burst behind the barrier (eight waves
  // of the CU at the same moment: 64 ds_read_b128 = 256 LDS cycles) fill the LDS queue, and a wave whose read is not
  // accepted cannot issue the MFMAs behind it either.  Slot plan (TC = 4): tile pair (0, 1): the 2 * TC weight fragments
  // of step kt + 1 into the other register set; tile pair (2, 3), whose products run X.L first: the pixel fragments of tiles
  // 0 and 1 (their registers are free), then the L fragments of tiles 2 and 3; behind the last MFMA only the four H reads
  // of tiles 2 and 3 are left.  The DMAs of step kt + 3 (into the stage this step's weights came from) go out between the
  // pairs.  The operand reads are unconditional (behind the last step they fetch stale LDS that nothing uses); only the
  // DMA issue and the wait counts know about the last chunk, through scalar branches outside the MFMA slots.
  auto kstep = [&](int kt, int chunk, int st, auto prc, auto hbc, auto parc) {
    constexpr int PR = decltype(prc)::value, HB = decltype(hbc)::value, PAR = decltype(parc)::value;
    using NextPR = std::integral_constant<int, (PR + 1) % 5>;
    using NextPar = std::integral_constant<int, PAR ^ 1>;
    constexpr int NHB = PR == 4 ? (HB ^ 1) : HB;
    const bool last = chunk + 1 == nch;
    // weights(kt + 1) have landed once only what was queued behind them is still in flight: weights(kt + 2) and, in the
    // step after a halo prefetch (PR == 1), its 6 pieces
    if (!last) { if (PR == 1) wait_vmcnt<2 * W_LD + 6>(); else wait_vmcnt<2 * W_LD>(); }
    else if (PR < 3) wait_vmcnt<2 * W_LD>();
    else wait_vmcnt<0>();
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int nst = st == 2 ? 0 : st + 1;
    constexpr int SLOTS = 2 * TC;
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
      load_w1(NextPar{}, nst, sl);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 3 * sl; i < 3 * sl + 3; ++i) mfma1(parc, 0, i, 0, 1, 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (PR == 0 && !last) issue_halo(chunk + 1, HB ^ 1);
    if (!last || PR < 2) issue_w(kt + 3, st);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
      if (sl < TC) {                 // 8 reads for tiles 0, 1 over the first TC slots
#pragma unroll
        for (int r = sl * (8 / TC); r < (sl + 1) * (8 / TC); ++r) load_px1(r >> 2, NextPR{}, NHB, r & 3);
      } else {                       // the 4 L reads of tiles 2, 3 over the other TC slots (X.L of this pair ran first)
#pragma unroll
        for (int r = (sl - TC) * 4 / TC; r < (sl - TC + 1) * 4 / TC; ++r) load_px1(2 + (r >> 1), NextPR{}, NHB, 2 + (r & 1));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 3 * sl; i < 3 * sl + 3; ++i) mfma1(parc, 2, i, 2, 0, 1);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) load_px1(2 + (r >> 1), NextPR{}, NHB, r & 1);
    // the next chunk's window has landed (this step's wait left only weight stages in flight): split this thread's own
    // pieces of it; the barrier of step PR == 3 publishes them, the end of step PR == 4 reads the first operands from them
    if (PR == 2 && !last) convert(HB ^ 1);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto next_stage = [](int st) { return st == NST - 1 ? 0 : st + 1; };
  // the five steps of a chunk; P0 = weight register set of its first step (a chunk is an odd number of steps, so the two
  // chunks of the loop body start on opposite sets)
  auto chunk_steps = [&](int chunk, int st0, auto hbc, auto p0c) {
    constexpr int P0 = decltype(p0c)::value;
    const int kt = chunk * 5;
    int st = st0;
    kstep(kt + 0, chunk, st, std::integral_constant<int, 0>{}, hbc, std::integral_constant<int, P0>{}); st = next_stage(st);
    kstep(kt + 1, chunk, st, std::integral_constant<int, 1>{}, hbc, std::integral_constant<int, P0 ^ 1>{}); st = next_stage(st);
    kstep(kt + 2, chunk, st, std::integral_constant<int, 2>{}, hbc, std::integral_constant<int, P0>{}); st = next_stage(st);
    kstep(kt + 3, chunk, st, std::integral_constant<int, 3>{}, hbc, std::integral_constant<int, P0 ^ 1>{}); st = next_stage(st);
    kstep(kt + 4, chunk, st, std::integral_constant<int, 4>{}, hbc, std::integral_constant<int, P0>{});
  };
  int chunk = 0, st0 = 0;                 // st0 = (chunk * 5) % 3
  for (; chunk + 1 < nch; chunk += 2) {
    chunk_steps(chunk, st0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    st0 = (st0 + 2) % 3;
    chunk_steps(chunk + 1, st0, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
    st0 = (st0 + 2) % 3;
  }
  if (chunk < nch) chunk_steps(chunk, st0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});

  const int q = lane >> 4;
  const int cb = n0 + wc * 16 * TC;
#pragma unroll
  for (int p = 0; p < TP; ++p) {
    const int y = ty0 + row0 + (p >> 1), x = tx0 + 16 * (p & 1) + l15;
    const int m = (b * a.H + y) * a.W + x;
    epilogue_tiles<float, TC>(a, m, cb, q, acc[p]);
  }
}

// ------------------------------------------------------------------------------------------
// korder 3: the pair kernel above spends a tenth tap of zeros on every 16-channel chunk (nine taps = four pairs and a half).
// Here tap 8 of a chunk is paired with tap 8 of the NEXT chunk: two chunks = nine steps instead of ten (-10 % MFMAs and
// operand reads).  Both windows of a chunk pair sit in the two halo buffers at once (A = even chunk in buffer 0, B = odd
// chunk in buffer 1); weights: ops.PackedConv._pack_pairs, nine 128-byte {X, Y} steps per chunk pair.  Cin % 32 == 0.
// ------------------------------------------------------------------------------------------
// BC = 128 (round 4, CTDET_TUNE_PAIR2_128): one workgroup per CU, a wave then has the whole 512-register file -- 128 accumulators
// (AGPRs) next to both operand sets -- and every pixel fragment read feeds twice the MFMAs.
template <int BC, int WP, int WC_, int TW = 32>
__global__ void __launch_bounds__(256, BC > 64 ? 1 : 2) conv3x3_halo_pair2_kernel(const ConvArgs a) {
  // TW = 32: 8 x 32-pixel tiles; TW = 16: 16 x 16 (maps whose width is not a multiple of 32: the 16 x 16 level of DLA-34 at 512^2)
  constexpr int TH = 256 / TW, BP = TH * TW;
  constexpr int EN = TW / 16;           // 16-pixel tiles per tile row
  constexpr int RS = TW * 64;           // LDS bytes per window row
  constexpr int RPR = 256 * 16 / RS;    // window rows a DMA round of the workgroup covers (2 / 4)
  constexpr int TP = BP / WP / 16;      // 16-pixel tiles per wave
  constexpr int TC = BC / WC_ / 16;
  constexpr int ROWS_W = TH / WP;       // tile rows per wave
  constexpr int BCL = BC < 64 ? 64 : BC;
  constexpr int W_LD = BCL / 64;        // DMA rounds per image (X or Y) and stage
  constexpr int HMAIN = 10 * 32 * 64, HSIDE = 4096, HBUF = HMAIN + HSIDE;
  constexpr int WIMG = BCL * 64, WST = 2 * WIMG, NST = 3;
  static_assert(WP * WC_ == 4 && TP == EN * ROWS_W && (TH + 2 + RPR - 1) / RPR == 5, "wave layout");
  static_assert(BC > 64 || 2 * HBUF + NST * WST <= 81920, "two workgroups per CU");
  __shared__ __attribute__((aligned(16))) char smem[2 * HBUF + NST * WST];
  char* const ring = smem + 2 * HBUF;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave / WC_, wc = wave % WC_;
  const int tiles_x = a.W / TW, tiles_y = a.H / TH;
  int m_tile, n_tile;
  if (!tile_of_block(a.B * tiles_y * tiles_x, a.Cout_pad / BC, m_tile, n_tile)) return;
  const int tx0 = (m_tile % tiles_x) * TW;
  const int ty0 = ((m_tile / tiles_x) % tiles_y) * TH;
  const int b = m_tile / (tiles_x * tiles_y);
  const int n0 = n_tile * BC;
  const float* zero = (const float*)g_zero_page;
  asm volatile("" : "+v"(zero));
  const float* ximg = (const float*)a.x + (long)b * a.H * a.W * a.in_stride;

  // ---- halo loader (as conv3x3_halo_kernel): 5 main pieces + 1 side piece per thread and chunk ----
  const int hslot = tid & 3, hpx = (tid >> 2) & (TW - 1), hr0 = tid / (4 * TW);
  const int y0 = ty0 - 1 + hr0;
  const float* hp0 = ximg + ((long)y0 * a.W + tx0 + hpx) * a.in_stride + (hslot ^ swz(hpx)) * 4;
  const long row2 = (long)RPR * a.W * a.in_stride;
  unsigned hmask = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) hmask |= (hr0 + RPR * i < TH + 2 && y0 + RPR * i >= 0 && y0 + RPR * i < a.H) ? (1u << i) : 0u;
  const float* hps;
  {
    const int side = (tid >> 2) & 1, hr = tid >> 3;   // [hr 0..TH+1][side][slot], tid < 8 * (TH + 2)
    const int y = ty0 - 1 + hr, x = side ? tx0 + TW : tx0 - 1;
    const bool ok = tid < 8 * (TH + 2) && y >= 0 && y < a.H && x >= 0 && x < a.W;
    hps = ximg + ((long)(ok ? y : 0) * a.W + (ok ? x : 0)) * a.in_stride + hslot * 4;
    hmask |= ok ? 32u : 0u;
  }
  const int lrow = tid >> 2;
  const int gw = hslot ^ swz(lrow);
  const float* wptr[W_LD];             // packed row (X image of step 0) of the cout this thread stages, + its k group
#pragma unroll
  for (int j = 0; j < W_LD; ++j) {
    const int L = lrow + 64 * j;
    const int Lw = L % (16 * TC), wv = L / (16 * TC);
    const int tt = Lw >> 4, r = Lw & 15;
    const int cl = wv * 16 * TC + cout_of<TC>(tt, r >> 2, r & 3);
    wptr[j] = (const float*)a.w + (long)(n0 + (L < BC ? cl : 0)) * a.Kpad + gw * 4;
  }
  auto issue_halo = [&](int chunk, int hb) {
    char* dst = smem + hb * HBUF + wave * 1024;
    const long coff = (long)chunk * 16;
#pragma unroll
    for (int i = 0; i < 5; ++i) dma16((hmask & (1u << i)) ? hp0 + i * row2 + coff : zero, dst + i * 4096);
    dma16((hmask & 32u) ? hps + coff : zero, smem + hb * HBUF + HMAIN + wave * 1024);
  };
  // K step kt = chunk * 5 + pair: 32 floats of every packed row = X (16) then Y (16)
  auto issue_w = [&](int kt, int st) {
#pragma unroll
    for (int j = 0; j < W_LD; ++j) {
      dma16(wptr[j] + (long)kt * 32, ring + st * WST + wave * 1024 + j * 4096);
      dma16(wptr[j] + (long)kt * 32 + 16, ring + st * WST + WIMG + wave * 1024 + j * 4096);
    }
  };
  // the thread's own six pieces of halo buffer hb: 4 f32 -> {hi[4], lo[4]} f16, in place
  auto convert = [&](int hb) {
    char* base = smem + hb * HBUF + wave * 1024 + lane * 16;
    f32x4 v[6];
#pragma unroll
    for (int i = 0; i < 5; ++i) v[i] = *(const f32x4*)(base + i * 4096);
    v[5] = *(const f32x4*)(base + HMAIN);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const f16x4 hi = __builtin_convertvector(v[i], f16x4);
      f32x4 r;
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = v[i][j] - (float)hi[j];
      const f16x4 lo = __builtin_convertvector(r, f16x4);
      *(f16x8*)(base + (i < 5 ? i * 4096 : HMAIN)) = __builtin_shufflevector(hi, lo, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };

  // ---- fragment addressing (as conv3x3_halo_kernel) ----
  const int l15 = lane & 15, kg = lane >> 4;
  const int row0 = wp * ROWS_W;
  int abase[EN][3];
#pragma unroll
  for (int e = 0; e < EN; ++e)
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2) {
      const int X = 16 * e + l15 + s2 - 1;
      if (X < 0) abase[e][s2] = HMAIN + kg * 16 + row0 * 128;
      else if (X > TW - 1) abase[e][s2] = HMAIN + 64 + kg * 16 + row0 * 128;
      else abase[e][s2] = X * 64 + ((kg ^ swz(X)) << 4) + row0 * RS;
    }
  const int estride0 = (l15 == 0) ? 128 : RS;      // row stride of this lane for (e = 0, s = 0)
  const int estride1 = (l15 == 15) ? 128 : RS;     // ... for (e = EN - 1, s = 2)
  const int fr_off = l15 * 64 + ((kg ^ swz(l15)) << 4);
  const char* fragB = ring + (wc * 16 * TC) * 64 + fr_off;

  f32x4 acc[TP][TC];
#pragma unroll
  for (int p = 0; p < TP; ++p)
#pragma unroll
    for (int c = 0; c < TC; ++c) acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  typedef unsigned long long u64;
  typedef u64 u64x2 __attribute__((ext_vector_type(2)));
  // window offset of this lane's fragment of pixel tile p for tap T
  auto tap_off = [&](int p, int T) {
    const int R_ = T / 3, S_ = T % 3;
    const int e = p % EN, lr = p / EN;
    if (e == 0 && S_ == 0) return abase[0][0] + (lr + R_) * estride0;
    if (e == EN - 1 && S_ == 2) return abase[EN - 1][2] + (lr + R_) * estride1;
    return abase[e][S_] + (lr + R_) * RS;
  };
  // operands of a K step: X / Y weight fragments (two register sets) and H / L pixel fragments
  static_assert(TP == 4, "two pixel-tile pairs per wave");
  typedef __attribute__((address_space(3))) const volatile u64 lds_u64;
  f16x8 xf[2][TC], yf[2][TC];
  u64x2 hq[TP], lq[TP];
  // weight fragment j of ring stage st into register set PAR: j = 2 * c + (0: X, 1: Y)
  auto load_w1 = [&](auto parc, int st, int j) {
    constexpr int PAR = decltype(parc)::value;
    const int c = j >> 1;
    if (j & 1) yf[PAR][c] = *(const f16x8*)(fragB + st * WST + WIMG + c * 1024);
    else xf[PAR][c] = *(const f16x8*)(fragB + st * WST + c * 1024);
  };
  // step S of a chunk pair (A in halo buffer 0, B in buffer 1) multiplies two (buffer, tap) operands: S = 0..3 taps (2S, 2S+1)
  // of A, S = 4 tap 8 of A and tap 8 of B, S = 5..8 taps (2(S-5), 2(S-5)+1) of B.  Piece q of the pixel fragments of tile p
  // for step S: q = 0, 1 the hi halves of the two operands (H), q = 2, 3 the lo halves (L).  Separate 8-byte reads
  // (volatile: not merged into one 16-byte read whose halves then have to be moved apart, with a wait for the data in the
  // middle of the MFMA stream)
  auto load_px1 = [&](int p, auto sc, int q) {
    constexpr int S = decltype(sc)::value;
    constexpr int B0 = S <= 4 ? 0 : 1, B1 = S < 4 ? 0 : 1;
    constexpr int T0 = S < 4 ? 2 * S : (S == 4 ? 8 : 2 * (S - 5)), T1 = S < 4 ? 2 * S + 1 : (S == 4 ? 8 : 2 * (S - 5) + 1);
    const char* src = smem + ((q & 1) ? B1 : B0) * HBUF + tap_off(p, (q & 1) ? T1 : T0) + (q >> 1) * 8;
    if (q >> 1) lq[p][q & 1] = *(lds_u64*)src;
    else hq[p][q & 1] = *(lds_u64*)src;
  };
  // MFMA i (0 .. 6 * TC) of the tile pair (p, p + 1): three groups of 2 * TC -- kinds[g] = 0: X.H, 1: Y.H, 2: X.L -- each
  // over (tile u, cout tile c); consecutive MFMAs write different accumulators, the three that accumulate into one tile are
  // 2 * TC instructions apart
  auto mfma1 = [&](auto parc, int p, int i, int k0, int k1, int k2) {
    constexpr int PAR = decltype(parc)::value;
    const int g = i / (2 * TC), w = i % (2 * TC), u = w / TC, c = w % TC;
    const int kind = g == 0 ? k0 : (g == 1 ? k1 : k2);
    const f16x8 wfrag = kind == 1 ? yf[PAR][c] : xf[PAR][c];
    const f16x8 pfrag = __builtin_bit_cast(f16x8, kind == 2 ? lq[p + u] : hq[p + u]);
    acc[p + u][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfrag, pfrag, acc[p + u][c], 0, 0, 0);
  };

  const int npair = a.Cin / 32, nk = npair * 9;
  using I0 = std::integral_constant<int, 0>;
  issue_halo(0, 0);
  issue_halo(1, 1);
  issue_w(0, 0);
  issue_w(1, 1);
  issue_w(2, 2);
  wait_vmcnt<6 * W_LD>();               // both windows of pair 0 (older than the weight stages)
  convert(0);
  convert(1);
  wait_vmcnt<4 * W_LD>();               // weights(0)
  wait_lgkm0();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int j = 0; j < 2 * TC; ++j) load_w1(std::integral_constant<int, 1>{}, 0, j);     // set 1: step 0 moves it to set 0
#pragma unroll
  for (int p = 0; p < TP; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) load_px1(p, I0{}, q);

  // K step kt = 9 * pair + S (ring stage S % 3, weight fragments in register set S & 1): the slot plan of
  // conv3x3_halo_pair_kernel.  Window traffic: B of this pair goes out in step 0 (buffer 1 was last read for step 8 of the
  // previous pair), is split at the end of step 2 and first read -- for step 4 -- during step 3; A of the next pair goes out in
  // step 5 (buffer 0 was last read for step 4), is split at the end of step 7 and first read during step 8.
  auto kstep = [&](int kt, int pair, auto sc) {
    constexpr int S = decltype(sc)::value, PAR = S & 1, ST = S % 3, NST = (S + 1) % 3;
    using NextS = std::integral_constant<int, (S + 1) % 9>;
    using Par = std::integral_constant<int, PAR>;
    using NextPar = std::integral_constant<int, PAR ^ 1>;
    const bool last = pair + 1 == npair;
    // weights(kt + 1) have landed once only what was queued behind them is still in flight: weights(kt + 2) and, in the
    // step after a window prefetch, its 6 pieces
    if (last && S >= 7) wait_vmcnt<0>();
    else if ((S == 1 && pair > 0) || (S == 6 && !last)) wait_vmcnt<2 * W_LD + 6>();
    else wait_vmcnt<2 * W_LD>();
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (S == 0) {      // nine steps per pair: the fragments step 8 (or the prologue) fetched sit in set 1
#pragma unroll
      for (int c = 0; c < TC; ++c) { xf[0][c] = xf[1][c]; yf[0][c] = yf[1][c]; }
    }
    constexpr int SLOTS = 2 * TC;
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
      load_w1(NextPar{}, NST, sl);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 3 * sl; i < 3 * sl + 3; ++i) mfma1(Par{}, 0, i, 0, 1, 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (S == 0 && pair > 0) issue_halo(2 * pair + 1, 1);
    if (S == 5 && !last) issue_halo(2 * pair + 2, 0);
    if (kt + 3 < nk) issue_w(kt + 3, ST);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
      if (sl < TC) {                 // 8 reads for tiles 0, 1 over the first TC slots
#pragma unroll
        for (int r = sl * (8 / TC); r < (sl + 1) * (8 / TC); ++r) load_px1(r >> 2, NextS{}, r & 3);
      } else {                       // the 4 L reads of tiles 2, 3 over the other TC slots (X.L of this pair ran first)
#pragma unroll
        for (int r = (sl - TC) * 4 / TC; r < (sl - TC + 1) * 4 / TC; ++r) load_px1(2 + (r >> 1), NextS{}, 2 + (r & 1));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 3 * sl; i < 3 * sl + 3; ++i) mfma1(Par{}, 2, i, 2, 0, 1);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) load_px1(2 + (r >> 1), NextS{}, r & 1);
    if (S == 2 && pair > 0) convert(1);
    if (S == 7 && !last) convert(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int pair = 0; pair < npair; ++pair) {
    const int kt = pair * 9;
    kstep(kt + 0, pair, std::integral_constant<int, 0>{});
    kstep(kt + 1, pair, std::integral_constant<int, 1>{});
    kstep(kt + 2, pair, std::integral_constant<int, 2>{});
    kstep(kt + 3, pair, std::integral_constant<int, 3>{});
    kstep(kt + 4, pair, std::integral_constant<int, 4>{});
    kstep(kt + 5, pair, std::integral_constant<int, 5>{});
    kstep(kt + 6, pair, std::integral_constant<int, 6>{});
    kstep(kt + 7, pair, std::integral_constant<int, 7>{});
    kstep(kt + 8, pair, std::integral_constant<int, 8>{});
  }

  const int q = lane >> 4;
  const int cb = n0 + wc * 16 * TC;
#pragma unroll
  for (int p = 0; p < TP; ++p) {
    const int y = ty0 + row0 + p / EN, x = tx0 + 16 * (p % EN) + l15;
    const int m = (b * a.H + y) * a.W + x;
    epilogue_tiles<float, TC>(a, m, cb, q, acc[p]);
  }
}

// ------------------------------------------------------------------------------------------
// f16 form of the kernel above (the per-tap cadence of conv3x3_halo_kernel -- a barrier, the DMA issue, the
// operand reads and an s_waitcnt in front of every 16..32 MFMAs): two taps per K step and barrier (nine steps per pair of
// 32-channel chunks, tap 8 of a chunk with tap 8 of the next), the next step's operand reads pinned between this step's
// MFMAs, weight fragments double-buffered in registers.  Same LDS images; weights are the chunk-major rows (korder 1)
// the per-tap kernel reads.  Cin % 64 == 0.
// ------------------------------------------------------------------------------------------
template <int BC, int WP, int WC_, typename TOut, int TW = 32>
__global__ void __launch_bounds__(256, 2) conv3x3_halo_tap2_kernel(const ConvArgs a) {
  constexpr int TH = 256 / TW, BP = TH * TW;     // TW = 32: 8 x 32-pixel tiles; TW = 16: 16 x 16 (see conv3x3_halo_pair2_kernel)
  constexpr int EN = TW / 16, RS = TW * 64, RPR = 256 * 16 / RS;
  constexpr int TP = BP / WP / 16;      // 16-pixel tiles per wave
  constexpr int TC = BC / WC_ / 16;
  constexpr int ROWS_W = TH / WP;       // tile rows per wave
  constexpr int BCL = BC < 64 ? 64 : BC;
  constexpr int W_LD = BCL / 64;        // DMA rounds per image (X or Y) and stage
  constexpr int HMAIN = 10 * 32 * 64, HSIDE = 4096, HBUF = HMAIN + HSIDE;
  constexpr int WIMG = BCL * 64, WST = 2 * WIMG, NST = 3;
  static_assert(WP * WC_ == 4 && TP == EN * ROWS_W && (TH + 2 + RPR - 1) / RPR == 5, "wave layout");
  static_assert(2 * HBUF + NST * WST <= 81920, "two workgroups per CU");
  __shared__ __attribute__((aligned(16))) char smem[2 * HBUF + NST * WST];
  char* const ring = smem + 2 * HBUF;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave / WC_, wc = wave % WC_;
  const int tiles_x = a.W / TW, tiles_y = a.H / TH;
  int m_tile, n_tile;
  if (!tile_of_block(a.B * tiles_y * tiles_x, a.Cout_pad / BC, m_tile, n_tile)) return;
  const int tx0 = (m_tile % tiles_x) * TW;
  const int ty0 = ((m_tile / tiles_x) % tiles_y) * TH;
  const int b = m_tile / (tiles_x * tiles_y);
  const int n0 = n_tile * BC;
  const f16* zero = (const f16*)g_zero_page;
  asm volatile("" : "+v"(zero));
  const f16* ximg = (const f16*)a.x + (long)b * a.H * a.W * a.in_stride;

  // ---- halo loader (as conv3x3_halo_kernel): 5 main pieces + 1 side piece per thread and chunk ----
  const int hslot = tid & 3, hpx = (tid >> 2) & (TW - 1), hr0 = tid / (4 * TW);
  const int y0 = ty0 - 1 + hr0;
  const f16* hp0 = ximg + ((long)y0 * a.W + tx0 + hpx) * a.in_stride + (hslot ^ swz(hpx)) * 8;
  const long row2 = (long)RPR * a.W * a.in_stride;
  unsigned hmask = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) hmask |= (hr0 + RPR * i < TH + 2 && y0 + RPR * i >= 0 && y0 + RPR * i < a.H) ? (1u << i) : 0u;
  const f16* hps;
  {
    const int side = (tid >> 2) & 1, hr = tid >> 3;   // [hr 0..9][side][slot], tid < 80
    const int y = ty0 - 1 + hr, x = side ? tx0 + TW : tx0 - 1;
    const bool ok = tid < 8 * (TH + 2) && y >= 0 && y < a.H && x >= 0 && x < a.W;
    hps = ximg + ((long)(ok ? y : 0) * a.W + (ok ? x : 0)) * a.in_stride + hslot * 8;
    hmask |= ok ? 32u : 0u;
  }
  const int lrow = tid >> 2;
  const int gw = hslot ^ swz(lrow);
  const f16* wptr[W_LD];               // chunk-major packed row of the cout this thread stages, + its k group
#pragma unroll
  for (int j = 0; j < W_LD; ++j) {
    const int L = lrow + 64 * j;
    const int Lw = L % (16 * TC), wv = L / (16 * TC);
    const int tt = Lw >> 4, r = Lw & 15;
    const int cl = wv * 16 * TC + cout_of<TC>(tt, r >> 2, r & 3);
    wptr[j] = (const f16*)a.w + (long)(n0 + (L < BC ? cl : 0)) * a.Kpad + gw * 8;
  }
  auto issue_halo = [&](int chunk, int hb) {
    char* dst = smem + hb * HBUF + wave * 1024;
    const long coff = (long)chunk * 32;
#pragma unroll
    for (int i = 0; i < 5; ++i) dma16((hmask & (1u << i)) ? hp0 + i * row2 + coff : zero, dst + i * 4096);
    dma16((hmask & 32u) ? hps + coff : zero, smem + hb * HBUF + HMAIN + wave * 1024);
  };
  // K step kt = 9 * pair + S: the 64-byte pieces (chunk, tap) of its two operands from the chunk-major rows (korder 1: piece
  // index chunk * 9 + tap) -- X = (A, 2S) / (A, 8) / (B, 2(S-5)), Y = (A, 2S+1) / (B, 8) / (B, 2(S-5)+1)
  auto issue_w = [&](int kt, int st) {
    const int pair = kt / 9, S = kt - 9 * pair;
    const int kx = 18 * pair + (S < 4 ? 2 * S : (S == 4 ? 8 : 9 + 2 * (S - 5)));
    const int ky = S == 4 ? 18 * pair + 17 : kx + 1;
#pragma unroll
    for (int j = 0; j < W_LD; ++j) {
      dma16(wptr[j] + (long)kx * 32, ring + st * WST + wave * 1024 + j * 4096);
      dma16(wptr[j] + (long)ky * 32, ring + st * WST + WIMG + wave * 1024 + j * 4096);
    }
  };

  // ---- fragment addressing (as conv3x3_halo_kernel) ----
  const int l15 = lane & 15, kg = lane >> 4;
  const int row0 = wp * ROWS_W;
  int abase[EN][3];
#pragma unroll
  for (int e = 0; e < EN; ++e)
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2) {
      const int X = 16 * e + l15 + s2 - 1;
      if (X < 0) abase[e][s2] = HMAIN + kg * 16 + row0 * 128;
      else if (X > TW - 1) abase[e][s2] = HMAIN + 64 + kg * 16 + row0 * 128;
      else abase[e][s2] = X * 64 + ((kg ^ swz(X)) << 4) + row0 * RS;
    }
  const int estride0 = (l15 == 0) ? 128 : RS;      // row stride of this lane for (e = 0, s = 0)
  const int estride1 = (l15 == 15) ? 128 : RS;     // ... for (e = EN - 1, s = 2)
  const int fr_off = l15 * 64 + ((kg ^ swz(l15)) << 4);
  const char* fragB = ring + (wc * 16 * TC) * 64 + fr_off;

  f32x4 acc[TP][TC];
#pragma unroll
  for (int p = 0; p < TP; ++p)
#pragma unroll
    for (int c = 0; c < TC; ++c) acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // window offset of this lane's fragment of pixel tile p for tap T
  auto tap_off = [&](int p, int T) {
    const int R_ = T / 3, S_ = T % 3;
    const int e = p % EN, lr = p / EN;
    if (e == 0 && S_ == 0) return abase[0][0] + (lr + R_) * estride0;
    if (e == EN - 1 && S_ == 2) return abase[EN - 1][2] + (lr + R_) * estride1;
    return abase[e][S_] + (lr + R_) * RS;
  };
  // operands of a K step: X / Y weight fragments (two register sets) and the pixel fragments of its two (buffer, tap) operands
  static_assert(TP == 4, "two pixel-tile pairs per wave");
  f16x8 xf[2][TC], yf[2][TC], pf[TP][2];
  // weight fragment j of ring stage st into register set PAR: j = 2 * c + (0: X, 1: Y)
  auto load_w1 = [&](auto parc, int st, int j) {
    constexpr int PAR = decltype(parc)::value;
    const int c = j >> 1;
    if (j & 1) yf[PAR][c] = *(const f16x8*)(fragB + st * WST + WIMG + c * 1024);
    else xf[PAR][c] = *(const f16x8*)(fragB + st * WST + c * 1024);
  };
  // step S of a chunk pair (A in halo buffer 0, B in buffer 1): operand q = 0 is (A, 2S) / (A, 8) / (B, 2(S-5)), q = 1 is
  // (A, 2S+1) / (B, 8) / (B, 2(S-5)+1) for S = 0..3 / 4 / 5..8; the fragment of tile p: 8 channels of 16 pixels
  auto load_px1 = [&](int p, auto sc, int q) {
    constexpr int S = decltype(sc)::value;
    constexpr int B0 = S <= 4 ? 0 : 1, B1 = S < 4 ? 0 : 1;
    constexpr int T0 = S < 4 ? 2 * S : (S == 4 ? 8 : 2 * (S - 5)), T1 = S < 4 ? 2 * S + 1 : (S == 4 ? 8 : 2 * (S - 5) + 1);
    pf[p][q] = *(const f16x8*)(smem + (q ? B1 : B0) * HBUF + tap_off(p, q ? T1 : T0));
  };
  // MFMA i (0 .. 4 * TC) of the tile pair (p, p + 1): two groups of 2 * TC -- kind 0: X . operand 0, kind 1: Y . operand 1 --
  // each over (tile u, cout tile c)
  auto mfma1 = [&](auto parc, int p, int i, int k0, int k1) {
    constexpr int PAR = decltype(parc)::value;
    const int g = i / (2 * TC), w = i % (2 * TC), u = w / TC, c = w % TC;
    const int kind = g == 0 ? k0 : k1;
    acc[p + u][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kind ? yf[PAR][c] : xf[PAR][c], pf[p + u][kind], acc[p + u][c], 0, 0, 0);
  };

  const int npair = a.Cin / 64, nk = npair * 9;      // a chunk = 32 f16 channels
  using I0 = std::integral_constant<int, 0>;
  issue_halo(0, 0);
  issue_halo(1, 1);
  issue_w(0, 0);
  issue_w(1, 1);
  issue_w(2, 2);
  wait_vmcnt<4 * W_LD>();               // weights(0), and both windows of pair 0 in front of them
  wait_lgkm0();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int j = 0; j < 2 * TC; ++j) load_w1(std::integral_constant<int, 1>{}, 0, j);     // set 1: step 0 moves it to set 0
#pragma unroll
  for (int p = 0; p < TP; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q) load_px1(p, I0{}, q);

  // K step kt = 9 * pair + S (ring stage S % 3, weight fragments in register set S & 1), the slot plan of
  // conv3x3_halo_pair_kernel with two MFMAs per slot.  Window traffic: B of this pair goes out in step 0 (buffer 1 was last
  // read for step 8 of the previous pair) and is first read -- for step 4 -- during step 3; A of the next pair goes out in
  // step 5 (buffer 0 was last read for step 4) and is first read during step 8.
  auto kstep = [&](int kt, int pair, auto sc) {
    constexpr int S = decltype(sc)::value, PAR = S & 1, ST = S % 3, NST = (S + 1) % 3;
    using NextS = std::integral_constant<int, (S + 1) % 9>;
    using Par = std::integral_constant<int, PAR>;
    using NextPar = std::integral_constant<int, PAR ^ 1>;
    const bool last = pair + 1 == npair;
    // weights(kt + 1) have landed once only what was queued behind them is still in flight: weights(kt + 2) and, in the
    // step after a window prefetch, its 6 pieces
    if (last && S >= 7) wait_vmcnt<0>();
    else if ((S == 1 && pair > 0) || (S == 6 && !last)) wait_vmcnt<2 * W_LD + 6>();
    else wait_vmcnt<2 * W_LD>();
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (S == 0) {      // nine steps per pair: the fragments step 8 (or the prologue) fetched sit in set 1
#pragma unroll
      for (int c = 0; c < TC; ++c) { xf[0][c] = xf[1][c]; yf[0][c] = yf[1][c]; }
    }
    constexpr int SLOTS = 2 * TC;        // 4 * TC MFMAs per tile pair in slots of two
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
      load_w1(NextPar{}, NST, sl);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 2 * sl; i < 2 * sl + 2; ++i) mfma1(Par{}, 0, i, 0, 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (S == 0 && pair > 0) issue_halo(2 * pair + 1, 1);
    if (S == 5 && !last) issue_halo(2 * pair + 2, 0);
    if (kt + 3 < nk) issue_w(kt + 3, ST);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
      if (sl < TC) {                 // the 4 fragments of tiles 0, 1 over the first TC slots
#pragma unroll
        for (int r = sl * 4 / TC; r < (sl + 1) * 4 / TC; ++r) load_px1(r >> 1, NextS{}, r & 1);
      } else {                       // operand 1 of tiles 2, 3 over the other TC slots (Y . operand 1 of this pair ran first)
#pragma unroll
        for (int r = (sl - TC) * 2 / TC; r < (sl - TC + 1) * 2 / TC; ++r) load_px1(2 + r, NextS{}, 1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 2 * sl; i < 2 * sl + 2; ++i) mfma1(Par{}, 2, i, 1, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    load_px1(2, NextS{}, 0);
    load_px1(3, NextS{}, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int pair = 0; pair < npair; ++pair) {
    const int kt = pair * 9;
    kstep(kt + 0, pair, std::integral_constant<int, 0>{});
    kstep(kt + 1, pair, std::integral_constant<int, 1>{});
    kstep(kt + 2, pair, std::integral_constant<int, 2>{});
    kstep(kt + 3, pair, std::integral_constant<int, 3>{});
    kstep(kt + 4, pair, std::integral_constant<int, 4>{});
    kstep(kt + 5, pair, std::integral_constant<int, 5>{});
    kstep(kt + 6, pair, std::integral_constant<int, 6>{});
    kstep(kt + 7, pair, std::integral_constant<int, 7>{});
    kstep(kt + 8, pair, std::integral_constant<int, 8>{});
  }

  const int q = lane >> 4;
  const int cb = n0 + wc * 16 * TC;
#pragma unroll
  for (int p = 0; p < TP; ++p) {
    const int y = ty0 + row0 + p / EN, x = tx0 + 16 * (p % EN) + l15;
    const int m = (b * a.H + y) * a.W + x;
    epilogue_tiles<TOut, TC>(a, m, cb, q, acc[p]);
  }
}

template <int BC, int WP, int WC_>
static int launch_halo_pair_t(const ConvArgs& a, hipStream_t s) {
  const int nbx = a.B * (a.H / 8) * (a.W / 32), nby = a.Cout_pad / BC;
  dim3 grid(8 * ((nbx + 7) / 8) * nby);
  hipLaunchKernelGGL((conv3x3_halo_pair_kernel<BC, WP, WC_>), grid, dim3(256), 0, s, a);
  CTDET_LAUNCH_CHECK();
  return 0;
}

template <int BC, int WP, int WC_, int TW = 32>
static int launch_halo_pair2_t(const ConvArgs& a, hipStream_t s) {
  const int nbx = a.B * (a.H / (256 / TW)) * (a.W / TW), nby = a.Cout_pad / BC;
  dim3 grid(8 * ((nbx + 7) / 8) * nby);
  hipLaunchKernelGGL((conv3x3_halo_pair2_kernel<BC, WP, WC_, TW>), grid, dim3(256), 0, s, a);
  CTDET_LAUNCH_CHECK();
  return 0;
}

template <int BC, int WP, int WC_, typename TOut, int TW = 32>
static int launch_halo_tap2(const ConvArgs& a, hipStream_t s) {
  const int nbx = a.B * (a.H / (256 / TW)) * (a.W / TW), nby = a.Cout_pad / BC;
  dim3 grid(8 * ((nbx + 7) / 8) * nby);
  hipLaunchKernelGGL((conv3x3_halo_tap2_kernel<BC, WP, WC_, TOut, TW>), grid, dim3(256), 0, s, a);
  CTDET_LAUNCH_CHECK();
  return 0;
}

// korder 3 (cross-chunk pair-packed split weights, Kpad = Cin / 32 * 288)
int launch_halo_pair2(const ConvArgs& a, hipStream_t s) {
  CTDET_CHECK(a.R == 3 && a.S == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 && a.in_dil == 1 && a.nsrc <= 1 &&
                  a.Cin % 32 == 0 && a.Kpad == a.Cin / 32 * 288 && ((a.H % 8 == 0 && a.W % 32 == 0) || (a.H % 16 == 0 && a.W % 16 == 0)) &&
                  a.Ho == a.H && a.Wo == a.W &&
                  a.in_stride % 4 == 0 && (((size_t)a.x | (size_t)a.w) & 15) == 0,
              "conv(f16x3, cross-chunk pair weights): needs 3x3/s1/p1, Cin %% 32 == 0 and a map divisible by 8x32 or 16x16 (Cin=%d, %dx%d, Kpad=%d)",
              a.Cin, a.H, a.W, a.Kpad);
  // a grid of 64-cout tiles that leaves CUs without a workgroup (the 512-channel level at batch 16: 16 pixel tiles x 8) runs on
  // 32-cout tiles instead: twice the workgroups, each with half the MFMAs behind the same window traffic
  const bool small_grid = (long)(a.M / 256) * (a.Cout_pad / 64) < ctdet_device_cu_count() && a.Cout_pad % 32 == 0 &&
                          !(ctdet_tuning_flags() & CTDET_TUNE_NO_SMALL_GRID_TILES);
  if (a.W % 32 != 0) {                  // 16 x 16-pixel tiles
    if (pick_bc(a.Cout) <= 32 || small_grid) return launch_halo_pair2_t<32, 4, 1, 16>(a, s);
    return launch_halo_pair2_t<64, 4, 1, 16>(a, s);
  }
  if (pick_bc(a.Cout) <= 32 || small_grid) return launch_halo_pair2_t<32, 4, 1>(a, s);
  if ((ctdet_tuning_flags() & CTDET_TUNE_PAIR2_128) && a.Cout_pad % 128 == 0) return launch_halo_pair2_t<128, 4, 1>(a, s);
  return launch_halo_pair2_t<64, 4, 1>(a, s);
}

// korder 2 (pair-packed split weights, Kpad = Cin / 16 * 160): only this kernel reads them
int launch_halo_pair(const ConvArgs& a, hipStream_t s) {
  CTDET_CHECK(a.R == 3 && a.S == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 && a.in_dil == 1 && a.nsrc <= 1 &&
                  a.Cin % 16 == 0 && a.Kpad == a.Cin / 16 * 160 && a.H % 8 == 0 && a.W % 32 == 0 && a.Ho == a.H && a.Wo == a.W &&
                  a.in_stride % 4 == 0 && (((size_t)a.x | (size_t)a.w) & 15) == 0,
              "conv(f16x3, pair weights): needs 3x3/s1/p1, Cin %% 16 == 0 and a map divisible by 8x32 (Cin=%d, %dx%d, Kpad=%d)",
              a.Cin, a.H, a.W, a.Kpad);
  // 64-cout tiles also for the 128-cout layers: with 128 accumulators the two operand register sets do not fit
  if (pick_bc(a.Cout) <= 32) return launch_halo_pair_t<32, 4, 1>(a, s);
  return launch_halo_pair_t<64, 4, 1>(a, s);
}

// f16x3 mode (called from conv_f32.hip): 3x3 / s1 / p1 on f32 activations, tap-major split weights, Cin % 16 == 0, map
// divisible by 8x32.  Returns 1 if the shape does not qualify (the caller then takes the uniform-K kernel).
int launch_halo_split(const ConvArgs& a, hipStream_t s) {
  const bool ok = a.R == 3 && a.S == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 && a.in_dil == 1 && a.nsrc <= 1 &&
                  a.korder == 0 && a.Cin % 16 == 0 && a.Kpad == a.K && a.H % 8 == 0 && a.W % 32 == 0 && a.Ho == a.H &&
                  a.Wo == a.W && a.in_stride % 4 == 0 && (((size_t)a.x) & 15) == 0 && !(ctdet_tuning_flags() & CTDET_TUNE_NO_HALO);
  if (!ok) return 1;
  switch (pick_bc(a.Cout)) {
    case 32: return launch_halo<32, 4, 1, float, true>(a, s);
    case 64: return launch_halo<64, 4, 1, float, true>(a, s);
    // 128-cout layers run on 64-cout tiles too: with 128 accumulators the split operands no longer fit in 256 registers (54-81
    // spilled, reloaded inside the K loop), and the spilling 256x128 instantiation measured 8-18 % slower than 256x64
    case 128: return launch_halo<64, 4, 1, float, true>(a, s);
  }
  return 1;
}

// ------------------------------------------------------------------------------------------
// Fused CenterNet head (centernet.py:115-121, 151-154): per head  out = W2 * relu(conv3x3(x, W1) + b1) + b2.
// The reference (and the first version here) writes the 256-channel hidden map of every head to memory and reads it
// back for the 1x1: 3 x 537 MB written + read per 64 images, the largest single item of HBM traffic in the step.
// Here a workgroup owns an 8x16 pixel tile of ONE head and all 256 hidden channels: wave w holds rows 2w, 2w+1 of the
// tile (2 pixel tiles) x 256 hidden (16 cout tiles) = 128 accumulator registers, the same count as the 256x128 halo
// kernel above, whose K loop this is (input window of a 32-channel chunk in LDS once, 9 taps read it at shifted
// addresses, weights through a 3-stage LDS-DMA ring).  Because a lane ends up with 8 consecutive hidden channels of a
// pixel per cout-tile pair (cout_of), relu(acc + b1) rounded to f16 IS the MFMA B fragment of the second GEMM: the
// 1x1 runs on the registers (W2 fragments straight from global/L2), no cross-wave reduction, and only the final
// maps are stored.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2) head_fused_kernel(const HeadArgs a) {
  constexpr int TH = 8, TW = 16, HID = 256, TC = HID / 16, TP = 2;
  constexpr int HMAIN = 3 * 4096, HSIDE = 4096, HBUF = HMAIN + HSIDE;   // main [12 rows][16 px][64 B] (10 used), side
  constexpr int WST = HID * 64, B_LD = HID / 64;
  static_assert(2 * HBUF + 3 * WST <= 81920, "two workgroups per CU");
  __shared__ __attribute__((aligned(16))) char smem[2 * HBUF + 3 * WST];
  char* const ring = smem + 2 * HBUF;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = a.W / TW, tiles_y = a.H / TH;
  int m_tile, head;
  if (!tile_of_block(a.B * tiles_y * tiles_x, a.nheads, m_tile, head)) return;
  const int tx0 = (m_tile % tiles_x) * TW;
  const int ty0 = ((m_tile / tiles_x) % tiles_y) * TH;
  const int b = m_tile / (tiles_x * tiles_y);
  const f16* zero = (const f16*)g_zero_page;
  asm volatile("" : "+v"(zero));
  const f16* ximg = (const f16*)a.x + (long)b * a.H * a.W * a.in_stride;

  // ---- halo loader: 3 main pieces (rows r, r+4, r+8) + 1 side piece per thread and chunk ----
  const int hslot = tid & 3, hpx = (tid >> 2) & 15, hr0 = tid >> 6;
  const int y0 = ty0 - 1 + hr0;
  const f16* hp0 = ximg + ((long)y0 * a.W + tx0 + hpx) * a.in_stride + (hslot ^ swz(hpx)) * 8;
  const long row4 = 4L * a.W * a.in_stride;
  unsigned hmask = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) hmask |= (hr0 + 4 * i < TH + 2 && y0 + 4 * i >= 0 && y0 + 4 * i < a.H) ? (1u << i) : 0u;
  const f16* hps;
  {
    const int side = (tid >> 2) & 1, hr = tid >> 3;   // [hr 0..9][side][slot], tid < 80
    const int y = ty0 - 1 + hr, x = side ? tx0 + TW : tx0 - 1;
    const bool ok = tid < 80 && y >= 0 && y < a.H && x >= 0 && x < a.W;
    hps = ximg + ((long)(ok ? y : 0) * a.W + (ok ? x : 0)) * a.in_stride + hslot * 8;
    hmask |= ok ? 8u : 0u;
  }
  const int Kpad = 9 * a.Cin;
  const int lrow = tid >> 2;
  const int gw = hslot ^ swz(lrow);
  const f16* wptr[B_LD];
#pragma unroll
  for (int j = 0; j < B_LD; ++j) {
    const int L = lrow + 64 * j;
    const int tt = L >> 4, r = L & 15;
    wptr[j] = (const f16*)a.w1 + (long)(head * HID + cout_of<TC>(tt, r >> 2, r & 3)) * Kpad + gw * 8;
  }
  auto issue_halo = [&](int chunk, int hb) {
    char* dst = smem + hb * HBUF + wave * 1024;
    const long coff = (long)chunk * 32;
#pragma unroll
    for (int i = 0; i < 3; ++i) dma16((hmask & (1u << i)) ? hp0 + i * row4 + coff : zero, dst + i * 4096);
    dma16((hmask & 8u) ? hps + coff : zero, smem + hb * HBUF + HMAIN + wave * 1024);
  };
  auto issue_w = [&](int kt, int st) {
#pragma unroll
    for (int j = 0; j < B_LD; ++j) dma16(wptr[j] + (long)kt * 32, ring + st * WST + wave * 1024 + j * 4096);
  };

  // ---- fragment addressing (one pixel tile per tile row): lane column l15, tap column s -> window column l15+s-1 ----
  const int l15 = lane & 15, kg = lane >> 4;
  const int row0 = wave * TP;
  int abase[3];
#pragma unroll
  for (int s2 = 0; s2 < 3; ++s2) {
    const int X = l15 + s2 - 1;
    if (X < 0) abase[s2] = HMAIN + kg * 16 + row0 * 128;
    else if (X > 15) abase[s2] = HMAIN + 64 + kg * 16 + row0 * 128;
    else abase[s2] = X * 64 + ((kg ^ swz(X)) << 4) + row0 * 1024;
  }
  const int estride0 = (l15 == 0) ? 128 : 1024;    // row stride of this lane for tap column 0
  const int estride2 = (l15 == 15) ? 128 : 1024;   // ... for tap column 2
  const char* fragB = ring + l15 * 64 + ((kg ^ swz(l15)) << 4);

  f32x4 acc[TP][TC];
#pragma unroll
  for (int p = 0; p < TP; ++p)
#pragma unroll
    for (int c = 0; c < TC; ++c) acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nch = a.Cin / 32, nk = nch * 9;
  issue_halo(0, 0);
  issue_w(0, 0);
  issue_w(1, 1);

  auto kstep = [&](int kt, int chunk, auto tapc, auto hbc) {
    constexpr int T = decltype(tapc)::value, HB = decltype(hbc)::value;
    constexpr int R_ = T / 3, S_ = T % 3, ST = T % 3, SL = (T + 2) % 3;
    if (kt + 1 < nk) { if (T == 1 && chunk + 1 < nch) wait_vmcnt<B_LD + 4>(); else wait_vmcnt<B_LD>(); }
    else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (T == 0 && chunk + 1 < nch) issue_halo(chunk + 1, HB ^ 1);
    if (kt + 2 < nk) issue_w(kt + 2, SL);
    const char* hbuf = smem + HB * HBUF;
    f16x8 pf[TP];
#pragma unroll
    for (int p = 0; p < TP; ++p) {
      int off;
      if (S_ == 0) off = abase[0] + (p + R_) * estride0;
      else if (S_ == 2) off = abase[2] + (p + R_) * estride2;
      else off = abase[1] + (p + R_) * 1024;
      pf[p] = *(const f16x8*)(hbuf + off);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f16x8 wf[TC / 2];
#pragma unroll
      for (int c = 0; c < TC / 2; ++c) wf[c] = *(const f16x8*)(fragB + ST * WST + (half * (TC / 2) + c) * 1024);
#pragma unroll
      for (int c = 0; c < TC / 2; ++c)
#pragma unroll
        for (int p = 0; p < TP; ++p)
          acc[p][half * (TC / 2) + c] =
              __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c], pf[p], acc[p][half * (TC / 2) + c], 0, 0, 0);
    }
  };
  auto chunk_steps = [&](int kt, int chunk, auto hbc) {
    kstep(kt + 0, chunk, std::integral_constant<int, 0>{}, hbc);
    kstep(kt + 1, chunk, std::integral_constant<int, 1>{}, hbc);
    kstep(kt + 2, chunk, std::integral_constant<int, 2>{}, hbc);
    kstep(kt + 3, chunk, std::integral_constant<int, 3>{}, hbc);
    kstep(kt + 4, chunk, std::integral_constant<int, 4>{}, hbc);
    kstep(kt + 5, chunk, std::integral_constant<int, 5>{}, hbc);
    kstep(kt + 6, chunk, std::integral_constant<int, 6>{}, hbc);
    kstep(kt + 7, chunk, std::integral_constant<int, 7>{}, hbc);
    kstep(kt + 8, chunk, std::integral_constant<int, 8>{}, hbc);
  };
  int chunk = 0;
  for (; chunk + 1 < nch; chunk += 2) {
    chunk_steps(chunk * 9, chunk, std::integral_constant<int, 0>{});
    chunk_steps(chunk * 9 + 9, chunk + 1, std::integral_constant<int, 1>{});
  }
  if (chunk < nch) chunk_steps(chunk * 9, chunk, std::integral_constant<int, 0>{});

  // ---- hidden = relu(acc + b1) as f16 B fragments: frag[p][hb] = hidden channels hb*32 + kg*8 .. +8 of pixel l15 ----
  const float* b1 = a.b1 + head * HID;
  f16x8 frag[TP][TC / 2];
#pragma unroll
  for (int hb = 0; hb < TC / 2; ++hb) {
    const f32x4 ba = *(const f32x4*)(b1 + hb * 32 + kg * 8), bb = *(const f32x4*)(b1 + hb * 32 + kg * 8 + 4);
#pragma unroll
    for (int p = 0; p < TP; ++p) {
      f16x8 f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f[j] = (f16)fmaxf(acc[p][2 * hb][j] + ba[j], 0.f);
        f[4 + j] = (f16)fmaxf(acc[p][2 * hb + 1][j] + bb[j], 0.f);
      }
      frag[p][hb] = f;
    }
  }
  // ---- 1x1: out tile ot (16 outputs) = sum over the 8 hidden blocks of W2[ot][hb] x frag[.][hb] ----
  const f16* w2 = (const f16*)a.w2[head];
  const float* b2 = a.b2[head];
  float* yh = a.y[head];
  const int ystride = a.y_stride[head], cout = a.cout[head], act = a.act[head];
  const int ntile2 = (cout + 15) >> 4;
  for (int ot = 0; ot < ntile2; ++ot) {
    f32x4 o[TP];
#pragma unroll
    for (int p = 0; p < TP; ++p) o[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f16* wrow = w2 + (long)(ot * 16 + l15) * HID + kg * 8;
#pragma unroll
    for (int hb = 0; hb < TC / 2; ++hb) {
      const f16x8 wa = *(const f16x8*)(wrow + hb * 32);
#pragma unroll
      for (int p = 0; p < TP; ++p) o[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, frag[p][hb], o[p], 0, 0, 0);
    }
    const int c0 = ot * 16 + 4 * kg;               // this lane's 4 outputs
    if (c0 < ((cout + 3) & ~3)) {
      const f32x4 bv = *(const f32x4*)(b2 + c0);
#pragma unroll
      for (int p = 0; p < TP; ++p) {
        f32x4 v = o[p] + bv;
        if (act == CTDET_ACT_RELU) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        } else if (act == CTDET_ACT_SIGMOID_CLAMP) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fminf(fmaxf(ctdet_sigmoid_exact(v[j]), a.clamp_lo), a.clamp_hi);
        }
        const long m = (long)(b * a.H + ty0 + row0 + p) * a.W + tx0 + l15;
        *(f32x4*)(yh + m * ystride + c0) = v;
      }
    }
  }
}

int launch_head_fused(const HeadArgs& a, hipStream_t s) {
  CTDET_CHECK(a.nheads >= 1 && a.nheads <= 4, "head_fused: 1..4 heads");
  CTDET_CHECK(a.Cin % 32 == 0 && a.H % 8 == 0 && a.W % 16 == 0 && a.in_stride % 8 == 0,
              "head_fused: needs Cin %% 32 == 0 and a map divisible by 8x16 (Cin=%d, %dx%d)", a.Cin, a.H, a.W);
  for (int h = 0; h < a.nheads; ++h)
    CTDET_CHECK(a.cout[h] >= 1 && a.cout[h] <= 256 && a.y_stride[h] % 4 == 0 && a.y_stride[h] >= ((a.cout[h] + 3) & ~3) &&
                    (((size_t)a.y[h]) & 15) == 0,
                "head_fused: head %d: bad output (cout %d, stride %d)", h, a.cout[h], a.y_stride[h]);
  const int nbx = a.B * (a.H / 8) * (a.W / 16);
  dim3 grid(8 * ((nbx + 7) / 8) * a.nheads);
  hipLaunchKernelGGL(head_fused_kernel, grid, dim3(256), 0, s, a);
  CTDET_LAUNCH_CHECK();
  return 0;
}

static bool halo16_ok(const ConvArgs& a) {
  return a.R == 3 && a.S == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 && a.in_dil == 1 && a.nsrc <= 1 &&
         a.korder == 1 && a.Cin % 32 == 0 && a.Kpad == a.K && a.H % 16 == 0 && a.W % 16 == 0 && a.W % 32 != 0 && a.Ho == a.H &&
         a.Wo == a.W && !(ctdet_tuning_flags() & CTDET_TUNE_NO_HALO);
}

static bool halo_ok(const ConvArgs& a) {
  return a.R == 3 && a.S == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 && a.in_dil == 1 && a.nsrc <= 1 &&
         a.korder == 1 && a.Cin % 32 == 0 && a.Kpad == a.K && a.H % 8 == 0 && a.W % 32 == 0 && a.Ho == a.H &&
         a.Wo == a.W && !(ctdet_tuning_flags() & CTDET_TUNE_NO_HALO);
}

// ------------------------------------------------------------------------------------------
// DCNv2 (3x3 / s1 / p1) with the bilinear gathers served from LDS.  The global-gather kernel above is bound by
// L1 bandwidth (36 KB of 16-byte corner loads per 32-deep K step against 64 B/clk/CU); here a workgroup owns an
// 8x16-pixel output tile, and for every 32-channel chunk the input window of the tile (+-1 for the taps, +-4 margin
// for the learned offsets: 18x26 pixels, 30 KB) is brought into LDS once by DMA.  All 9 taps x 4 corners then read
// it with ds_read_b128 (256 B/clk/CU).  Sampling geometry (corner offset, validity bits, f16 weight*mask) is
// computed once per (pixel, tap) and kept in 27 VGPRs across the chunks.  If any in-image corner of the tile falls
// outside the window, the whole workgroup takes the gather-from-global path instead (block-uniform, exact).
// ------------------------------------------------------------------------------------------
// packed-f16 blend steps with the bilinear weight taken from one half of a register holding two of them (VOP3P
// op_sel broadcast), so that the 36 weights of a pixel stay in 18 VGPRs instead of being splat into 144
__device__ __forceinline__ unsigned pk_mul_wlo(unsigned v, unsigned w) {
  unsigned r;
  asm("v_pk_mul_f16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(v), "v"(w));
  return r;
}
__device__ __forceinline__ unsigned pk_fma_wlo(unsigned v, unsigned w, unsigned c) {
  unsigned r;
  asm("v_pk_fma_f16 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(v), "v"(w), "v"(c));
  return r;
}
__device__ __forceinline__ unsigned pk_fma_whi(unsigned v, unsigned w, unsigned c) {
  unsigned r;
  asm("v_pk_fma_f16 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(v), "v"(w), "v"(c));
  return r;
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// r = v0*w0 + v1*w1 + v2*w2 + v3*w3 on 8 packed f16 lanes (w01 = w0|w1<<16, w23 = w2|w3<<16), rounding after every
// step like the separate v_pk ops.  One asm block, the four dword chains interleaved, so that consecutive
// instructions are independent and no hazard nops are needed between them.
__device__ __forceinline__ u32x4 dcn_blend(const u32x4& v0, const u32x4& v1, const u32x4& v2, const u32x4& v3,
                                           unsigned w01, unsigned w23) {
  unsigned r0, r1, r2, r3;
  asm("v_pk_mul_f16 %0, %4, %20 op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f16 %1, %5, %20 op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f16 %2, %6, %20 op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f16 %3, %7, %20 op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f16 %0, %8, %20, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f16 %1, %9, %20, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f16 %2, %10, %20, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f16 %3, %11, %20, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f16 %0, %12, %21, %0 op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f16 %1, %13, %21, %1 op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f16 %2, %14, %21, %2 op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f16 %3, %15, %21, %3 op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f16 %0, %16, %21, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f16 %1, %17, %21, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f16 %2, %18, %21, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f16 %3, %19, %21, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
      : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
      : "v"(v0.x), "v"(v0.y), "v"(v0.z), "v"(v0.w), "v"(v1.x), "v"(v1.y), "v"(v1.z), "v"(v1.w), "v"(v2.x), "v"(v2.y),
        "v"(v2.z), "v"(v2.w), "v"(v3.x), "v"(v3.y), "v"(v3.z), "v"(v3.w), "v"(w01), "v"(w23));
  u32x4 r;
  r.x = r0; r.y = r1; r.z = r2; r.w = r3;
  return r;
}

// PARTIAL: the map is not a multiple of the 8x16 tile (edge tiles carry pixels outside it)
// MIXED selects how a tap is served when some lane of the wave samples outside the window.  false (default): the whole wave
// gathers that tap from global memory, re-deriving the coordinates from the offsets.  true (CTDET_DCN_MIXED=1): the geometry
// stage stores the image coordinates of such samples and only those lanes go to global memory, the rest keep reading the
// window -- 10 % / 35 % faster at offset sigma 2 / 4 px, 5 % slower when (almost) nothing leaves the window, which is the
// regime of the benchmark's weights; a separate instantiation so that the default kernel's code is untouched.
template <int BC, int NST, bool PARTIAL, typename TOut, bool MIXED = false>
__global__ void __launch_bounds__(256, 2) dcn_window_kernel(const ConvArgs a) {
  constexpr int TH = 8, TW = 16, BP = 128, MG = 4;
  constexpr int WR = TH + 2 + 2 * MG, WCOLS = TW + 2 + 2 * MG;  // 18 x 26 window pixels
  constexpr int NPIECE = WR * WCOLS * 4;                        // 1872 16-byte pieces
  constexpr int W_LD = (NPIECE + 255) / 256;                    // 8 DMA rounds; the last one only on waves 0-1
  constexpr int WINB = ((NPIECE + 63) / 64) * 1024;             // 30720
  constexpr int GEOW = 9 * BP * 8, GEOO = 9 * BP * 4;           // staged geometry: 4 f16 weights, window offset
  constexpr int TP = 2, TC = BC / 16;                           // wave = 32 pixels (2 tile rows) x all BC couts
  constexpr int B_LD = BC / 64, WST = BC * 64;
  constexpr int PF = NST - 1;                                   // weight stages in flight ahead of the consumer
  static_assert(BC % 64 == 0, "BC");
  constexpr int SBB = 2 * BC * 4;                               // per-cout scale, bias (f32) for the epilogue
  constexpr bool WPRE = BC == 64;                               // weight fragments fetched one step ahead
  static_assert(WINB + GEOW + GEOO + NST * WST + SBB <= 81920, "two workgroups per CU");
  __shared__ __attribute__((aligned(16))) char smem[WINB + GEOW + GEOO + NST * WST + SBB];
  char* const win = smem;
  char* const geow = smem + WINB;
  char* const geoo = smem + WINB + GEOW;
  char* const ring = smem + WINB + GEOW + GEOO;
  float* const sbuf = (float*)(smem + WINB + GEOW + GEOO + NST * WST);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;   // edge tiles may be partial
  int m_tile, n_tile;
  if (!tile_of_block(a.B * tiles_y * tiles_x, a.Cout_pad / BC, m_tile, n_tile)) return;
  const int tx0 = (m_tile % tiles_x) * TW;
  const int ty0 = ((m_tile / tiles_x) % tiles_y) * TH;
  const int b = m_tile / (tiles_x * tiles_y);
  const int n0 = n_tile * BC;
  const f16* zero = (const f16*)g_zero_page;
  asm volatile("" : "+v"(zero));
  const f16* ximg = (const f16*)a.x + (long)b * a.H * a.W * a.in_stride;
  const int wy0 = ty0 - 1 - MG, wx0 = tx0 - 1 - MG;  // image coordinates of window pixel (0,0)
  const int nch = a.Cin / 32, nk = nch * 9;

  // ---- loaders ----
  const int lrow = tid >> 2, slotw = tid & 3;
  const int gwk = slotw ^ swz(lrow);
  const f16* wptr[B_LD];
#pragma unroll
  for (int j = 0; j < B_LD; ++j) {
    const int L = lrow + 64 * j;                 // LDS row; holds cout cl so that lane (fr, q) finds couts 4*TC*q.. contiguous
    const int tt = L >> 4, r = L & 15;
    const int cl = cout_of<TC>(tt, r >> 2, r & 3);
    wptr[j] = (const f16*)a.w + (long)(n0 + cl) * a.Kpad + gwk * 8;
  }
  auto issue_w = [&](int kt) {
    const int st = kt % NST;
#pragma unroll
    for (int j = 0; j < B_LD; ++j) dma16(wptr[j] + (long)kt * 32, ring + st * WST + wave * 1024 + j * 4096);
  };
  // element offset (chunk 0) of the 16-byte window piece each DMA round of this lane fetches; -1 = zero fill
  int wofs[W_LD];
#pragma unroll
  for (int i = 0; i < W_LD; ++i) {
    const int pid = tid + 256 * i;
    const int pw = pid >> 2, sl = pid & 3;
    const int wr = pw / WCOLS, wcn = pw - wr * WCOLS;
    const int y = wy0 + wr, x = wx0 + wcn;
    const bool ok = pid < NPIECE && y >= 0 && y < a.H && x >= 0 && x < a.W;
    // LDS slot sl of a pixel in an odd window row holds channel group sl^2: the 16 lanes of one ds_read_b128 phase
    // (8 pixels of one group q, 8 of another, over two rows) then touch 16 different 16-byte bank groups
    wofs[i] = ok ? (y * a.W + x) * a.in_stride + (sl ^ (2 * (wr & 1))) * 8 : -1;
  }
  auto issue_window = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < W_LD; ++i) {
      if (i < W_LD - 1 || (wave * 64 + 256 * i) * 16 < WINB)  // wave-uniform: the last round ends with the window
        dma16(wofs[i] >= 0 ? ximg + wofs[i] + chunk * 32 : zero, win + (wave * 64 + 256 * i) * 16);
    }
  };
  // the window of chunk 0 and the first PF weight stages go out before anything else: their latency covers the
  // offset/mask loads and the geometry set-up below
  issue_window(0);
#pragma unroll
  for (int i = 0; i < PF; ++i)
    if (i < nk) issue_w(i);

  if (tid < BC) {
    const int c = n0 + tid;
    sbuf[tid] = (a.scale && c < a.Cout) ? a.scale[c] : 1.f;
    sbuf[BC + tid] = (a.bias && c < a.Cout) ? a.bias[c] : 0.f;
  }
  // ---- sampling geometry, once per (pixel, tap): thread = pixel gp, taps gh, gh+2, ... ----
  {
    const int gp = tid & 127, gh = tid >> 7;
    const int py = ty0 + (gp >> 4), pxx = tx0 + (gp & 15);
    const bool pin = !PARTIAL || (py < a.H && pxx < a.W);   // pixel of a partial edge tile outside the map: contributes nothing
    const float* omrow = a.om + ((long)(b * a.H + (pin ? py : 0)) * a.W + (pin ? pxx : 0)) * a.om_stride;
    float oh[5], ow[5], om_[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int t = 2 * i + gh;
      const bool on = t < 9 && pin;
      oh[i] = on ? omrow[2 * t] : 0.f; ow[i] = on ? omrow[2 * t + 1] : 0.f; om_[i] = on ? omrow[18 + t] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int t = 2 * i + gh;
      if (t < 9) {
        const int tr = t / 3, ts = t - 3 * tr;
        const float h_im = (float)(py - 1 + tr) + oh[i], w_im = (float)(pxx - 1 + ts) + ow[i];
        const float mk = a.mask_is_prob ? om_[i] : __builtin_amdgcn_rcpf(1.f + __expf(-om_[i]));
        const float fh = floorf(h_im), fw = floorf(w_im);
        const int h_low = (int)fh, w_low = (int)fw;
        const float lh = h_im - fh, lw = w_im - fw, hh = 1.f - lh, hw = 1.f - lw;
        const bool valid = pin && h_im > -1.f && w_im > -1.f && h_im < (float)a.H && w_im < (float)a.W;
        const bool r0 = valid && h_low >= 0, r1 = valid && h_low + 1 <= a.H - 1, c0 = w_low >= 0, c1 = w_low + 1 <= a.W - 1;
        // corners outside the image contribute 0 (deform_conv_cuda_kernel.cu:259-266): zero their weights
        const float w0 = (r0 && c0) ? hh * hw * mk : 0.f, w1 = (r0 && c1) ? hh * lw * mk : 0.f;
        const float w2 = (r1 && c0) ? lh * hw * mk : 0.f, w3 = (r1 && c1) ? lh * lw * mk : 0.f;
        const int wr = h_low - wy0, wcn = w_low - wx0;  // window coordinates of corner 0
        // in a valid sample every in-image corner must lie inside the window; out-of-image corners then do too
        // (they are 1 px outside the image, the window reaches 5 px past the tile and is zero-filled there)
        const bool inside = wr >= 0 && wr + 1 < WR && wcn >= 0 && wcn + 1 < WCOLS;
        const bool oow = valid && !inside;
        const unsigned off = (valid && inside) ? (unsigned)((wr * WCOLS + wcn) * 64) : 0u;
        const f16 h0 = (f16)w0, h1 = (f16)w1, h2 = (f16)w2, h3 = (f16)w3;
        uint2 wv;
        wv.x = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
        wv.y = (unsigned)__builtin_bit_cast(unsigned short, h2) | ((unsigned)__builtin_bit_cast(unsigned short, h3) << 16);
        *(uint2*)(geow + (t * BP + gp) * 8) = wv;
        if constexpr (MIXED) {
          // out-of-window sample (bit 31): the image coordinates of corner 0, biased by one (h_low, w_low >= -1 in a valid
          // sample), instead of a window offset
          *(unsigned*)(geoo + (t * BP + gp) * 4) =
              oow ? (0x80000000u | ((unsigned)(h_low + 1) << 12) | (unsigned)(w_low + 1)) : (off | ((unsigned)(wr & 1) << 16));
        } else {
          *(unsigned*)(geoo + (t * BP + gp) * 4) = off | ((unsigned)(wr & 1) << 16) | (oow ? 0x80000000u : 0u);
        }
      }
    }
  }
  wait_vmcnt<0>();     // window of chunk 0 (and the first weight stages) landed for this wave's DMAs
  __syncthreads();     // ... and for everyone's; geometry staged
  // ---- consumer mapping: lane (fr, q) blends tile pixel (row 2*wave + fr/8, col 8p + fr%8), channels 8q..8q+7 of
  // the chunk: exactly the B-operand fragment of mfma_f32_16x16x32_f16 for the wave's pixel tile p (2 rows x 8) ----
  const int fr = lane & 15, q = lane >> 4;
  const int prow = 2 * wave + (fr >> 3), pcol = fr & 7;
  unsigned gofs[TP][9], gw01[TP][9], gw23[TP][9];
  unsigned oow_bits = 0;
#pragma unroll
  for (int p = 0; p < TP; ++p)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int px = prow * 16 + 8 * p + pcol;
      const uint2 wv = *(const uint2*)(geow + (t * BP + px) * 8);
      const unsigned o = *(const unsigned*)(geoo + (t * BP + px) * 4);
      gw01[p][t] = wv.x; gw23[p][t] = wv.y;
      gofs[p][t] = (MIXED && (o >> 31)) ? (unsigned)(q << 4) : (o & 0xFFFFu) + ((q ^ (2 * ((o >> 16) & 1u))) << 4);
      oow_bits |= (o >> 31) << t;
    }
  // taps for which some lane of this wave samples outside the window take the global-gather path (wave-uniform)
  unsigned slow_taps = 0;
#pragma unroll
  for (int t = 0; t < 9; ++t)
    if (__builtin_amdgcn_ballot_w64((oow_bits >> t) & 1u) != 0) slow_taps |= 1u << t;
  slow_taps = __builtin_amdgcn_readfirstlane(slow_taps);

  typedef const u32x4 __attribute__((address_space(1)))* gp16;
  // fast gather: the 4 corner fragments of tap t for both pixel tiles, from the LDS window (blended later, behind
  // the MFMAs of the current step)
  auto gather_lds = [&](int t, u32x4 (&v)[TP][4]) {
#pragma unroll
    for (int p = 0; p < TP; ++p) {
      const char* c0p = win + gofs[p][t];
      const char* c2p = win + (gofs[p][t] ^ 32u) + WCOLS * 64;  // next window row: the other slot swizzle
      v[p][0] = *(const u32x4*)(c0p);
      v[p][1] = *(const u32x4*)(c0p + 64);
      v[p][2] = *(const u32x4*)(c2p);
      v[p][3] = *(const u32x4*)(c2p + 64);
    }
  };
  // slow gather (some lane of the wave samples outside the window for this tap): corners from global memory,
  // blended right away so that no global load is pending when control flow joins the fast path again (the
  // compiler would otherwise put vmcnt(0) waits -- which also wait for the weight prefetch -- into the fast path).
  // Corner coordinates are recomputed from the offsets with the arithmetic of the staging pass; corners outside
  // the image read the zero page, their weights are 0 as well.
  auto gather_global = [&](int t, int chunk, f16x8 (&pf)[TP]) {
    if constexpr (MIXED) {
#pragma unroll
      for (int p = 0; p < TP; ++p) {
        const int px = prow * 16 + 8 * p + pcol;
        const unsigned o = *(const unsigned*)(geoo + (t * BP + px) * 4);
        const bool out = o >> 31;
        const int h_low = (int)((o >> 12) & 0xFFFu) - 1, w_low = (int)(o & 0xFFFu) - 1;
        const bool r0 = out && h_low >= 0, r1 = out && h_low + 1 <= a.H - 1, c0 = w_low >= 0, c1 = w_low + 1 <= a.W - 1;
        const int o0 = (h_low * a.W + w_low) * a.in_stride, o2 = o0 + a.W * a.in_stride;
        const f16* base = ximg + chunk * 32 + q * 8;
        const u32x4 g0 = *(gp16)((r0 && c0) ? base + o0 : zero);      // lanes inside the window: the zero page
        const u32x4 g1 = *(gp16)((r0 && c1) ? base + o0 + a.in_stride : zero);
        const u32x4 g2 = *(gp16)((r1 && c0) ? base + o2 : zero);
        const u32x4 g3 = *(gp16)((r1 && c1) ? base + o2 + a.in_stride : zero);
        const char* c0p = win + gofs[p][t];
        const char* c2p = win + (gofs[p][t] ^ 32u) + WCOLS * 64;
        const u32x4 l0 = *(const u32x4*)(c0p), l1 = *(const u32x4*)(c0p + 64);
        const u32x4 l2 = *(const u32x4*)(c2p), l3 = *(const u32x4*)(c2p + 64);
        pf[p] = __builtin_bit_cast(f16x8, dcn_blend(out ? g0 : l0, out ? g1 : l1, out ? g2 : l2, out ? g3 : l3, gw01[p][t],
                                                    gw23[p][t]));
      }
      return;
    }
    const int tr = t / 3, ts = t % 3;
#pragma unroll
    for (int p = 0; p < TP; ++p) {
      const int py = ty0 + prow, pxx = tx0 + 8 * p + pcol;
      const bool pin = !PARTIAL || (py < a.H && pxx < a.W);
      const float* omrow = a.om + ((long)(b * a.H + (pin ? py : 0)) * a.W + (pin ? pxx : 0)) * a.om_stride;
      const float h_im = (float)(py - 1 + tr) + omrow[2 * t], w_im = (float)(pxx - 1 + ts) + omrow[2 * t + 1];
      const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
      const bool valid = pin && h_im > -1.f && w_im > -1.f && h_im < (float)a.H && w_im < (float)a.W;
      const bool r0 = valid && h_low >= 0, r1 = valid && h_low + 1 <= a.H - 1, c0 = w_low >= 0, c1 = w_low + 1 <= a.W - 1;
      const long o0 = ((long)h_low * a.W + w_low) * a.in_stride, o2 = o0 + (long)a.W * a.in_stride;
      const f16* base = ximg + chunk * 32 + q * 8;
      const u32x4 v0 = *(gp16)((r0 && c0) ? base + o0 : zero);
      const u32x4 v1 = *(gp16)((r0 && c1) ? base + o0 + a.in_stride : zero);
      const u32x4 v2 = *(gp16)((r1 && c0) ? base + o2 : zero);
      const u32x4 v3 = *(gp16)((r1 && c1) ? base + o2 + a.in_stride : zero);
      pf[p] = __builtin_bit_cast(f16x8, dcn_blend(v0, v1, v2, v3, gw01[p][t], gw23[p][t]));
    }
  };
  auto blend_lds = [&](int t, const u32x4 (&v)[TP][4], f16x8 (&pf)[TP]) {
#pragma unroll
    for (int p = 0; p < TP; ++p)
      pf[p] = __builtin_bit_cast(f16x8, dcn_blend(v[p][0], v[p][1], v[p][2], v[p][3], gw01[p][t], gw23[p][t]));
  };

  f32x4 acc[TP][TC];
#pragma unroll
  for (int p = 0; p < TP; ++p)
#pragma unroll
    for (int c = 0; c < TC; ++c) acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const char* fragB = ring + fr * 64 + ((q ^ swz(fr)) << 4);

  f16x8 pf[TP];  // B operands (sampled * mask) of the step about to run
  {
    u32x4 raw[TP][4];
    if (!(slow_taps & 1u)) { gather_lds(0, raw); blend_lds(0, raw, pf); }
    else gather_global(0, 0, pf);
  }

  // K step kt = (chunk, tap T): MFMAs of tap T against weight stage kt, with the gather + blend of tap T+1 around
  // them.  WPRE: the weight fragments of step kt+1 are fetched during step kt, so the barrier of step kt also
  // vouches for stage kt+1 (one stage less in flight).
  f16x8 wf[TC];
  if constexpr (WPRE) {
#pragma unroll
    for (int c = 0; c < TC; ++c) wf[c] = *(const f16x8*)(fragB + c * 1024);  // stage 0 (landed: prologue wait)
  }
  auto kstep = [&](int kt, int chunk, auto tapc) {
    constexpr int T = decltype(tapc)::value;
    constexpr int TN = (T + 1) % 9;                    // tap whose operands are produced during this step
    constexpr int KEEP = WPRE ? PF - 2 : PF - 1;       // younger weight stages that may still be in flight
    const bool more = T < 8;                           // T == 8: the next tap belongs to the next chunk (see below)
    const bool slow_next = (slow_taps >> TN) & 1u;     // wave-uniform
    if (kt + PF - 1 < nk) wait_vmcnt<KEEP * B_LD>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                      // stage kt (WPRE: kt+1) landed for all; everyone left step kt-1
    if (kt + PF < nk) issue_w(kt + PF);                // into the stage step kt-1 consumed
    if (T == 8 && chunk + 1 < nch) issue_window(chunk + 1);  // every wave's gathers of this chunk are complete
    f16x8 wfn[TC], pfn[TP];
    u32x4 raw[TP][4];
    if constexpr (WPRE) {
      if (kt + 1 < nk) {
        const int st = (kt + 1) % NST;
#pragma unroll
        for (int c = 0; c < TC; ++c) wfn[c] = *(const f16x8*)(fragB + st * WST + c * 1024);
      }
    } else {
      const int st = kt % NST;
#pragma unroll
      for (int c = 0; c < TC; ++c) wf[c] = *(const f16x8*)(fragB + st * WST + c * 1024);
    }
    if (more) {
      if (!slow_next) gather_lds(TN, raw); else gather_global(TN, chunk, pfn);
    }
#pragma unroll
    for (int p = 0; p < TP; ++p)
#pragma unroll
      for (int c = 0; c < TC; ++c) acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c], pf[p], acc[p][c], 0, 0, 0);
    if (more) {
      if (!slow_next) blend_lds(TN, raw, pfn);
    } else if (chunk + 1 < nch) {
      wait_vmcnt<0>();                                 // next chunk's window, behind this step's MFMAs
      __builtin_amdgcn_s_barrier();
      if (!slow_next) { gather_lds(0, raw); blend_lds(0, raw, pfn); }
      else gather_global(0, chunk + 1, pfn);
    }
#pragma unroll
    for (int p = 0; p < TP; ++p) pf[p] = pfn[p];
    if constexpr (WPRE) {
#pragma unroll
      for (int c = 0; c < TC; ++c) wf[c] = wfn[c];
    }
  };
  for (int chunk = 0; chunk < nch; ++chunk) {
    const int kt = chunk * 9;
    kstep(kt + 0, chunk, std::integral_constant<int, 0>{});
    kstep(kt + 1, chunk, std::integral_constant<int, 1>{});
    kstep(kt + 2, chunk, std::integral_constant<int, 2>{});
    kstep(kt + 3, chunk, std::integral_constant<int, 3>{});
    kstep(kt + 4, chunk, std::integral_constant<int, 4>{});
    kstep(kt + 5, chunk, std::integral_constant<int, 5>{});
    kstep(kt + 6, chunk, std::integral_constant<int, 6>{});
    kstep(kt + 7, chunk, std::integral_constant<int, 7>{});
    kstep(kt + 8, chunk, std::integral_constant<int, 8>{});
  }

  // epilogue: per cout-tile pair h a lane holds 8 consecutive couts (cout_of) of 2 pixels; scale/bias from LDS,
  // 16-byte stores
#pragma unroll
  for (int p = 0; p < TP; ++p) {
    if (PARTIAL && (ty0 + prow >= a.H || tx0 + 8 * p + pcol >= a.W)) continue;   // partial edge tile
    const long m = (long)(b * a.H + ty0 + prow) * a.W + tx0 + 8 * p + pcol;
#pragma unroll
    for (int h = 0; h < TC / 2; ++h) {
      const int cl = h * 32 + q * 8;  // first of the lane's 8 couts inside the BC tile
      TOut* yp = (TOut*)a.y + m * a.out_stride + n0 + cl;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = acc[p][2 * h + (j >> 2)][j & 3] * sbuf[cl + j] + sbuf[BC + cl + j];
        if (a.act == CTDET_ACT_RELU) t = fmaxf(t, 0.f);
        else if (a.act == CTDET_ACT_SIGMOID_CLAMP) t = fminf(fmaxf(ctdet_sigmoid_exact(t), a.clamp_lo), a.clamp_hi);
        v[j] = t;
      }
      if (n0 + cl + 8 <= a.Cout) {
        if constexpr (sizeof(TOut) == 2) {
          f16x8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (f16)v[j];
          *(f16x8*)yp = o;
        } else {
          *(f32x4*)yp = (f32x4){v[0], v[1], v[2], v[3]};
          *(f32x4*)(yp + 4) = (f32x4){v[4], v[5], v[6], v[7]};
        }
      } else if (n0 + cl + 4 <= a.Cout) {   // Cout % 4 == 0: a group of 4 is all-in or all-out
        if constexpr (sizeof(TOut) == 2) {
          f16x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (f16)v[j];
          *(f16x4*)yp = o;
        } else {
          *(f32x4*)yp = (f32x4){v[0], v[1], v[2], v[3]};
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// The same tile / window / operand mapping with the per-tile instruction count cut (the kernel above is issue-bound:
// ~3.6 k instructions per wave for the 64->64 layers, of which 576 blend + 144 MFMA + 216 LDS reads are the work).
// For 64-cout layers on tile-divisible maps:
//   * window rows padded to 32 pixels: a DMA round is two window rows (thread = row parity, column, slot), so the piece
//     offsets and the sample -> window address are shifts instead of divisions by 26;
//   * geometry as ONE 16-byte LDS record per (tap, pixel) {f16 w0|w1, w2|w3 (already * mask), window byte offset with the
//     row-parity swizzle folded in, image coordinates for far samples}, read when the tap is sampled (one ds_read_b128
//     broadcast to the pixel's four lanes) instead of being unpacked into 54 VGPRs up front;
//   * out-of-image corners are not tested: a valid in-window sample reads them from the zero-filled window border;
//   * the "some lane samples outside the window" flags come from the staging pass (one ballot per tap);
//   * one barrier per kernel ROW: its three weight taps are one ring stage (two stages, fetched a row ahead), operands of
//     tap t+1 are read during the MFMAs of tap t.
// ------------------------------------------------------------------------------------------
// FUSED: the offset / mask conv (3x3/s1/p1, 27 couts in 32 packed rows, a.w_off / a.b_off) is computed here as well, from the
// same LDS window, before the sampling: per 32-channel chunk nine taps of 4 MFMAs per wave into two accumulator tiles, then
// the 128 x 32 f32 results go through LDS to the geometry stage (and to a.om_out if the caller wants them: training).  One
// kernel and no 112-byte-per-pixel offset tensor instead of two kernels; the sampling pass then walks the chunks backwards,
// starting on the window the offset conv finished with.
// COLS: no contraction at all -- the sampled * mask operands are stored as the training backward's `columns` tensor
// ([M][9*Cin] f16, tap-major; a.y), i.e. modulated_deformable_im2col (kernel.cu:786-868) with the gathers served from the LDS
// window instead of 4 x 16 bytes per (pixel, tap, 8 channels) through L2.
template <typename TOut, bool FUSED, bool COLS = false>
__global__ void __launch_bounds__(256, 2) dcn_window_rows_kernel(const ConvArgs a) {
  constexpr int BC = 64, TH = 8, TW = 16, BP = 128, MG = 4, TP = 2, TC = 4;
  constexpr int WR = TH + 2 + 2 * MG, WCU = TW + 2 + 2 * MG, WCP = 32;   // 18 rows x 26 used of 32 columns
  constexpr int WINB = WR * WCP * 64;                                    // 36864
  constexpr int ROWB = WCP * 64;                                         // 2048 bytes per window row
  constexpr int W_LD = WR / 2;                                           // 9 DMA rounds of two rows
  constexpr int GEOB = 9 * BP * 16;                                      // 18432
  constexpr int WST = BC * 64, STG = 3 * WST, NST = 2;                   // a stage = the three taps of a kernel row
  constexpr int SBB = 2 * BC * 4, FLG = 4 * 2 * 4;
  static_assert(WINB + GEOB + NST * STG + SBB + FLG <= 81920, "two workgroups per CU");
  __shared__ __attribute__((aligned(16))) char smem[WINB + GEOB + NST * STG + SBB + FLG];
  char* const win = smem;
  char* const geo = smem + WINB;
  char* const ring = smem + WINB + GEOB;
  float* const sbuf = (float*)(smem + WINB + GEOB + NST * STG);
  unsigned* const slowf = (unsigned*)(smem + WINB + GEOB + NST * STG + SBB);   // [consumer wave][tap parity]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = a.W / TW, tiles_y = a.H / TH;
  int m_tile, n_tile;
  if (!tile_of_block(a.B * tiles_y * tiles_x, a.Cout_pad / BC, m_tile, n_tile)) return;
  const int tx0 = (m_tile % tiles_x) * TW;
  const int ty0 = ((m_tile / tiles_x) % tiles_y) * TH;
  const int b = m_tile / (tiles_x * tiles_y);
  const int n0 = n_tile * BC;
  const f16* zero = (const f16*)g_zero_page;
  asm volatile("" : "+v"(zero));
  const f16* ximg = (const f16*)a.x + (long)b * a.H * a.W * a.in_stride;
  const int wy0 = ty0 - 1 - MG, wx0 = tx0 - 1 - MG;  // image coordinates of window pixel (0,0)
  const int nch = a.Cin / 32, ns = nch * 3;

  // ---- loaders ----
  const int lrow = tid >> 2, slotw = tid & 3;
  const int tt = lrow >> 4, r16 = lrow & 15;
  const f16* wptr = (const f16*)a.w + (long)(n0 + cout_of<TC>(tt, r16 >> 2, r16 & 3)) * a.Kpad + (slotw ^ swz(lrow)) * 8;
  auto issue_w = [&](int s, int st) {            // row-step s = chunk*3 + kernel row: taps 3s..3s+2 of the chunk-major K order
#pragma unroll
    for (int ts = 0; ts < 3; ++ts) dma16(wptr + (s * 3 + ts) * 32, ring + st * STG + ts * WST + wave * 1024);
  };
  // DMA round i fetches window rows 2i and 2i+1: thread = (row parity, column, 16-byte slot).  Slot sl of an odd row holds
  // channel group sl^2 (the 16 lanes of a ds_read_b128 phase then touch 16 different bank groups).
  const int wrs = tid >> 7, wcol = (tid >> 2) & 31, wsl = tid & 3;
  const int wx = wx0 + wcol;
  const bool wxok = wcol < WCU && wx >= 0 && wx < a.W;
  const int wrow_stride = 2 * a.W * a.in_stride;
  const int wbase = ((wy0 + wrs) * a.W + wx) * a.in_stride + (wsl ^ (2 * wrs)) * 8;
  auto issue_window = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < W_LD; ++i) {
      const bool ok = wxok && (unsigned)(wy0 + wrs + 2 * i) < (unsigned)a.H;
      dma16(ok ? ximg + wbase + i * wrow_stride + chunk * 32 : zero, win + (2 * i) * ROWB + wave * 1024);
    }
  };
  issue_window(0);
  if constexpr (!FUSED && !COLS) issue_w(0, 0);
  // consumer mapping: lane (fr, q) = tile pixel (row 2*wave + fr/8, col 8p + fr%8), channels 8q..8q+7 of the chunk
  const int fr = lane & 15, q = lane >> 4;
  const int prow = 2 * wave + (fr >> 3), pcol = fr & 7;
  float* const om_lds = (float*)geo;                     // FUSED: [128 pixels][32] f32, before the geometry records take the space
  if constexpr (FUSED) {
    constexpr int OSTG = 3 * 2048;                       // stage = 3 taps x 32 rows x 64 bytes
    const int orow = (tid >> 2) & 31, otap = wave >> 1;  // waves 0-1: tap A (rows 0-15 / 16-31), waves 2-3: tap A+1
    const f16* woptr = (const f16*)a.w_off + (long)orow * a.Kpad + (slotw ^ swz(orow)) * 8;
    auto issue_wo = [&](int s, int st) {                 // row-step s = chunk*3 + kernel row of the chunk-major K order
      dma16(woptr + (s * 3 + otap) * 32, ring + st * OSTG + otap * 2048 + (wave & 1) * 1024);
      if (wave < 2) dma16(woptr + (s * 3 + 2) * 32, ring + st * OSTG + 2 * 2048 + (wave & 1) * 1024);
    };
    issue_wo(0, 0);
    f32x4 oacc[TP][2];
#pragma unroll
    for (int p = 0; p < TP; ++p) oacc[p][0] = oacc[p][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // pixel fragments straight from the window: tap (tr, ts) of tile pixel (prow, 8p + pcol) is window pixel
    // (prow + MG + tr, 8p + pcol + MG + ts); the slot swizzle flips with the row parity
    unsigned pa[TP][2];
#pragma unroll
    for (int p = 0; p < TP; ++p) {
      const unsigned base = (unsigned)(((prow + MG) * WCP + 8 * p + pcol + MG) * 64);
      pa[p][0] = base + (((unsigned)q << 4) ^ ((unsigned)(prow & 1) << 5));
      pa[p][1] = pa[p][0] ^ 32u;
    }
    const char* fragO = ring + fr * 64 + ((q ^ swz(fr)) << 4);
    for (int s = 0; s < ns; ++s) {
      const int tr = s % 3, st = s & 1;
      wait_vmcnt<0>();                                   // this row's offset-conv weights (and, tr == 0, the chunk's window)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (s + 1 < ns) issue_wo(s + 1, st ^ 1);
#pragma unroll
      for (int ts = 0; ts < 3; ++ts) {
        const f16x8 w0 = *(const f16x8*)(fragO + st * OSTG + ts * 2048);
        const f16x8 w1 = *(const f16x8*)(fragO + st * OSTG + ts * 2048 + 1024);
#pragma unroll
        for (int p = 0; p < TP; ++p) {
          const f16x8 px = *(const f16x8*)(win + pa[p][tr & 1] + tr * ROWB + ts * 64);
          oacc[p][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, px, oacc[p][0], 0, 0, 0);
          oacc[p][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, px, oacc[p][1], 0, 0, 0);
        }
      }
      if (tr == 2 && s + 1 < ns) {                       // next chunk's window
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_window(s / 3 + 1);
      }
    }
    // offsets / mask logits of the tile -> LDS (+ global for the caller): D rows = couts 16c + 4q + r, column = pixel
#pragma unroll
    for (int p = 0; p < TP; ++p) {
      const int px = prow * 16 + 8 * p + pcol;
      const long m = (long)(b * a.H + ty0 + prow) * a.W + tx0 + 8 * p + pcol;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int co = 16 * c + 4 * q;
        const f32x4 v = co < 28 ? oacc[p][c] + *(const f32x4*)(a.b_off + co) : oacc[p][c];   // b_off: 28 floats (27 + pad)
        *(f32x4*)(om_lds + px * 32 + co) = v;
        if (a.om_out && co < a.om_out_stride) *(f32x4*)(a.om_out + m * a.om_out_stride + co) = v;
      }
    }
    __syncthreads();   // om tile complete; every wave is done with the offset-conv weight stages
    issue_w((nch - 1) * 3, 0);                           // first row-step of the sampling pass: the last chunk
  }

  if (!COLS && tid < BC) {
    const int c = n0 + tid;
    sbuf[tid] = (a.scale && c < a.Cout) ? a.scale[c] : 1.f;
    sbuf[BC + tid] = (a.bias && c < a.Cout) ? a.bias[c] : 0.f;
  }
  // ---- sampling geometry, once per (pixel, tap): thread = pixel gp, taps gh, gh+2, ... ----
  {
    const int gp = tid & 127, gh = tid >> 7;
    const int py = ty0 + (gp >> 4), pxx = tx0 + (gp & 15);
    const float* omrow = FUSED ? om_lds + gp * 32 : a.om + ((long)(b * a.H + py) * a.W + pxx) * a.om_stride;
    float oh[5], ow[5], om_[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int t = 2 * i + gh;
      const bool on = t < 9;
      oh[i] = on ? omrow[2 * t] : 0.f; ow[i] = on ? omrow[2 * t + 1] : 0.f; om_[i] = on ? omrow[18 + t] : 0.f;
    }
    if constexpr (FUSED) __syncthreads();                // the records below overwrite the om tile
    unsigned far_lo = 0, far_hi = 0;     // taps with a far sample among lanes 0-31 / 32-63 (= consumer waves 2*(w&1), +1)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int t = 2 * i + gh;
      if (t < 9) {
        const int tr = t / 3, ts = t - 3 * tr;
        const float h_im = (float)(py - 1 + tr) + oh[i], w_im = (float)(pxx - 1 + ts) + ow[i];
        const float mk = a.mask_is_prob ? om_[i] : __builtin_amdgcn_rcpf(1.f + __expf(-om_[i]));
        const float fh = floorf(h_im), fw = floorf(w_im);
        const int h_low = (int)fh, w_low = (int)fw;
        const float lh = h_im - fh, lw = w_im - fw, hh = 1.f - lh, hw = 1.f - lw;
        const bool valid = h_im > -1.f && w_im > -1.f && h_im < (float)a.H && w_im < (float)a.W;
        const float mv = valid ? mk : 0.f;
        const float w0 = hh * hw * mv, w1 = hh * lw * mv, w2 = lh * hw * mv, w3 = lh * lw * mv;
        const int wr = h_low - wy0, wcn = w_low - wx0;  // window coordinates of corner 0
        // every in-image corner inside the window => the out-of-image ones are too (1 px outside the image, where the
        // window is zero-filled): the window read is the guarded read of deform_conv_cuda_kernel.cu:683-693
        const bool inside = (unsigned)wr < (unsigned)(WR - 1) && (unsigned)wcn < (unsigned)(WCU - 1);
        const bool oow = valid && !inside;
        const unsigned off = (valid && inside) ? (unsigned)(wr * ROWB + wcn * 64) | ((unsigned)(wr & 1) << 5) : 0u;
        const f16 h0 = (f16)w0, h1 = (f16)w1, h2 = (f16)w2, h3 = (f16)w3;
        u32x4 rec;
        rec.x = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
        rec.y = (unsigned)__builtin_bit_cast(unsigned short, h2) | ((unsigned)__builtin_bit_cast(unsigned short, h3) << 16);
        rec.z = off | (oow ? 0x80000000u : 0u);
        rec.w = ((unsigned)(h_low + 1) << 16) | (unsigned)((w_low + 1) & 0xFFFF);   // far samples: h_low, w_low >= -1
        *(u32x4*)(geo + (t * BP + gp) * 16) = rec;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(oow);
        far_lo |= ((unsigned)bal != 0u ? 1u : 0u) << t;
        far_hi |= ((unsigned)(bal >> 32) != 0u ? 1u : 0u) << t;
      }
    }
    if (lane == 0) {
      const int cw = 2 * (wave & 1);
      slowf[cw * 2 + gh] = far_lo;
      slowf[(cw + 1) * 2 + gh] = far_hi;
    }
  }
  wait_vmcnt<0>();
  __syncthreads();
  const unsigned slow_taps = __builtin_amdgcn_readfirstlane(slowf[wave * 2] | slowf[wave * 2 + 1]);
  const unsigned q4 = (unsigned)q << 4;
  const char* georow = geo + (prow * 16 + pcol) * 16;
  typedef const u32x4 __attribute__((address_space(1)))* gp16;

  // corner fragments of tap t for both pixel tiles (+ the weights): raw[p][0..3], wts[p] = {w01, w23}
  auto gather = [&](int t, int chunk, u32x4 (&raw)[TP][4], unsigned (&wts)[TP][2]) {
    u32x4 g[TP];
#pragma unroll
    for (int p = 0; p < TP; ++p) g[p] = *(const u32x4*)(georow + (t * BP + 8 * p) * 16);
#pragma unroll
    for (int p = 0; p < TP; ++p) {
      const unsigned A = (g[p].z & 0xFFFFu) ^ q4;
      const char* c0p = win + A;
      const char* c2p = win + (A ^ 32u);                  // next window row: the other slot swizzle
      raw[p][0] = *(const u32x4*)(c0p);
      raw[p][1] = *(const u32x4*)(c0p + 64);
      raw[p][2] = *(const u32x4*)(c2p + ROWB);
      raw[p][3] = *(const u32x4*)(c2p + ROWB + 64);
      wts[p][0] = g[p].x; wts[p][1] = g[p].y;
    }
    if ((slow_taps >> t) & 1u) {                          // wave-uniform: the lanes concerned read global memory instead
#pragma unroll
      for (int p = 0; p < TP; ++p) {
        const bool out = g[p].z >> 31;
        const int h_low = (int)(g[p].w >> 16) - 1, w_low = (int)(g[p].w & 0xFFFFu) - 1;
        const bool r0 = out && h_low >= 0, r1 = out && h_low + 1 <= a.H - 1, c0 = w_low >= 0, c1 = w_low + 1 <= a.W - 1;
        const int o0 = (h_low * a.W + w_low) * a.in_stride, o2 = o0 + a.W * a.in_stride;
        const f16* base = ximg + chunk * 32 + q * 8;
        const u32x4 g0 = *(gp16)((r0 && c0) ? base + o0 : zero);
        const u32x4 g1 = *(gp16)((r0 && c1) ? base + o0 + a.in_stride : zero);
        const u32x4 g2 = *(gp16)((r1 && c0) ? base + o2 : zero);
        const u32x4 g3 = *(gp16)((r1 && c1) ? base + o2 + a.in_stride : zero);
        if (out) { raw[p][0] = g0; raw[p][1] = g1; raw[p][2] = g2; raw[p][3] = g3; }
      }
    }
  };
  auto blend = [&](const u32x4 (&raw)[TP][4], const unsigned (&wts)[TP][2], f16x8 (&pf)[TP]) {
#pragma unroll
    for (int p = 0; p < TP; ++p)
      pf[p] = __builtin_bit_cast(f16x8, dcn_blend(raw[p][0], raw[p][1], raw[p][2], raw[p][3], wts[p][0], wts[p][1]));
  };

  f32x4 acc[TP][TC];
#pragma unroll
  for (int p = 0; p < TP; ++p)
#pragma unroll
    for (int c = 0; c < TC; ++c) acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const char* fragB = ring + fr * 64 + ((q ^ swz(fr)) << 4);
  auto frags = [&](int st, int ts, f16x8 (&wf)[TC]) {
#pragma unroll
    for (int c = 0; c < TC; ++c) wf[c] = *(const f16x8*)(fragB + st * STG + ts * WST + c * 1024);
  };

  f16x8 pf[TP], wf[TC];
  {
    u32x4 raw[TP][4];
    unsigned wts[TP][2];
    gather(0, FUSED ? nch - 1 : 0, raw, wts);
    blend(raw, wts, pf);
  }
  // i-th chunk of the sampling pass (FUSED walks them backwards: the offset conv left the last one's window in LDS)
  auto chunk_at = [&](int i) { return FUSED ? nch - 1 - i : i; };
  auto tap = [&](int s, int ci, auto tc) {               // s = 3*ci + kernel row: position in the pass
    constexpr int T = decltype(tc)::value, TS = T % 3;
    const int st = s & 1;
    const int chunk = chunk_at(ci);
    if constexpr (COLS) {                                // pf = sampled * mask of tap T: 8 channels of this lane's two pixels
#pragma unroll
      for (int p = 0; p < TP; ++p) {
        const long m = (long)(b * a.H + ty0 + prow) * a.W + tx0 + 8 * p + pcol;
        *(f16x8*)((f16*)a.y + m * (9L * a.Cin) + T * a.Cin + chunk * 32 + q * 8) = pf[p];
      }
    }
    if (!COLS && TS == 0) {
      wait_vmcnt<0>();                                   // this row's weights (issued a row ago)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (s + 1 < ns) issue_w(chunk_at((s + 1) / 3) * 3 + (s + 1) % 3, st ^ 1);
      frags(st, 0, wf);
    }
    f16x8 wfn[TC], pfn[TP];
    u32x4 raw[TP][4];
    unsigned wts[TP][2];
    if (!COLS && TS < 2) frags(st, TS + 1, wfn);
    if (T < 8) {
      gather(T + 1, chunk, raw, wts);
    } else if (ci + 1 < nch) {                           // tap 8 was sampled during tap 7: the window is free
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue_window(chunk_at(ci + 1));
    }
    if constexpr (!COLS) {
#pragma unroll
      for (int p = 0; p < TP; ++p)
#pragma unroll
        for (int c = 0; c < TC; ++c) acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c], pf[p], acc[p][c], 0, 0, 0);
    }
    if (T < 8) {
      blend(raw, wts, pfn);
    } else if (ci + 1 < nch) {
      wait_vmcnt<0>();                                   // next chunk's window, behind this tap's MFMAs
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      gather(0, chunk_at(ci + 1), raw, wts);
      blend(raw, wts, pfn);
    }
#pragma unroll
    for (int p = 0; p < TP; ++p) pf[p] = pfn[p];
    if (!COLS && TS < 2) {
#pragma unroll
      for (int c = 0; c < TC; ++c) wf[c] = wfn[c];
    }
  };
  for (int ci = 0; ci < nch; ++ci) {
    const int s = ci * 3;
    tap(s, ci, std::integral_constant<int, 0>{});
    tap(s, ci, std::integral_constant<int, 1>{});
    tap(s, ci, std::integral_constant<int, 2>{});
    tap(s + 1, ci, std::integral_constant<int, 3>{});
    tap(s + 1, ci, std::integral_constant<int, 4>{});
    tap(s + 1, ci, std::integral_constant<int, 5>{});
    tap(s + 2, ci, std::integral_constant<int, 6>{});
    tap(s + 2, ci, std::integral_constant<int, 7>{});
    tap(s + 2, ci, std::integral_constant<int, 8>{});
  }

  if constexpr (COLS) return;
  // epilogue: per cout-tile pair h a lane holds 8 consecutive couts (cout_of) of 2 pixels; scale/bias from LDS, 16-byte stores
#pragma unroll
  for (int p = 0; p < TP; ++p) {
    const long m = (long)(b * a.H + ty0 + prow) * a.W + tx0 + 8 * p + pcol;
#pragma unroll
    for (int h = 0; h < TC / 2; ++h) {
      const int cl = h * 32 + q * 8;
      TOut* yp = (TOut*)a.y + m * a.out_stride + n0 + cl;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = acc[p][2 * h + (j >> 2)][j & 3] * sbuf[cl + j] + sbuf[BC + cl + j];
        if (a.act == CTDET_ACT_RELU) t = fmaxf(t, 0.f);
        else if (a.act == CTDET_ACT_SIGMOID_CLAMP) t = fminf(fmaxf(ctdet_sigmoid_exact(t), a.clamp_lo), a.clamp_hi);
        v[j] = t;
      }
      if (n0 + cl + 8 <= a.Cout) {
        if constexpr (sizeof(TOut) == 2) {
          f16x8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (f16)v[j];
          *(f16x8*)yp = o;
        } else {
          *(f32x4*)yp = (f32x4){v[0], v[1], v[2], v[3]};
          *(f32x4*)(yp + 4) = (f32x4){v[4], v[5], v[6], v[7]};
        }
      } else if (n0 + cl + 4 <= a.Cout) {
        if constexpr (sizeof(TOut) == 2) {
          f16x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (f16)v[j];
          *(f16x4*)yp = o;
        } else {
          *(f32x4*)yp = (f32x4){v[0], v[1], v[2], v[3]};
        }
      }
    }
  }
}

// `columns` of the DCNv2 backward from the LDS window (train_bwd.hip's launch_dcn_cols decides when)
int launch_dcn_cols_window(const f16* x, int x_stride, const float* om, int om_stride, f16* col, int B, int H, int W, int Cin,
                           int mask_is_prob, hipStream_t s) {
  ConvArgs a = {};
  a.x = x; a.y = col; a.om = om; a.om_stride = om_stride; a.mask_is_prob = mask_is_prob;
  a.B = B; a.H = H; a.W = W; a.Ho = H; a.Wo = W; a.Cin = Cin; a.in_stride = x_stride; a.Cout = 64; a.Cout_pad = 64;
  a.R = a.S = 3; a.stride = 1; a.pad = 1; a.dil = 1; a.in_dil = 1; a.K = a.Kpad = 9 * Cin; a.M = B * H * W; a.korder = 1;
  const int nbx = B * (H / 8) * (W / 16);
  hipLaunchKernelGGL((dcn_window_rows_kernel<f16, false, true>), dim3(8 * ((nbx + 7) / 8)), dim3(256), 0, s, a);
  CTDET_LAUNCH_CHECK();
  return 0;
}

// geometry the row-step kernel (and with it the fused offset conv) serves
bool dcn_offset_fused_ok(const ConvArgs& a) {
  return a.R == 3 && a.S == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 && a.in_dil == 1 && a.Cin % 32 == 0 && a.Kpad == a.K &&
         a.korder == 1 && a.Cout_pad == 64 && a.H % 8 == 0 && a.W % 16 == 0 && a.H <= 65534 && a.W <= 65534 && !a.res &&
         a.out_stride % 8 == 0 && ((size_t)a.y & 15) == 0 &&
         !(ctdet_tuning_flags() & (CTDET_TUNE_DCN_MIXED | CTDET_TUNE_DCN_WINDOW_V1));
}

template <int BC, int WP, int WC_, typename TOut>
static int launch_dcn_window(const ConvArgs& a, hipStream_t s) {
  const int nbx = a.B * ((a.H + 7) / 8) * ((a.W + 15) / 16), nby = a.Cout_pad / BC;
  dim3 grid(8 * ((nbx + 7) / 8) * nby);
  const bool mixed = (ctdet_tuning_flags() & CTDET_TUNE_DCN_MIXED) != 0;
  if (BC == 64 && !mixed && a.H % 8 == 0 && a.W % 16 == 0 && a.H <= 65534 && a.W <= 65534 &&
      !(ctdet_tuning_flags() & CTDET_TUNE_DCN_WINDOW_V1)) {
    if (a.w_off) hipLaunchKernelGGL((dcn_window_rows_kernel<TOut, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((dcn_window_rows_kernel<TOut, false>), grid, dim3(256), 0, s, a);
    CTDET_LAUNCH_CHECK();
    return 0;
  }
  if (mixed && a.H % 8 == 0 && a.W % 16 == 0 && a.H <= 4094 && a.W <= 4094)
    hipLaunchKernelGGL((dcn_window_kernel<BC, (BC > 64 ? 4 : 8), false, TOut, true>), grid, dim3(256), 0, s, a);
  else if (a.H % 8 == 0 && a.W % 16 == 0)
    hipLaunchKernelGGL((dcn_window_kernel<BC, (BC > 64 ? 4 : 8), false, TOut>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((dcn_window_kernel<BC, (BC > 64 ? 4 : 8), true, TOut>), grid, dim3(256), 0, s, a);
  CTDET_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// Small-channel layers (DLA base_layer 7x7 3->16, level0 3x3 16->16, level1 3x3 16->32 s2; dla.py:212-220):
// HBM-bound shapes where an LDS-tiled GEMM is all overhead (4 MFMAs per barrier).  Here every weight fragment of
// the layer lives in registers for the whole kernel (NK*TC fragments), and the pixel (MFMA B) fragments are read
// straight from global memory: with Cin*2 <= 32 bytes per pixel, 16 consecutive pixels x one k-group is a
// contiguous, fully coalesced run.  No LDS, no barriers.  A wave owns 64 consecutive pixels of one image row
// (requires Wo % 64 == 0, else the generic kernel is used), so b/ho are scalar and no per-lane division exists.
// ------------------------------------------------------------------------------------------
template <int TC, int NK, bool NOCHECK, typename TOut>
__global__ void __launch_bounds__(256) conv_smallc_kernel(const ConvArgs a) {
  // all weights of the layer in LDS: [16*TC couts][NK*32 + 8 pad] f16 (row pitch 16-byte aligned, odd in 16-byte
  // slots so the 16-row fragment reads spread over the banks)
  constexpr int WPITCH = NK * 32 + 8;
  __shared__ __attribute__((aligned(16))) f16 sw[16 * TC * WPITCH];
  const int lane = threadIdx.x & 63;
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const f16* __restrict__ x = (const f16*)a.x;
  const int px = lane & 15, kg = lane >> 4;
  for (int i = threadIdx.x; i < 16 * TC * NK * 4; i += 256) {
    const int row = i / (NK * 4), g8 = i % (NK * 4);
    *(f16x8*)(sw + row * WPITCH + g8 * 8) = *(const f16x8*)((const f16*)a.w + (long)row * a.Kpad + g8 * 8);
  }
  __syncthreads();
  // fragment row of this lane for cout tile c: permuted cout (see the epilogue mapping of the tiled kernels)
  const f16* wrow[TC];
#pragma unroll
  for (int c = 0; c < TC; ++c) wrow[c] = sw + (4 * TC * (px >> 2) + 4 * c + (px & 3)) * WPITCH + kg * 8;

  // per K step: tap geometry of this lane's k-group (k-group G = kt*4 + kg; 8 channels)
  int dh[NOCHECK ? 1 : NK], dw[NOCHECK ? 1 : NK], koff[NK];
  const int gpt = a.Cin >> 3;  // k-groups per tap (1 or 2)
#pragma unroll
  for (int kt = 0; kt < NK; ++kt) {
    const int G = kt * 4 + kg;
    const int tap = G / gpt;
    const int tr = tap / a.S, ts = tap - tr * a.S;
    const bool tail = tr >= a.R;  // K tail: weights are zero there; NOCHECK reads tap (0,0) instead
    const int dhk = tail ? (NOCHECK ? 0 : (1 << 20)) : tr * a.dil;
    const int dwk = tail ? 0 : ts * a.dil;
    if constexpr (!NOCHECK) { dh[kt] = dhk; dw[kt] = dwk; }
    // byte offset of this k-group relative to the lane's pixel of tap (0,0)
    koff[kt] = (((tail ? 0 : dhk) * a.W + dwk) * a.in_stride + (tail ? 0 : (G - tap * gpt) * 8)) * 2;
  }
  int poff[4];  // byte offset of px-tile p's pixel relative to the segment's first pixel
#pragma unroll
  for (int p = 0; p < 4; ++p) poff[p] = (p * 16 + px) * a.stride * a.in_stride * 2;

  const int segs_per_row = a.Wo >> 6;
  const int nseg = a.B * a.Ho * segs_per_row;
  const int q = lane >> 4;
  for (int seg = blockIdx.x * 4 + wave_in_block; seg < nseg; seg += gridDim.x * 4) {
    const int sw_ = seg % segs_per_row, t = seg / segs_per_row;
    const int ho = t % a.Ho, b = t / a.Ho;
    const int hb = ho * a.stride - a.pad;
    const long img = (long)b * a.H * a.W;
    // scalar base: pixel (hb, first column of the segment - pad); per-lane parts are 32-bit byte offsets
    const char* sbase = (const char*)(x + (img + (long)hb * a.W + (sw_ * 64 * a.stride - a.pad)) * a.in_stride);
    auto load_frags = [&](int kt, f16x8 (&af)[4]) {
      if constexpr (NOCHECK) {
#pragma unroll
        for (int p = 0; p < 4; ++p) af[p] = *(const f16x8*)(sbase + (unsigned)(koff[kt] + poff[p]));
      } else {
        const int hi = hb + dh[kt];
        const bool hok = hi >= 0 && hi < a.H;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int wo = sw_ * 64 + p * 16 + px;
          const int wi = wo * a.stride - a.pad + dw[kt];
          const bool ok = hok && wi >= 0 && wi < a.W;
          const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
          af[p] = ok ? *(const f16x8*)(sbase + (koff[kt] + poff[p])) : z;
        }
      }
    };
    f32x4 acc[4][TC];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int c = 0; c < TC; ++c) acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x8 afA[4], afB[4];
    load_frags(0, afA);
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
      // pixel fragments of step kt+1 are requested before the MFMAs of step kt (explicit one-step prefetch)
      if (kt + 1 < NK) { if (kt & 1) load_frags(kt + 1, afA); else load_frags(kt + 1, afB); }
      f16x8 wf[TC];
#pragma unroll
      for (int c = 0; c < TC; ++c) wf[c] = *(const f16x8*)(wrow[c] + kt * 32);
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < TC; ++c)
          acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c], (kt & 1) ? afB[p] : afA[p], acc[p][c], 0, 0, 0);
    }
    const int mrow = (b * a.Ho + ho) * a.Wo + sw_ * 64;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int c = 0; c < TC; ++c) epilogue_store4<TOut>(a, mrow + p * 16 + px, 4 * TC * q + 4 * c, acc[p][c]);
  }
}

template <int TC, int NK, typename TOut>
static int launch_smallc(const ConvArgs& a, hipStream_t s) {
  const int nseg = a.B * a.Ho * (a.Wo / 64);
  int blocks = (nseg + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (a.pad == 0)  // pre-padded input (zero frame in memory): every tap of every pixel is in bounds
    hipLaunchKernelGGL((conv_smallc_kernel<TC, NK, true, TOut>), dim3(blocks), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv_smallc_kernel<TC, NK, false, TOut>), dim3(blocks), dim3(256), 0, s, a);
  CTDET_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// Small-channel layers, LDS-window form.  conv_smallc_kernel above re-reads every input pixel R*S times through L1
// (the 7x7 stem 49 times): rocprof shows it L1-bandwidth bound at ~7x its HBM time.  Here a workgroup owns a TH x TW
// output tile, brings the (TH*s + R - s) x (TW*s + R - s) input window into LDS once (dense rows, 16 or 32 bytes per
// pixel) and every MFMA B fragment is one ds_read_b128 from it: lane (pixel, k-group) reads the 8 channels of tap
// (r, s) at window pixel (y*s + r, x*s + s').  The weight fragments of the whole layer stay in registers (NK*TC*4
// VGPRs).  One LDS address register per K step and lane; tile row/column offsets are instruction immediates.
// ------------------------------------------------------------------------------------------
template <int R, int CIN, int TC, int STRIDE, bool NOCHECK, typename TOut>
__global__ void __launch_bounds__(256) conv_win_kernel(const ConvArgs a) {
  constexpr int TH = STRIDE == 1 ? 16 : 8, TW = STRIDE == 1 ? 64 : 32;
  constexpr int WH = (TH - 1) * STRIDE + R, WW = (TW - 1) * STRIDE + R;
  constexpr int PB = CIN * 2;                       // bytes per pixel
  constexpr int PPR = WW * PB / 16;                 // 16-byte pieces per window row
  constexpr int NP = WH * PPR, ROUNDS = (NP + 255) / 256;
  constexpr int K = R * R * CIN, NK = (K + 31) / 32;
  constexpr int GPT = CIN / 8;                      // k-groups per tap
  constexpr int NT = TW / 16, RW = TH / 4;          // pixel tiles per tile row; tile rows per wave
  __shared__ __attribute__((aligned(16))) char win[ROUNDS * 4096];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, q = lane >> 4;
  const int tiles_x = a.Wo / TW, tiles_y = a.Ho / TH;
  const int tx0 = (blockIdx.x % tiles_x) * TW;
  const int ty0 = ((blockIdx.x / tiles_x) % tiles_y) * TH;
  const int b = blockIdx.x / (tiles_x * tiles_y);
  const f16* zero = (const f16*)g_zero_page;
  asm volatile("" : "+v"(zero));
  const f16* ximg = (const f16*)a.x + (long)b * a.H * a.W * a.in_stride;
  const int iy0 = ty0 * STRIDE - a.pad, ix0 = tx0 * STRIDE - a.pad;   // image coordinates of window pixel (0,0)

  // ---- window DMA: piece pid -> window row pid / PPR, 16-byte column pid % PPR (pixels are contiguous in memory:
  // in_stride == CIN) ----
#pragma unroll
  for (int i = 0; i < ROUNDS; ++i) {
    const int pid = tid + 256 * i;
    const int wr = pid / PPR, cp = pid - wr * PPR;
    const int y = iy0 + wr, x = ix0 + cp / (PB / 16);
    bool ok = pid < NP;
    if constexpr (!NOCHECK) ok = ok && y >= 0 && y < a.H && x >= 0 && x < a.W;
    dma16(ok ? ximg + ((long)y * a.W + ix0) * CIN + cp * 8 : zero, win + (wave * 64 + 256 * i) * 16);
  }

  // ---- weights: fragment (kt, c) of this lane = 8 k of cout row cout_of(c, fr/4, fr%4), straight into registers ----
  f16x8 wf[NK][TC];
#pragma unroll
  for (int c = 0; c < TC; ++c) {
    const f16* wr = (const f16*)a.w + (long)cout_of<TC>(c, fr >> 2, fr & 3) * a.Kpad + q * 8;
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) wf[kt][c] = *(const f16x8*)(wr + kt * 32);
  }
  // ---- per K step: LDS byte address of this lane's k-group for tile pixel (row 0 of the wave, column fr) ----
  int kaddr[NK];
#pragma unroll
  for (int kt = 0; kt < NK; ++kt) {
    const int G = kt * 4 + q;
    const int tap = G / GPT;
    const bool tail = tap >= R * R;                  // K tail: zero weights; read tap (0,0)
    const int tr = tail ? 0 : tap / R, ts = tail ? 0 : tap - (tap / R) * R;
    kaddr[kt] = ((wave * RW * STRIDE + tr) * WW + fr * STRIDE + ts) * PB + (tail ? 0 : (G - tap * GPT) * 16);
  }
  wait_vmcnt<0>();
  __syncthreads();

#pragma unroll
  for (int rr = 0; rr < RW; ++rr) {
#pragma unroll
    for (int tc = 0; tc < NT; ++tc) {
      constexpr int dummy = 0; (void)dummy;
      const int toff = (rr * STRIDE * WW + tc * 16 * STRIDE) * PB;   // compile time after unrolling
      f32x4 acc[TC];
#pragma unroll
      for (int c = 0; c < TC; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < NK; ++kt) {
        const f16x8 pf = *(const f16x8*)(win + kaddr[kt] + toff);
#pragma unroll
        for (int c = 0; c < TC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kt][c], pf, acc[c], 0, 0, 0);
      }
      const int m = (b * a.Ho + ty0 + wave * RW + rr) * a.Wo + tx0 + tc * 16 + fr;
      epilogue_tiles<TOut, TC>(a, m, 0, q, acc);
    }
  }
}

template <int R, int CIN, int TC, int STRIDE, typename TOut>
static int launch_win(const ConvArgs& a, hipStream_t s) {
  constexpr int TH = STRIDE == 1 ? 16 : 8, TW = STRIDE == 1 ? 64 : 32;
  const int blocks = a.B * (a.Ho / TH) * (a.Wo / TW);
  if (a.pad == 0)  // pre-padded input (zero frame in memory): every tap of every pixel is in bounds
    hipLaunchKernelGGL((conv_win_kernel<R, CIN, TC, STRIDE, true, TOut>), dim3(blocks), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv_win_kernel<R, CIN, TC, STRIDE, false, TOut>), dim3(blocks), dim3(256), 0, s, a);
  CTDET_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------


template <int BP, int BC, int WP, int WC_, typename TOut>
static int launch_dma(const ConvArgs& a, hipStream_t s) {
  const int nbx = (a.M + BP - 1) / BP, nby = a.Cout_pad / BC;
  dim3 grid(8 * ((nbx + 7) / 8) * nby);
  hipLaunchKernelGGL((conv_igemm_dma_kernel<BP, BC, WP, WC_, TOut>), grid, dim3(256), 0, s, a);
  CTDET_LAUNCH_CHECK();
  return 0;
}

template <int BP, int BC, int WP, int WC_, typename TOut>
static int launch_uk(const ConvArgs& a, hipStream_t s) {
  const int nbx = (a.M + BP - 1) / BP, nby = a.Cout_pad / BC;
  dim3 grid(8 * ((nbx + 7) / 8) * nby);
  if (a.nsrc > 1)
    hipLaunchKernelGGL((conv_igemm_uk_kernel<BP, BC, WP, WC_, true, TOut>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv_igemm_uk_kernel<BP, BC, WP, WC_, false, TOut>), grid, dim3(256), 0, s, a);
  CTDET_LAUNCH_CHECK();
  return 0;
}

static bool uniform_k_ok(const ConvArgs& a) {
  if (a.R * a.S > 32 || a.Kpad != a.K || a.in_dil != 1) return false;
  if (a.korder == 0 && a.R * a.S > 1) return false;  // the uniform-K kernel walks k chunk-major
  if (a.nsrc > 1) {
    int prev = 0;
    for (int j = 0; j < a.nsrc; ++j) {
      if ((a.xs_cend[j] - prev) % 32) return false;
      prev = a.xs_cend[j];
    }
    return true;
  }
  return a.Cin % 32 == 0;
}

template <typename TOut>
static int launch_conv_f16_t(const ConvArgs& a, bool deform, hipStream_t s) {
  const int bc = pick_bc(a.Cout);
  CTDET_CHECK(a.Cout_pad % bc == 0 && a.Cout_pad >= a.Cout, "conv: Cout_pad=%d does not match tile %d (Cout=%d)",
              a.Cout_pad, bc, a.Cout);
  CTDET_CHECK(a.in_dil >= 1 && (!deform || a.in_dil == 1), "conv: bad in_dil %d", a.in_dil);
  if (deform && a.korder == 1) {
    // chunk-major weights: LDS-window kernel (3x3/s1/p1, map divisible by the 8x16 tile)
    CTDET_CHECK(a.R == 3 && a.S == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 && a.Cin % 32 == 0 && a.Kpad == a.K && (a.w_off || (a.om_stride % 4 == 0 && ((size_t)a.om & 15) == 0)) && !a.res &&
                    a.out_stride % 8 == 0 && ((size_t)a.y & 15) == 0,
                "dcnv2(window): unsupported geometry");
    // 64- or 128-cout tiles whatever Cout is: the packed rows are padded to a multiple of 64 (zero rows; nothing beyond
    // Cout is stored)
    CTDET_CHECK(a.Cout_pad % 64 == 0, "dcnv2(window): weight rows must be padded to a multiple of 64 (Cout=%d, packed %d)",
                a.Cout, a.Cout_pad);
    CTDET_CHECK(!a.w_off || dcn_offset_fused_ok(a), "dcnv2(offset conv fused): unsupported geometry (ctdet_dcnv2_offset_supported)");
    if (!a.w_off && a.Cout > 64 && a.Cout_pad % 128 == 0) return launch_dcn_window<128, 2, 2, TOut>(a, s);
    return launch_dcn_window<64, 2, 2, TOut>(a, s);
  }
  CTDET_CHECK(!deform, "dcnv2: the f16 path takes chunk-major weights (korder 1, Cin %% 32 == 0); got korder %d, Cin %d",
              a.korder, a.Cin);
  // enough pixel tiles to fill 256 CUs with the big tile? otherwise use the 128-pixel variants
  const long tiles256 = ((long)a.M + 255) / 256 * (a.Cout_pad / bc);
  const bool big = tiles256 >= 512;
  CTDET_CHECK(a.R * a.S <= 64, "conv: at most 64 taps (R*S=%d)", a.R * a.S);
  if ((a.Cin == 8 || a.Cin == 16) && a.in_dil == 1 && a.korder == 0 && a.nsrc <= 1 && a.Wo % 64 == 0 && a.Cout_pad <= 32 && a.Cout_pad == bc) {
    const int nk = a.Kpad / 32;
    // LDS-window form for the three DLA base layers (tile-divisible maps, contiguous pixels)
    if (!(ctdet_tuning_flags() & CTDET_TUNE_NO_WIN) && a.R == a.S && a.dil == 1 && a.in_stride == a.Cin && a.Kpad == nk * 32) {
      if (a.R == 7 && a.Cin == 8 && bc == 16 && a.stride == 1 && a.Ho % 16 == 0 && a.Wo % 64 == 0 && (a.pad == 0 || a.pad == 3))
        return launch_win<7, 8, 1, 1, TOut>(a, s);
      if (a.R == 3 && a.Cin == 16 && bc == 16 && a.stride == 1 && a.Ho % 16 == 0 && a.Wo % 64 == 0 && a.pad == 1)
        return launch_win<3, 16, 1, 1, TOut>(a, s);
      if (a.R == 3 && a.Cin == 16 && bc == 32 && a.stride == 2 && a.Ho % 8 == 0 && a.Wo % 32 == 0 && a.pad == 1)
        return launch_win<3, 16, 2, 2, TOut>(a, s);
    }
    if (nk == 13 && bc == 16) return launch_smallc<1, 13, TOut>(a, s);
    if (nk == 5 && bc == 16) return launch_smallc<1, 5, TOut>(a, s);
    if (nk == 5 && bc == 32) return launch_smallc<2, 5, TOut>(a, s);
    if (nk == 13 && bc == 32) return launch_smallc<2, 13, TOut>(a, s);
  }
  if (halo_ok(a) && a.Cin % 64 == 0 && a.Kpad == 9 * a.Cin && !(ctdet_tuning_flags() & CTDET_TUNE_NO_HALO_TAP2)) {
    if (bc == 32) return launch_halo_tap2<32, 4, 1, TOut>(a, s);
    if (bc == 64 || bc == 128) return launch_halo_tap2<64, 4, 1, TOut>(a, s);
  }
  // 16 x 16-pixel tiles of the two-tap kernel: maps the 8 x 32 tile does not divide (the 512-channel level at 16 x 16)
  if (halo16_ok(a) && a.Cin % 64 == 0 && a.Kpad == 9 * a.Cin && !(ctdet_tuning_flags() & CTDET_TUNE_NO_HALO_TAP2)) {
    if (bc == 32) return launch_halo_tap2<32, 4, 1, TOut, 16>(a, s);
    if (bc == 64 || bc == 128) return launch_halo_tap2<64, 4, 1, TOut, 16>(a, s);
  }
  if (halo_ok(a)) {
    switch (bc) {
      case 32: return launch_halo<32, 4, 1, TOut>(a, s);
      case 64: return launch_halo<64, 4, 1, TOut>(a, s);
      case 128: {
        // fewer workgroups than the chip holds (2 per CU): 64-cout tiles double them.  Training batches on the 64^2 / 32^2
        // levels: 128->128 @64^2, batch 16: 28.7 -> 24.2 us; 256->256 @32^2: 42.4 -> 27.7 us (32-cout tiles: no further gain)
        const long wgs = (long)a.B * (a.H / 8) * (a.W / 32) * (a.Cout_pad / 128);
        if (wgs < 512 && !(ctdet_tuning_flags() & CTDET_TUNE_NO_SMALL_GRID_TILES)) return launch_halo<64, 4, 1, TOut>(a, s);
        return launch_halo<128, 2, 2, TOut>(a, s);
      }
    }
  }
  if (uniform_k_ok(a)) {
    switch (bc) {
      case 16: return launch_uk<256, 16, 4, 1, TOut>(a, s);
      case 32: return big ? launch_uk<256, 32, 4, 1, TOut>(a, s) : launch_uk<128, 32, 4, 1, TOut>(a, s);
      case 64: return big ? launch_uk<256, 64, 4, 1, TOut>(a, s) : launch_uk<128, 64, 2, 2, TOut>(a, s);
      case 128: {
        const long wgs128 = ((long)a.M + 127) / 128 * (a.Cout_pad / 128);
        if (!big && wgs128 < 512 && !(ctdet_tuning_flags() & CTDET_TUNE_NO_SMALL_GRID_TILES))   // as for the halo kernel
          return launch_uk<128, 64, 2, 2, TOut>(a, s);
        return big ? launch_uk<256, 128, 2, 2, TOut>(a, s) : launch_uk<128, 128, 2, 2, TOut>(a, s);
      }
    }
  }
  switch (bc) {
    case 16: return launch_dma<256, 16, 4, 1, TOut>(a, s);
    case 32: return big ? launch_dma<256, 32, 4, 1, TOut>(a, s) : launch_dma<128, 32, 4, 1, TOut>(a, s);
    case 64: return big ? launch_dma<256, 64, 4, 1, TOut>(a, s) : launch_dma<128, 64, 2, 2, TOut>(a, s);
    case 128: return big ? launch_dma<256, 128, 2, 2, TOut>(a, s) : launch_dma<128, 128, 2, 2, TOut>(a, s);
  }
  CTDET_CHECK(false, "conv: no tile for Cout=%d", a.Cout);
}

int launch_conv_f16(const ConvArgs& a, int out_dtype, bool deform, hipStream_t s) {
  CTDET_CHECK(a.Cin % 8 == 0 && a.in_stride % 8 == 0, "conv(f16): Cin=%d / in_stride=%d must be multiples of 8", a.Cin,
              a.in_stride);
  CTDET_CHECK(a.Cout % 4 == 0 && a.out_stride % 4 == 0, "conv(f16): Cout=%d / out_stride=%d must be multiples of 4",
              a.Cout, a.out_stride);
  CTDET_CHECK(a.Kpad % 32 == 0 && a.Kpad >= a.K, "conv(f16): Kpad=%d invalid for K=%d", a.Kpad, a.K);
  CTDET_CHECK((long)a.B * a.H * a.W * a.in_stride < (1L << 31), "conv: input too large for 32-bit element offsets");
  if (out_dtype == CTDET_F16) return launch_conv_f16_t<f16>(a, deform, s);
  if (out_dtype == CTDET_F32) return launch_conv_f16_t<float>(a, deform, s);
  CTDET_CHECK(false, "conv: bad out dtype %d", out_dtype);
}
