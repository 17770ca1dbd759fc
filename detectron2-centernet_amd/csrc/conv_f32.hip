// f32 mode of the conv-shaped contractions (the reference's own arithmetic: fp32 in, fp32 accumulate) on the gfx950
// f32 matrix pipe: v_mfma_f32_16x16x4_f32 is bit-for-bit a k-ordered f32 fmaf chain at the f32 vector peak
// (157 TFLOP/s), so this path keeps reference precision while the contraction runs on MFMA.
//   conv_f32_mfma_kernel   implicit GEMM of any plain conv / Root (multi-source 1x1) / input-dilated conv:
//                          D[cout][pixel] = sum_k W[cout][k] * im2col(x)[pixel][k], k tap-major; 16-k LDS tiles
//                          ([row][16 f32] = the same 64-byte swizzled rows as the f16 kernels), LDS-DMA 3-stage ring
//   dcn_f32_mfma_kernel    DCNv2 (deform_conv_cuda_kernel.cu:666-868 + deform_conv_cuda.cu:874-927): the four bilinear
//                          corners of a (pixel, tap, 4 channels) are gathered as float4s, blended in f32 exactly as the
//                          reference's im2col does (val = sum wt_q * v_q; val * mask) and written to the LDS tile the MFMAs
//                          read -- no `columns` buffer
//   conv_direct_f32_kernel one thread per (pixel, cout): shapes the vector paths cannot take (Cin or strides not
//                          multiples of 4 floats)
// Weights: f32 [Cout_pad][Kpad], k = (r*S + s)*Cin + c, Kpad = roundup(K, 16), zero padded.
// Every kernel has a second instantiation SP = true, the f16x3 ("split") mode: the same f32 activations, LDS images, DMA
// schedule and epilogue, with each 16-k step contracted by two v_mfma_f32_16x16x32_f16 on hi/lo f16 halves (conv_common.h:
// split_b / mma_px; weights pre-split by ctdet_split_weights) instead of four v_mfma_f32_16x16x4_f32.
#include "conv_common.h"

template <int BP, int BC, int WP, int WC_, bool SP>
__global__ void __launch_bounds__(256) conv_f32_mfma_kernel(const ConvArgs a) {
  constexpr int KS = 16, EPV = 4;            // k per LDS row (64 bytes), elements per 16-byte vector
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
  const float* __restrict__ x = (const float*)a.x;
  const float* __restrict__ w = (const float*)a.w;
  const float* zero = (const float*)g_zero_page;   // opaque: one v_cndmask + ONE DMA per staged row (see conv_igemm.hip)
  asm volatile("" : "+v"(zero));

  const int lrow = tid >> 2, slot = tid & 3;
  const int g = slot ^ swz(lrow);
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
    b_off[j] = (long)(n0 + (b_ok[j] ? cl : 0)) * a.Kpad + g * EPV;
  }

  // per-lane k state (tap-major k): this lane's 4 channels of the current K step sit in tap (tr, ts) at channel c0
  int c0, tr, ts;
  {
    const int kc = g * EPV, tap = kc / a.Cin;
    c0 = kc - tap * a.Cin;
    tr = tap / a.S;
    ts = tap - tr * a.S;
  }
  auto advance_k = [&]() {
    c0 += KS;
    while (c0 >= a.Cin) {
      c0 -= a.Cin;
      if (++ts == a.S) { ts = 0; ++tr; }
    }
  };

  auto issue = [&](int kt, int stage) {
    char* sb = smem + stage * STAGE + wave * 1024;
    if (a.nsrc > 1) {
      const int kk = kt * KS + g * EPV;
      const float* src = (const float*)a.xs[0];
      int st = a.xs_stride[0], cb0 = 0;
      if (kk >= a.xs_cend[0]) { src = (const float*)a.xs[1]; st = a.xs_stride[1]; cb0 = a.xs_cend[0]; }
      if (a.nsrc > 2 && kk >= a.xs_cend[1]) { src = (const float*)a.xs[2]; st = a.xs_stride[2]; cb0 = a.xs_cend[1]; }
      if (a.nsrc > 3 && kk >= a.xs_cend[2]) { src = (const float*)a.xs[3]; st = a.xs_stride[3]; cb0 = a.xs_cend[2]; }
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
    for (int j = 0; j < B_LD; ++j) dma16(b_ok[j] ? w + b_off[j] + kt * KS : zero, sb + BP * 64 + j * 4096);
  };

  f32x4 acc[TP][TC];
#pragma unroll
  for (int p = 0; p < TP; ++p)
#pragma unroll
    for (int c = 0; c < TC; ++c) acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15;
  const int frag_off = fr * 64 + (((lane >> 4) ^ swz(fr)) << 4);
  const int nk = a.Kpad / KS;

  issue(0, 0);
  advance_k();
  if (nk > 1) { issue(1, 1); advance_k(); }

  int st_c = 0, st_l = 2;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) wait_vmcnt<NLOAD>(); else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + 2 < nk) { issue(kt + 2, st_l); advance_k(); }
    const char* base = smem + st_c * STAGE;
    // a lane's 16 bytes = k {4q .. 4q+3} of its row (q = lane / 16): element e of both fragments is the operand of the
    // e-th 16x16x4 MFMA of this K step (its k index lane/16 then stands for k = 4q + e on both sides)
    f32x4 wf[TC], pf[TP];
#pragma unroll
    for (int c = 0; c < TC; ++c) wf[c] = *(const f32x4*)(base + BP * 64 + (wc * 16 * TC + 16 * c) * 64 + frag_off);
#pragma unroll
    for (int p = 0; p < TP; ++p) pf[p] = *(const f32x4*)(base + (wp * 16 * TP + 16 * p) * 64 + frag_off);
    if constexpr (SP) {
#pragma unroll
      for (int p = 0; p < TP; ++p) mma_px<true, TC>(wf, pf[p], acc[p]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int p = 0; p < TP; ++p)
#pragma unroll
          for (int c = 0; c < TC; ++c)
            acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[c][e], pf[p][e], acc[p][c], 0, 0, 0);
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
    epilogue_tiles<float, TC>(a, m, cb, q, acc[p]);
  }
}

// ------------------------------------------------------------------------------------------
// Uniform-K form (Cin and every concat source a multiple of 16): the tap / channel position of a K step is the same
// for every lane and lives in SGPRs; per lane only a row pointer and a 32-bit tap-validity mask remain, which fits
// the 128 accumulators of the 256x128 tile into 256 registers -> two workgroups per CU, so one workgroup's LDS-DMA
// issue, barrier and fragment reads overlap the other's MFMAs.  k stays tap-major (the order of the packed weights).
// ------------------------------------------------------------------------------------------
template <int BP, int BC, int WP, int WC_, bool CAT, bool SP>
__global__ void __launch_bounds__(256, 2) conv_f32_uk_kernel(const ConvArgs a) {
  constexpr int KS = 16, EPV = 4;
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
  const float* zero = (const float*)g_zero_page;
  asm volatile("" : "+v"(zero));

  const int lrow = tid >> 2, slot = tid & 3;
  const int g = slot ^ swz(lrow);

  const float* rowp[A_LD];  // conv: pixel of tap (0,0) + g*4 channels; cat: row of the current source + g*4
  unsigned tapmask[A_LD];   // conv: bit t = tap t inside the image; cat: bit 0 = row < M
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
      rowp[i] = (const float*)a.xs[0] + (long)mm * a.xs_stride[0] + g * EPV;
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
      rowp[i] = (const float*)a.x + ((long)b * a.H * a.W + (long)hb * a.W + wb) * a.in_stride + g * EPV;
    }
    tapmask[i] = mk;
  }
  const float* wptr[B_LD];
#pragma unroll
  for (int j = 0; j < B_LD; ++j) {
    const int L = lrow + 64 * j;
    const int Lw = L % (16 * TC), wv = L / (16 * TC);
    const int tt = Lw >> 4, r = Lw & 15;
    const int cl = wv * 16 * TC + cout_of<TC>(tt, r >> 2, r & 3);
    wptr[j] = (const float*)a.w + (long)(n0 + (L < BC ? cl : 0)) * a.Kpad + g * EPV;
  }

  // uniform K-step state (tap-major): 16 channels further inside the tap, then the next tap of the row, then the next row
  const long step_s = (long)a.dil * a.in_stride - a.Cin;
  const long step_r = ((long)a.dil * a.W - (long)(a.S - 1) * a.dil) * a.in_stride - a.Cin;
  long dlt = 0;
  unsigned tbit = 1u;
  int ts = 0, cc = 0;
  int sj = 0, sleft = CAT ? a.xs_cend[0] : 0;
  long woff = 0;
  auto advance_k = [&]() {
    woff += KS;
    dlt += KS;
    if constexpr (CAT) {
      sleft -= KS;
      if (sleft == 0 && sj + 1 < a.nsrc) {
        ++sj;
        sleft = a.xs_cend[sj] - a.xs_cend[sj - 1];
        dlt = 0;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) rowp[i] = (const float*)a.xs[sj] + (long)rowm[i] * a.xs_stride[sj] + g * EPV;
      }
    } else {
      cc += KS;
      if (cc == a.Cin) {
        cc = 0;
        tbit <<= 1;
        if (++ts < a.S) dlt += step_s; else { ts = 0; dlt += step_r; }
      }
    }
  };
  auto issue = [&](char* sb) {
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
  const char* fragA = smem + (wp * 16 * TP) * 64 + frag_off;
  const char* fragB = smem + BP * 64 + (wc * 16 * TC) * 64 + frag_off;
  char* dmab = smem + wave * 1024;
  const int nk = a.K / KS;   // K = R*S*Cin, Cin % 16 == 0

  issue(dmab);
  advance_k();
  if (nk > 1) { issue(dmab + STAGE); advance_k(); }

  auto kstep = [&](int kt, auto st_c, auto st_l) {
    constexpr int ST = decltype(st_c)::value, SL = decltype(st_l)::value;
    if (kt + 1 < nk) wait_vmcnt<NLOAD>(); else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + 2 < nk) { issue(dmab + SL * STAGE); advance_k(); }
    f32x4 wf[TC];
#pragma unroll
    for (int c = 0; c < TC; ++c) wf[c] = *(const f32x4*)(fragB + ST * STAGE + c * 1024);
#pragma unroll
    for (int p = 0; p < TP; ++p) {
      const f32x4 pf = *(const f32x4*)(fragA + ST * STAGE + p * 1024);
      mma_px<SP, TC>(wf, pf, acc[p]);
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
    epilogue_tiles<float, TC>(a, m, cb, q, acc[p]);
  }
}

// ------------------------------------------------------------------------------------------
// DCNv2, f32: Cin % 16 == 0, so the 16 k of a K step belong to one tap; the sampling geometry of a (pixel, tap) is
// recomputed when the tap changes (every Cin/16 steps) and kept in registers.  Per K step a thread gathers the four
// corners of its (row, 4 channels) for the NEXT step before the MFMAs of the current one, blends after them and
// writes the blended float4 where the LDS-DMA of the plain kernel would have put it; weights stream by LDS-DMA.
// ------------------------------------------------------------------------------------------
template <int BP, int BC, int WP, int WC_, bool SP>
__global__ void __launch_bounds__(256, 2) dcn_f32_mfma_kernel(const ConvArgs a) {
  constexpr int KS = 16, EPV = 4;
  constexpr int TP = BP / WP / 16;
  constexpr int TC = BC / WC_ / 16;
  constexpr int A_LD = BP / 64;
  constexpr int BCL = BC < 64 ? 64 : BC;
  constexpr int B_LD = BCL / 64;
  constexpr int ASTAGE = BP * 64, WSTAGE = BCL * 64;
  static_assert(WP * WC_ == 4, "4 waves per workgroup");
  __shared__ __attribute__((aligned(16))) char smem[2 * ASTAGE + 2 * WSTAGE];
  char* const smA = smem;
  char* const smW = smem + 2 * ASTAGE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave / WC_, wc = wave % WC_;
  int m_tile, n_tile;
  if (!tile_of_block((a.M + BP - 1) / BP, a.Cout_pad / BC, m_tile, n_tile)) return;
  const int m0 = m_tile * BP, n0 = n_tile * BC;
  const float* __restrict__ x = (const float*)a.x;
  const float* __restrict__ w = (const float*)a.w;
  const float* zero = (const float*)g_zero_page;
  asm volatile("" : "+v"(zero));

  const int lrow = tid >> 2, slot = tid & 3;
  const int g = slot ^ swz(lrow);
  bool rok[A_LD];
  int rpix[A_LD], rhb[A_LD], rwb[A_LD];
  const float* omrow[A_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const int m = m0 + lrow + 64 * i;
    rok[i] = m < a.M;
    const int mm = rok[i] ? m : 0;
    const int wo = mm % a.Wo, t = mm / a.Wo;
    const int ho = t % a.Ho, b = t / a.Ho;
    rpix[i] = b * a.H * a.W;
    rhb[i] = ho * a.stride - a.pad;
    rwb[i] = wo * a.stride - a.pad;
    omrow[i] = a.om + (long)mm * a.om_stride;
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
    b_off[j] = (long)(n0 + (b_ok[j] ? cl : 0)) * a.Kpad + g * EPV;
  }

  int goff[A_LD][4];
  float gwt[A_LD][4], gmask[A_LD];
  auto geometry = [&](int tap) {
    const int tr = tap / a.S, ts = tap - tr * a.S;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      DcnSample sp;
      dcn_setup(a, rok[i], rpix[i], rhb[i], rwb[i], tr, ts, omrow[i], sp);
#pragma unroll
      for (int q = 0; q < 4; ++q) { goff[i][q] = sp.off[q]; gwt[i][q] = sp.wt[q]; }
      gmask[i] = sp.mask;
    }
  };
  f32x4 cv[A_LD][4];
  auto gather = [&](int c) {   // corners of channels c .. c+3 (this lane's k group of the step)
#pragma unroll
    for (int i = 0; i < A_LD; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        cv[i][q] = goff[i][q] >= 0 ? *(const f32x4*)(x + (long)goff[i][q] + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  auto blend_store = [&](char* stage) {
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float val = gwt[i][0] * cv[i][0][j] + gwt[i][1] * cv[i][1][j] + gwt[i][2] * cv[i][2][j] + gwt[i][3] * cv[i][3][j];
        v[j] = val * gmask[i];
      }
      *(f32x4*)(stage + i * 4096 + tid * 16) = v;
    }
  };
  auto issue_w = [&](int kt, char* stage) {
#pragma unroll
    for (int j = 0; j < B_LD; ++j) dma16(b_ok[j] ? w + b_off[j] + kt * KS : zero, stage + wave * 1024 + j * 4096);
  };

  f32x4 acc[TP][TC];
#pragma unroll
  for (int p = 0; p < TP; ++p)
#pragma unroll
    for (int c = 0; c < TC; ++c) acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15;
  const int frag_off = fr * 64 + (((lane >> 4) ^ swz(fr)) << 4);
  const int nk = a.K / KS;          // K = R*S*Cin, Cin % 16 == 0
  const int spt = a.Cin / KS;       // K steps per tap

  int tap = 0, cstep = 0;           // tap / channel step of the K step whose corners are gathered next
  geometry(0);
  gather(g * EPV);
  issue_w(0, smW);
  blend_store(smA);
  for (int kt = 0; kt < nk; ++kt) {
    wait_vmcnt<0>();
    __syncthreads();
    const bool more = kt + 1 < nk;
    if (more) {
      issue_w(kt + 1, smW + ((kt + 1) & 1) * WSTAGE);
      if (++cstep == spt) { cstep = 0; ++tap; geometry(tap); }
      gather(cstep * KS + g * EPV);
    }
    const char* bA = smA + (kt & 1) * ASTAGE;
    const char* bW = smW + (kt & 1) * WSTAGE;
    f32x4 wf[TC], pf[TP];
#pragma unroll
    for (int c = 0; c < TC; ++c) wf[c] = *(const f32x4*)(bW + (wc * 16 * TC + 16 * c) * 64 + frag_off);
#pragma unroll
    for (int p = 0; p < TP; ++p) pf[p] = *(const f32x4*)(bA + (wp * 16 * TP + 16 * p) * 64 + frag_off);
    if constexpr (SP) {
#pragma unroll
      for (int p = 0; p < TP; ++p) mma_px<true, TC>(wf, pf[p], acc[p]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int p = 0; p < TP; ++p)
#pragma unroll
          for (int c = 0; c < TC; ++c)
            acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[c][e], pf[p][e], acc[p][c], 0, 0, 0);
    }
    if (more) blend_store(smA + ((kt + 1) & 1) * ASTAGE);
  }

  const int q = lane >> 4;
  const int cb = n0 + wc * 16 * TC;
#pragma unroll
  for (int p = 0; p < TP; ++p) {
    const int m = m0 + wp * 16 * TP + 16 * p + fr;
    if (m >= a.M) continue;
    epilogue_tiles<float, TC>(a, m, cb, q, acc[p]);
  }
}

// ------------------------------------------------------------------------------------------
// DCNv2 3x3 / stride 1 / pad 1 in f32 from an LDS window (maps divisible by 8x16, Cin % 16 == 0): the gather form above
// pulls four 64-byte corner pieces per (pixel, tap, 16 channels) through L2 -- 9.7 GB for one 64->64 @128^2 layer at batch
// 64, 9 TB/s, twice the time of its MFMAs.  Here (the f16 dcn_window_kernel's structure in f32) a workgroup owns an 8x16
// tile; per 16-channel chunk the 18x26 window (+-4 px margin for the learned offsets; 64 bytes per pixel, the same image as
// the f16 kernel's 32-channel chunk) is brought into LDS ONCE and all nine taps sample it: lane (pixel, 4-channel group)
// reads its 4 corners with ds_read_b128, blends in f32 in the reference's operation order
// (val = w1 v1 + w2 v2 + w3 v3 + w4 v4; val * mask, kernel.cu:697, :862) and the four results ARE its B operands of the next
// four 16x16x4 MFMAs -- the sampled tile never goes through LDS.  Geometry ({lh, lw, mask, window offset}) once per
// (pixel, tap), staged in LDS.  Samples outside the window are gathered from global memory by the lanes concerned.
// Weights stay tap-major in memory (k = tap*Cin + c): the K loop runs chunk-major and fetches the 64-byte piece it needs.
// ------------------------------------------------------------------------------------------
template <int BC, bool SP>
__global__ void __launch_bounds__(256, 2) dcn_f32_window_kernel(const ConvArgs a) {
  constexpr int TH = 8, TW = 16, BP = 128, MG = 4;
  constexpr int WR = TH + 2 + 2 * MG, WCOLS = TW + 2 + 2 * MG;  // 18 x 26 window pixels
  constexpr int NPIECE = WR * WCOLS * 4;                        // 1872 16-byte pieces
  constexpr int W_LD = (NPIECE + 255) / 256;                    // 8 DMA rounds; the last one only on waves 0-1
  constexpr int WINB = ((NPIECE + 63) / 64) * 1024;             // 30720
  constexpr int GEOB = 9 * BP * 16, GEOC = 9 * BP * 4;          // {lh, lw, mask, code} and the image coordinates of far samples
  constexpr int TP = 2, TC = BC / 16;                           // wave = 32 pixels (2 tile rows) x all BC couts
  constexpr int B_LD = BC / 64, WST = BC * 64, NST = 2, STG = 3 * WST;   // a stage = the three taps of a kernel row
  static_assert(BC % 64 == 0 && WINB + GEOB + GEOC + NST * STG <= 81920, "two workgroups per CU");
  __shared__ __attribute__((aligned(16))) char smem[WINB + GEOB + GEOC + NST * STG];
  char* const win = smem;
  char* const geo = smem + WINB;
  char* const geoc = smem + WINB + GEOB;
  char* const ring = smem + WINB + GEOB + GEOC;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  const int wy0 = ty0 - 1 - MG, wx0 = tx0 - 1 - MG;  // image coordinates of window pixel (0,0)
  const int nch = a.Cin / 16, nk = nch * 9;

  // ---- loaders
  const int lrow = tid >> 2, slotw = tid & 3;
  const int gwk = slotw ^ swz(lrow);
  const float* wptr[B_LD];
#pragma unroll
  for (int j = 0; j < B_LD; ++j) {
    const int L = lrow + 64 * j;
    const int tt = L >> 4, r = L & 15;
    wptr[j] = (const float*)a.w + (long)(n0 + cout_of<TC>(tt, r >> 2, r & 3)) * a.Kpad + gwk * 4;
  }
  auto issue_w = [&](int chunk, int tr, int st) {      // step (chunk, kernel row tr): 3 taps x 16 channels of every cout
#pragma unroll
    for (int ts = 0; ts < 3; ++ts)
#pragma unroll
      for (int j = 0; j < B_LD; ++j)
        dma16(wptr[j] + (tr * 3 + ts) * a.Cin + chunk * 16, ring + st * STG + ts * WST + wave * 1024 + j * 4096);
  };
  int wofs[W_LD];
#pragma unroll
  for (int i = 0; i < W_LD; ++i) {
    const int pid = tid + 256 * i;
    const int pw = pid >> 2, sl = pid & 3;
    const int wr = pw / WCOLS, wcn = pw - wr * WCOLS;
    const int y = wy0 + wr, x = wx0 + wcn;
    const bool ok = pid < NPIECE && y >= 0 && y < a.H && x >= 0 && x < a.W;
    // slot sl of a pixel in an odd window row holds channel group sl^2 (conflict-free corner reads over two rows)
    wofs[i] = ok ? (y * a.W + x) * a.in_stride + (sl ^ (2 * (wr & 1))) * 4 : -1;
  }
  auto issue_window = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < W_LD; ++i) {
      if (i < W_LD - 1 || (wave * 64 + 256 * i) * 16 < WINB)  // wave-uniform: the last round ends with the window
        dma16(wofs[i] >= 0 ? ximg + wofs[i] + chunk * 16 : zero, win + (wave * 64 + 256 * i) * 16);
    }
  };
  issue_window(0);
  issue_w(0, 0, 0);

  // ---- sampling geometry, once per (pixel, tap): thread = pixel gp, taps gh, gh+2, ...
  {
    const int gp = tid & 127, gh = tid >> 7;
    const int py = ty0 + (gp >> 4), pxx = tx0 + (gp & 15);
    const float* omrow = a.om + ((long)(b * a.H + py) * a.W + pxx) * a.om_stride;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int t = 2 * i + gh;
      if (t < 9) {
        const int tr = t / 3, ts = t - 3 * tr;
        const float h_im = (float)(py - 1 + tr) + omrow[2 * t], w_im = (float)(pxx - 1 + ts) + omrow[2 * t + 1];
        const float mraw = omrow[18 + t];
        const bool valid = h_im > -1.f && w_im > -1.f && h_im < (float)a.H && w_im < (float)a.W;     // kernel.cu:852
        const float mk = valid ? (a.mask_is_prob ? mraw : ctdet_sigmoid_exact(mraw)) : 0.f;           // invalid: contributes 0
        const float fh = floorf(h_im), fw = floorf(w_im);
        const int h_low = (int)fh, w_low = (int)fw;
        const int wr = h_low - wy0, wcn = w_low - wx0;  // window coordinates of corner (h_low, w_low)
        // every in-image corner of a valid sample inside the window => the out-of-image ones are too (1 px outside the
        // image, where the window is zero-filled): the window read then IS the guarded read of kernel.cu:683-693
        const bool inside = wr >= 0 && wr + 1 < WR && wcn >= 0 && wcn + 1 < WCOLS;
        const bool oow = valid && !inside;
        const unsigned code = (valid && inside) ? ((unsigned)((wr * WCOLS + wcn) * 64) | ((unsigned)(wr & 1) << 16)) : (oow ? 0x80000000u : 0u);
        f32x4 gv;
        gv[0] = valid ? h_im - fh : 0.f; gv[1] = valid ? w_im - fw : 0.f; gv[2] = mk; gv[3] = __uint_as_float(code);
        *(f32x4*)(geo + (t * BP + gp) * 16) = gv;
        *(unsigned*)(geoc + (t * BP + gp) * 4) = oow ? (((unsigned)(h_low + 1) << 12) | (unsigned)(w_low + 1)) : 0u;
      }
    }
  }
  wait_vmcnt<0>();
  __syncthreads();
  // ---- consumer mapping: lane (fr, q) samples tile pixel (row 2*wave + fr/8, col 8p + fr%8), channels 4q..4q+3 of the chunk
  const int fr = lane & 15, q = lane >> 4;
  const int prow = 2 * wave + (fr >> 3), pcol = fr & 7;

  // sampled * mask for tap t of the current chunk, both pixel tiles of this lane
  auto sample = [&](int t, int chunk, f32x4 (&pf)[TP]) {
    f32x4 g[TP];
    unsigned far = 0;
#pragma unroll
    for (int p = 0; p < TP; ++p) {
      g[p] = *(const f32x4*)(geo + (t * BP + prow * 16 + 8 * p + pcol) * 16);
      far |= __float_as_uint(g[p][3]) >> 31;
    }
    const bool any_far = __builtin_amdgcn_ballot_w64(far != 0) != 0;      // wave-uniform
#pragma unroll
    for (int p = 0; p < TP; ++p) {
      const unsigned code = __float_as_uint(g[p][3]);
      const unsigned ofs = (code & 0xFFFFu) + ((q ^ (2 * ((code >> 16) & 1u))) << 4);
      const char* c0p = win + ofs;
      const char* c2p = win + (ofs ^ 32u) + WCOLS * 64;       // next window row: the other slot swizzle
      f32x4 v1 = *(const f32x4*)(c0p), v2 = *(const f32x4*)(c0p + 64);
      f32x4 v3 = *(const f32x4*)(c2p), v4 = *(const f32x4*)(c2p + 64);
      if (any_far) {
        const bool out = code >> 31;
        const unsigned cc = *(const unsigned*)(geoc + (t * BP + prow * 16 + 8 * p + pcol) * 4);
        const int h_low = (int)((cc >> 12) & 0xFFFFFu) - 1, w_low = (int)(cc & 0xFFFu) - 1;
        const bool r0 = out && h_low >= 0, r1 = out && h_low + 1 <= a.H - 1, c0 = w_low >= 0, c1 = w_low + 1 <= a.W - 1;
        const long o0 = ((long)h_low * a.W + w_low) * a.in_stride, o2 = o0 + (long)a.W * a.in_stride;
        const float* base = ximg + chunk * 16 + q * 4;
        if (out) {      // asm loads (gload4_sync_into): no VMEM state for the compiler to wait on in the fast path; corners
          gload4_sync_into((r0 && c0) ? base + o0 : zero, v1);                    // outside the image read the zero page
          gload4_sync_into((r0 && c1) ? base + o0 + a.in_stride : zero, v2);
          gload4_sync_into((r1 && c0) ? base + o2 : zero, v3);
          gload4_sync_into((r1 && c1) ? base + o2 + a.in_stride : zero, v4);
        }
      }
      const float lh = g[p][0], lw = g[p][1], mk = g[p][2];
      const float hh = 1.f - lh, hw = 1.f - lw;
      const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float val = w1 * v1[e] + w2 * v2[e] + w3 * v3[e] + w4 * v4[e];
        pf[p][e] = val * mk;
      }
    }
  };

  f32x4 acc[TP][TC];
#pragma unroll
  for (int p = 0; p < TP; ++p)
#pragma unroll
    for (int c = 0; c < TC; ++c) acc[p][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const char* fragB = ring + fr * 64 + ((q ^ swz(fr)) << 4);

  // One barrier per kernel row (96 MFMAs per wave): its three weight taps are one ring stage, fetched a row ahead.  Inside
  // the row the operands of tap t+1 (weight fragments, sampled pixels) are read into registers during the MFMAs of tap t.
  auto frags = [&](int st, int ts, f32x4 (&wf)[TC]) {
#pragma unroll
    for (int c = 0; c < TC; ++c) wf[c] = *(const f32x4*)(fragB + st * STG + ts * WST + c * 1024);
  };
  f32x4 pf[TP], wf[TC];
  sample(0, 0, pf);
  const int ns = nch * 3;

  auto tap = [&](int s, int chunk, auto tc) {
    constexpr int T = decltype(tc)::value, TS = T % 3;
    const int st = s & 1;
    if (TS == 0) {
      wait_vmcnt<0>();                                             // this row's weights (issued a row ago)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (s + 1 < ns) issue_w(T == 6 ? chunk + 1 : chunk, T == 6 ? 0 : T / 3 + 1, st ^ 1);
      frags(st, 0, wf);
    }
    f32x4 wfn[TC], pfn[TP];
    if (TS < 2) frags(st, TS + 1, wfn);
    if (T < 8) {
      sample(T + 1, chunk, pfn);
    } else if (chunk + 1 < nch) {                                  // tap 8 was sampled during tap 7: the window is free
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue_window(chunk + 1);
    }
    if constexpr (SP) {
#pragma unroll
      for (int p = 0; p < TP; ++p) mma_px<true, TC>(wf, pf[p], acc[p]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int p = 0; p < TP; ++p)
#pragma unroll
          for (int c = 0; c < TC; ++c)
            acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[c][e], pf[p][e], acc[p][c], 0, 0, 0);
    }
    if (T == 8 && chunk + 1 < nch) {
      wait_vmcnt<0>();                                             // next chunk's window, behind this tap's MFMAs
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      sample(0, chunk + 1, pfn);
    }
#pragma unroll
    for (int p = 0; p < TP; ++p) pf[p] = pfn[p];
    if (TS < 2) {
#pragma unroll
      for (int c = 0; c < TC; ++c) wf[c] = wfn[c];
    }
  };
  for (int chunk = 0; chunk < nch; ++chunk) {
    const int s = chunk * 3;
    tap(s, chunk, std::integral_constant<int, 0>{});
    tap(s, chunk, std::integral_constant<int, 1>{});
    tap(s, chunk, std::integral_constant<int, 2>{});
    tap(s + 1, chunk, std::integral_constant<int, 3>{});
    tap(s + 1, chunk, std::integral_constant<int, 4>{});
    tap(s + 1, chunk, std::integral_constant<int, 5>{});
    tap(s + 2, chunk, std::integral_constant<int, 6>{});
    tap(s + 2, chunk, std::integral_constant<int, 7>{});
    tap(s + 2, chunk, std::integral_constant<int, 8>{});
  }

#pragma unroll
  for (int p = 0; p < TP; ++p) {
    const int m = (b * a.H + ty0 + prow) * a.W + tx0 + 8 * p + pcol;
    epilogue_tiles<float, TC>(a, m, n0, q, acc[p]);
  }
}

// ------------------------------------------------------------------------------------------
// f16x3 mode of the LDS-window DCNv2 above, re-cut for instruction count: the SP instantiation of dcn_f32_window_kernel
// issues 58 VALU per (pixel tile, tap) -- 4,181 per wave on a 64->64 tile against 576 MFMAs -- and is bound by vector issue
// (SQ counters: vector ALU 267 us of issue per SIMD in a 507 us launch, matrix pipe 147 us, LDS array 50 % busy).  Here
//   * the geometry stage stores the four bilinear weights ALREADY multiplied by the mask (f32: the split keeps them exact
//     to 2^-22) and the window byte offset with the row-parity swizzle folded in; the consumer's address is one v_xor with
//     its channel-group bits, the blend 16 FMAs per 4 channels (was: 6 for the weights + 20 + 9 of address arithmetic);
//   * far samples (outside the +-4 px window) re-derive their coordinates from the offsets in the wave-uniform slow path
//     instead of carrying a second LDS table;
//   * the split of a sampled fragment feeds the MFMAs directly: b1 = {hi, 0}, b2 = {lo, hi} (conv_common.h).
// Same tile (8x16 pixels x 64 couts), window (18x26 pixels, 16-channel chunks of f32), ring (a stage = the three taps of a
// kernel row) and barriers as dcn_f32_window_kernel.
// ------------------------------------------------------------------------------------------
// COLS: the instantiation that also writes the sampled columns (training forward); a template parameter so that the inference
// kernel's register allocation does not carry the pointer and index (with it the <2,64> form spilled 14 registers: +8 % time)
template <int TP, int BC, bool COLS = false>
__global__ void __launch_bounds__(512 / TP, (TP == 1 && BC == 64) ? 4 : 2) dcn_split_window_kernel(const ConvArgs a) {
  constexpr int TH = 8, TW = 16, BP = 128, MG = 4;
  constexpr int NT = 512 / TP;                                  // TP = 2: 4 waves of 32 pixels, TP = 1: 8 waves of 16
  static_assert(BC * 4 <= NT, "at most one weight piece per thread and tap (waves beyond BC / 16 fetch none)");
  constexpr int WR = TH + 2 + 2 * MG, WCOLS = TW + 2 + 2 * MG;  // 18 x 26 window pixels
  constexpr int NPIECE = WR * WCOLS * 4;                        // 1872 16-byte pieces
  constexpr int W_LD = (NPIECE + NT - 1) / NT;                  // DMA rounds; the last one ends with the window
  constexpr int WINB = ((NPIECE + 63) / 64) * 1024;             // 30720
  constexpr int GEOW = 9 * BP * 16, GEOC = 9 * BP * 4;          // {w1 m, w2 m, w3 m, w4 m} and the window code per (tap, pixel)
  constexpr int TC = BC / 16;                                   // wave = 16 * TP pixels x all BC couts
  constexpr int WST = BC * 64, NST = 2, STG = 3 * WST;          // a stage = the three taps of a kernel row
  static_assert(WINB + GEOW + GEOC + NST * STG <= ((TP == 2 || BC == 64) ? 81920 : 163840), "two workgroups / one (128 couts) per CU");
  __shared__ __attribute__((aligned(16))) char smem[WINB + GEOW + GEOC + NST * STG];
  char* const win = smem;
  char* const geow = smem + WINB;
  char* const geoc = smem + WINB + GEOW;
  char* const ring = smem + WINB + GEOW + GEOC;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  const int wy0 = ty0 - 1 - MG, wx0 = tx0 - 1 - MG;  // image coordinates of window pixel (0,0)
  const int nch = a.Cin / 16;

  // ---- loaders (as dcn_f32_window_kernel)
  const int lrow = tid >> 2, slotw = tid & 3;
  const int gwk = slotw ^ swz(lrow);
  const float* wptr;
  {
    const int tt = lrow >> 4, r = lrow & 15;
    wptr = (const float*)a.w + (long)(n0 + cout_of<TC>(tt, r >> 2, r & 3)) * a.Kpad + gwk * 4;
  }
  auto issue_w = [&](int chunk, int tr, int st) {      // step (chunk, kernel row tr): 3 taps x 16 channels of every cout
    if (BC * 4 < NT && wave >= BC / 16) return;        // (eight waves on a 64-cout tile: the first four carry the weights)
#pragma unroll
    for (int ts = 0; ts < 3; ++ts)
      dma16(wptr + (tr * 3 + ts) * a.Cin + chunk * 16, ring + st * STG + ts * WST + wave * 1024);
  };
  int wofs[W_LD];
#pragma unroll
  for (int i = 0; i < W_LD; ++i) {
    const int pid = tid + NT * i;
    const int pw = pid >> 2, sl = pid & 3;
    const int wr = pw / WCOLS, wcn = pw - wr * WCOLS;
    const int y = wy0 + wr, x = wx0 + wcn;
    const bool ok = pid < NPIECE && y >= 0 && y < a.H && x >= 0 && x < a.W;
    // slot sl of a pixel in an odd window row holds channel group sl^2 (conflict-free corner reads over two rows)
    wofs[i] = ok ? (y * a.W + x) * a.in_stride + (sl ^ (2 * (wr & 1))) * 4 : -1;
  }
  auto issue_window = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < W_LD; ++i) {
      if (i < W_LD - 1 || (wave * 64 + NT * i) * 16 < WINB) { // wave-uniform: the last round ends with the window
        int o = wofs[i];
        asm volatile("" : "+v"(o));     // the 64-bit address is formed here, from the 32-bit offset, not kept across the K loop
        dma16(o >= 0 ? ximg + o + chunk * 16 : zero, win + (wave * 64 + NT * i) * 16);
      }
    }
  };
  issue_window(0);
  issue_w(0, 0, 0);

  // ---- sampling geometry, once per (pixel, tap): thread = pixel gp, taps gh, gh + NT/128, ...
  {
    constexpr int GS = NT / 128;
    const int gp = tid & 127, gh = tid >> 7;
    const int py = ty0 + (gp >> 4), pxx = tx0 + (gp & 15);
    const float* omrow = a.om + ((long)(b * a.H + py) * a.W + pxx) * a.om_stride;
#pragma unroll
    for (int i = 0; i < (9 + GS - 1) / GS; ++i) {
      const int t = GS * i + gh;
      if (t < 9) {
        const int tr = t / 3, ts = t - 3 * tr;
        const float h_im = (float)(py - 1 + tr) + omrow[2 * t], w_im = (float)(pxx - 1 + ts) + omrow[2 * t + 1];
        const float mraw = omrow[18 + t];
        const bool valid = h_im > -1.f && w_im > -1.f && h_im < (float)a.H && w_im < (float)a.W;     // kernel.cu:852
        const float mk = valid ? (a.mask_is_prob ? mraw : ctdet_sigmoid_exact(mraw)) : 0.f;           // invalid: contributes 0
        const float fh = floorf(h_im), fw = floorf(w_im);
        const int h_low = (int)fh, w_low = (int)fw;
        const float lh = valid ? h_im - fh : 0.f, lw = valid ? w_im - fw : 0.f, hh = 1.f - lh, hw = 1.f - lw;
        const int wr = h_low - wy0, wcn = w_low - wx0;  // window coordinates of corner (h_low, w_low)
        // every in-image corner of a valid sample inside the window => the out-of-image ones are too (1 px outside the
        // image, where the window is zero-filled): the window read then IS the guarded read of kernel.cu:683-693
        const bool inside = wr >= 0 && wr + 1 < WR && wcn >= 0 && wcn + 1 < WCOLS;
        const bool oow = valid && !inside;
        // window byte offset of corner 1 for channel group 0, the row parity in bit 5 (group g sits at (g << 4) ^ (parity << 5))
        const unsigned code = (valid && inside) ? ((unsigned)((wr * WCOLS + wcn) * 64) | ((unsigned)(wr & 1) << 5)) : (oow ? 0x80000000u : 0u);
        f32x4 gv;
        gv[0] = hh * hw * mk; gv[1] = hh * lw * mk; gv[2] = lh * hw * mk; gv[3] = lh * lw * mk;
        *(f32x4*)(geow + (t * BP + gp) * 16) = gv;
        *(unsigned*)(geoc + (t * BP + gp) * 4) = code;
      }
    }
  }
  wait_vmcnt<0>();
  __syncthreads();
  // ---- consumer mapping: lane (fr, q) samples channels 4q..4q+3 of the chunk for tile pixel (row 2*wave + fr/8, col 8p +
  // fr%8) [TP = 2] or (row wave, col fr) [TP = 1]
  const int fr = lane & 15, q = lane >> 4;
  const int prow = TP == 2 ? 2 * wave + (fr >> 3) : wave, pcol = TP == 2 ? (fr & 7) : fr;
  const unsigned q4 = (unsigned)q << 4;

  // sampling of tap t in two halves, so that the MFMAs of the previous tap can be issued between them: `gather` starts the
  // LDS reads (weights, window code, the four corner fragments of both pixel tiles), `blend` turns them into split operands
  struct Raw { f32x4 w[TP], v[TP][4]; };
  auto gather = [&](int t, int chunk, Raw& r) {
    unsigned code[TP];
    unsigned far = 0;
#pragma unroll
    for (int p = 0; p < TP; ++p) {
      const int px = prow * 16 + (TP == 2 ? 8 * p : 0) + pcol;
      r.w[p] = *(const f32x4*)(geow + (t * BP + px) * 16);
      code[p] = *(const unsigned*)(geoc + (t * BP + px) * 4);
      far |= code[p] >> 31;
    }
    const bool any_far = __builtin_amdgcn_ballot_w64(far != 0) != 0;      // wave-uniform
#pragma unroll
    for (int p = 0; p < TP; ++p) {
      const unsigned A = (code[p] & 0xFFFFu) ^ q4;
      const char* c0p = win + A;
      const char* c2p = win + (A ^ 32u) + WCOLS * 64;       // next window row: the other slot swizzle
      r.v[p][0] = *(const f32x4*)(c0p); r.v[p][1] = *(const f32x4*)(c0p + 64);
      r.v[p][2] = *(const f32x4*)(c2p); r.v[p][3] = *(const f32x4*)(c2p + 64);
      if (any_far) {                                          // rare: the lanes concerned re-derive the sample from the offsets
        const bool out = code[p] >> 31;
        const int py = ty0 + prow, pxx = tx0 + (TP == 2 ? 8 * p : 0) + pcol, tr = t / 3, ts = t - 3 * (t / 3);
        const float* omrow = a.om + ((long)(b * a.H + py) * a.W + pxx) * a.om_stride;
        float dh, dw;
        gload2_sync(omrow + 2 * t, dh, dw);                  // (asm loads: see gload2_sync)
        const float h_im = (float)(py - 1 + tr) + dh, w_im = (float)(pxx - 1 + ts) + dw;
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const bool r0 = out && h_low >= 0, r1 = out && h_low + 1 <= a.H - 1, c0 = w_low >= 0, c1 = w_low + 1 <= a.W - 1;
        const long o0 = ((long)h_low * a.W + w_low) * a.in_stride, o2 = o0 + (long)a.W * a.in_stride;
        const float* base = ximg + chunk * 16 + q * 4;
        if (out) {                                            // corners outside the image read the zero page
          gload4_sync_into((r0 && c0) ? base + o0 : zero, r.v[p][0]);
          gload4_sync_into((r0 && c1) ? base + o0 + a.in_stride : zero, r.v[p][1]);
          gload4_sync_into((r1 && c0) ? base + o2 : zero, r.v[p][2]);
          gload4_sync_into((r1 && c1) ? base + o2 + a.in_stride : zero, r.v[p][3]);
        }
      }
    }
  };
  // training: the sampled value of (pixel, tap t, channels chunk*16 + 4q..) is also the backward pass's `columns` entry --
  // stored from here (the cout tile 0 workgroup of the pixel tile) it costs a 16-byte store per blend and saves the separate
  // sampling pass over the layer (ctdet_dcn_cols) in the backward
  float* const colp = (COLS && n0 == 0) ? a.cols_out + q * 4 : nullptr;
  auto blend = [&](const Raw& r, f16x8 (&b1)[TP], f16x8 (&b2)[TP], int t, int chunk) {
#pragma unroll
    for (int p = 0; p < TP; ++p) {
      f32x4 val;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        val[e] = r.w[p][0] * r.v[p][0][e] + r.w[p][1] * r.v[p][1][e] + r.w[p][2] * r.v[p][2][e] + r.w[p][3] * r.v[p][3][e];
      if constexpr (COLS) {
        if (colp) {
          const long m = (long)(b * a.H + ty0 + prow) * a.W + tx0 + (TP == 2 ? 8 * p : 0) + pcol;
          *(f32x4*)(colp + m * (9L * a.Cin) + t * a.Cin + chunk * 16) = val;
        }
      }
      split_b(val, b1[p], b2[p]);
    }
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
  f16x8 b1[TP], b2[TP], wf[TC];
  {
    Raw r0;
    gather(0, 0, r0);
    blend(r0, b1, b2, 0, 0);
  }
  const int ns = nch * 3;

  auto tap = [&](int s, int chunk, auto tc) {
    constexpr int T = decltype(tc)::value, TS = T % 3;
    const int st = s & 1;
    if (TS == 0) {
      wait_vmcnt<0>();                                             // this row's weights (issued a row ago)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (s + 1 < ns) issue_w(T == 6 ? chunk + 1 : chunk, T == 6 ? 0 : T / 3 + 1, st ^ 1);
    }
    Raw raw;
    frags(st, TS, wf);                                             // this tap's weight fragments
    if (T < 8) {
      gather(T + 1, chunk, raw);                                   // LDS reads in flight behind the MFMAs below
    } else if (chunk + 1 < nch) {                                  // tap 8 was sampled during tap 7: the window is free
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue_window(chunk + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < TP; ++p) {
#pragma unroll
      for (int c = 0; c < TC; ++c) acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c], b1[p], acc[p][c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < TC; ++c) acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c], b2[p], acc[p][c], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);                             // the operands above are consumed: b1 / b2 may be rewritten
    if (T < 8) {
      blend(raw, b1, b2, T + 1, chunk);
    } else if (chunk + 1 < nch) {
      wait_vmcnt<0>();                                             // next chunk's window, behind this tap's MFMAs
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      gather(0, chunk + 1, raw);
      blend(raw, b1, b2, 0, chunk + 1);
    }
  };
  for (int chunk = 0; chunk < nch; ++chunk) {
    const int s = chunk * 3;
    tap(s, chunk, std::integral_constant<int, 0>{});
    tap(s, chunk, std::integral_constant<int, 1>{});
    tap(s, chunk, std::integral_constant<int, 2>{});
    tap(s + 1, chunk, std::integral_constant<int, 3>{});
    tap(s + 1, chunk, std::integral_constant<int, 4>{});
    tap(s + 1, chunk, std::integral_constant<int, 5>{});
    tap(s + 2, chunk, std::integral_constant<int, 6>{});
    tap(s + 2, chunk, std::integral_constant<int, 7>{});
    tap(s + 2, chunk, std::integral_constant<int, 8>{});
  }

#pragma unroll
  for (int p = 0; p < TP; ++p) {
    const int m = (b * a.H + ty0 + prow) * a.W + tx0 + (TP == 2 ? 8 * p : 0) + pcol;
    epilogue_tiles<float, TC>(a, m, n0, q, acc[p]);
  }
}

// ------------------------------------------------------------------------------------------
// LDS-window form for the narrow layers of the DLA base in f32 (7x7 stem on the 4- or 8-channel padded image, level0 3x3
// 16->16, level1 3x3 16->32 stride 2): with 8 or 16 input channels the im2col-on-the-fly kernels above fetch every
// input pixel R*S times through L2 (26 GB for the stem at batch 64: 8.2 ms for 79 GFLOP).  Here a workgroup owns a
// TH x TW output tile, brings the input window into LDS once, keeps ALL weights in registers and reads every MFMA
// B fragment (4 channels of 16 consecutive pixels) with one ds_read_b128 at a compile-time offset.
// Pixels must be contiguous in memory (in_stride == CIN); maps divisible by the tile.
// ------------------------------------------------------------------------------------------
template <int R, int CIN, int TC, int STRIDE, int TH, bool NOCHECK, bool SP>
__global__ void __launch_bounds__(256) conv_f32_win_kernel(const ConvArgs a) {
  constexpr int TW = STRIDE == 1 ? 64 : 32;
  constexpr int WH = (TH - 1) * STRIDE + R, WW = (TW - 1) * STRIDE + R;
  constexpr int PB = CIN * 4;                       // bytes per pixel
  constexpr int PPR = WW * PB / 16;                 // 16-byte pieces per window row
  constexpr int NP = WH * PPR, ROUNDS = (NP + 255) / 256;
  constexpr int K = R * R * CIN, NK = (K + 15) / 16;
  constexpr int GPT = CIN / 4;                      // 4-channel k-groups per tap
  constexpr int NT = TW / 16, RW = TH / 4;          // pixel tiles per tile row; tile rows per wave
  static_assert(TH % 4 == 0 && ROUNDS * 4096 <= 65536, "tile");
  __shared__ __attribute__((aligned(16))) char win[ROUNDS * 4096];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, q = lane >> 4;
  const int tiles_x = a.Wo / TW, tiles_y = a.Ho / TH;
  const int tx0 = (blockIdx.x % tiles_x) * TW;
  const int ty0 = ((blockIdx.x / tiles_x) % tiles_y) * TH;
  const int b = blockIdx.x / (tiles_x * tiles_y);
  const float* zero = (const float*)g_zero_page;
  asm volatile("" : "+v"(zero));
  const float* ximg = (const float*)a.x + (long)b * a.H * a.W * CIN;
  const int iy0 = ty0 * STRIDE - a.pad, ix0 = tx0 * STRIDE - a.pad;   // image coordinates of window pixel (0,0)

#pragma unroll
  for (int i = 0; i < ROUNDS; ++i) {
    const int pid = tid + 256 * i;
    const int wr = pid / PPR, cp = pid - wr * PPR;
    const int y = iy0 + wr, x = ix0 + cp / (PB / 16);
    bool ok = pid < NP;
    if constexpr (!NOCHECK) ok = ok && y >= 0 && y < a.H && x >= 0 && x < a.W;
    dma16(ok ? ximg + ((long)y * a.W + ix0) * CIN + cp * 4 : zero, win + (wave * 64 + 256 * i) * 16);
  }
  // weights: fragment (kt, c) of this lane = 4 k of cout row cout_of(c, fr/4, fr%4), straight into registers
  f32x4 wf[NK][TC];
#pragma unroll
  for (int c = 0; c < TC; ++c) {
    const float* wr = (const float*)a.w + (long)cout_of<TC>(c, fr >> 2, fr & 3) * a.Kpad + q * 4;
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) wf[kt][c] = *(const f32x4*)(wr + kt * 16);
  }
  // SP: with 16 or 32 couts a split in registers per fragment (10 VALU for 2 * TC MFMAs) makes the kernel vector-issue bound.
  // Instead the window is split ONCE, in place (every thread the pieces its own DMAs fetched: no barrier of its own), each
  // 16-byte piece becoming {hi[4], lo[4]}; a fragment read then IS the B operand of both products, with the weights held as
  //   wa = {w_lo, w_hi} -> wa . {hi, lo} = w_lo hi + w_hi lo      wb = {w_hi, 0} -> wb . {hi, lo} = w_hi hi
  // (the packed group is {w_hi, w_lo}): two MFMAs per tile and K step, no VALU in the loop.
  f16x8 wa[SP ? NK : 1][TC], wb[SP ? NK : 1][TC];
  if constexpr (SP) {
    const f16x4 z4 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
#pragma unroll
    for (int kt = 0; kt < NK; ++kt)
#pragma unroll
      for (int c = 0; c < TC; ++c) {
        const f16x8 raw = __builtin_bit_cast(f16x8, wf[kt][c]);
        wa[kt][c] = __builtin_shufflevector(raw, raw, 4, 5, 6, 7, 0, 1, 2, 3);
        wb[kt][c] = __builtin_shufflevector(__builtin_shufflevector(raw, raw, 0, 1, 2, 3), z4, 0, 1, 2, 3, 4, 5, 6, 7);
      }
  }
  // per K step: LDS byte address of this lane's k-group for tile pixel (row 0 of the wave, column fr)
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
  if constexpr (SP) {
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
      char* pc = win + (tid + 256 * i) * 16;
      const f32x4 v = *(const f32x4*)pc;
      const f16x4 hi = __builtin_convertvector(v, f16x4);
      f32x4 r;
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = v[j] - (float)hi[j];
      const f16x4 lo = __builtin_convertvector(r, f16x4);
      *(f16x8*)pc = __builtin_shufflevector(hi, lo, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  }
  __syncthreads();

  // two pixel tiles at a time: two independent accumulator chains per cout tile (a 16x16x4 f32 MFMA has 40 cycles of
  // dependent latency against 32 of issue)
  static_assert(NT % 2 == 0, "pixel tiles come in pairs");
#pragma unroll
  for (int rr = 0; rr < RW; ++rr) {
#pragma unroll
    for (int tc = 0; tc < NT; tc += 2) {
      const int toff = (rr * STRIDE * WW + tc * 16 * STRIDE) * PB;   // compile time after unrolling
      f32x4 acc[2][TC];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int c = 0; c < TC; ++c) acc[u][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < NK; ++kt) {
        const f32x4 pf0 = *(const f32x4*)(win + kaddr[kt] + toff);
        const f32x4 pf1 = *(const f32x4*)(win + kaddr[kt] + toff + 16 * STRIDE * PB);
        if constexpr (SP) {
          const f16x8 h0 = __builtin_bit_cast(f16x8, pf0), h1 = __builtin_bit_cast(f16x8, pf1);
#pragma unroll
          for (int c = 0; c < TC; ++c) {
            acc[0][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[kt][c], h0, acc[0][c], 0, 0, 0);
            acc[1][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[kt][c], h1, acc[1][c], 0, 0, 0);
          }
#pragma unroll
          for (int c = 0; c < TC; ++c) {
            acc[0][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[kt][c], h0, acc[0][c], 0, 0, 0);
            acc[1][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[kt][c], h1, acc[1][c], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < TC; ++c) {
              acc[0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kt][c][e], pf0[e], acc[0][c], 0, 0, 0);
              acc[1][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kt][c][e], pf1[e], acc[1][c], 0, 0, 0);
            }
        }
      }
      const int m = (b * a.Ho + ty0 + wave * RW + rr) * a.Wo + tx0 + tc * 16 + fr;
      epilogue_tiles<float, TC>(a, m, 0, q, acc[0]);
      epilogue_tiles<float, TC>(a, m + 16, 0, q, acc[1]);
    }
  }
}

template <int R, int CIN, int TC, int STRIDE, int TH, bool SP>
static int launch_f32_win(const ConvArgs& a, hipStream_t s) {
  constexpr int TW = STRIDE == 1 ? 64 : 32;
  const int tiles = a.B * (a.Ho / TH) * (a.Wo / TW);
  // pad 0 (a pre-padded image: ops.preprocess border): the window never leaves the tensor -> no bounds checks
  if (a.pad == 0)
    hipLaunchKernelGGL((conv_f32_win_kernel<R, CIN, TC, STRIDE, TH, true, SP>), dim3(tiles), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv_f32_win_kernel<R, CIN, TC, STRIDE, TH, false, SP>), dim3(tiles), dim3(256), 0, s, a);
  CTDET_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// Direct form: one thread per (pixel, cout), f32 FMA chain in k order (any Cin / stride).
// ------------------------------------------------------------------------------------------
template <bool DEFORM, bool SP>
__global__ void __launch_bounds__(256) conv_direct_f32_kernel(const ConvArgs a) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int CP = (a.Cout + 3) & ~3;
  if (idx >= (long)a.M * CP) return;
  const int n = (int)(idx % CP);
  const int m = (int)(idx / CP);
  if (n >= a.Cout) return;
  const float* __restrict__ x = (const float*)a.x;
  const float* __restrict__ w = (const float*)a.w + (long)n * a.Kpad;
  const int wo = m % a.Wo, t = m / a.Wo;
  const int ho = t % a.Ho, b = t / a.Ho;
  const int pix_base = b * a.H * a.W;
  const int hb = ho * a.stride - a.pad, wb = wo * a.stride - a.pad;
  const int idl = a.in_dil;
  float acc = 0.f;
  for (int tr = 0; tr < a.R; ++tr)
    for (int ts = 0; ts < a.S; ++ts) {
      const int kb = (tr * a.S + ts) * a.Cin;   // k of channel 0 of this tap in the packed row
      if constexpr (!DEFORM) {
        const int hn = hb + tr * a.dil, wn = wb + ts * a.dil;
        if (hn < 0 || wn < 0 || hn % idl || wn % idl) continue;
        const int hi = hn / idl, wi = wn / idl;
        if (hi >= a.H || wi >= a.W) continue;
        if (a.nsrc > 1) {
          int c = 0;
          for (int j = 0; j < a.nsrc; ++j) {
            const float* xp = (const float*)a.xs[j] + (long)m * a.xs_stride[j] - c;
            for (; c < a.xs_cend[j]; ++c) acc = fmaf(xp[c], packed_w<SP>(w, kb + c), acc);
          }
          continue;
        }
        const float* xp = x + (long)(pix_base + hi * a.W + wi) * a.in_stride;
        for (int c = 0; c < a.Cin; ++c) acc = fmaf(xp[c], packed_w<SP>(w, kb + c), acc);
      } else {
        DcnSample sp;
        dcn_setup(a, true, pix_base, hb, wb, tr, ts, a.om + (long)m * a.om_stride, sp);
        for (int c = 0; c < a.Cin; ++c) {
          const float v1 = sp.off[0] >= 0 ? x[(long)sp.off[0] + c] : 0.f;
          const float v2 = sp.off[1] >= 0 ? x[(long)sp.off[1] + c] : 0.f;
          const float v3 = sp.off[2] >= 0 ? x[(long)sp.off[2] + c] : 0.f;
          const float v4 = sp.off[3] >= 0 ? x[(long)sp.off[3] + c] : 0.f;
          const float val = sp.wt[0] * v1 + sp.wt[1] * v2 + sp.wt[2] * v3 + sp.wt[3] * v4;
          acc = fmaf(val * sp.mask, packed_w<SP>(w, kb + c), acc);
        }
      }
    }
  float v = acc;
  if (a.scale) v *= a.scale[n];
  if (a.bias) v += a.bias[n];
  if (a.res) v += ((const float*)a.res)[(long)m * a.res_stride + n];
  if (a.act == CTDET_ACT_RELU) v = fmaxf(v, 0.f);
  else if (a.act == CTDET_ACT_SIGMOID_CLAMP) v = fminf(fmaxf(ctdet_sigmoid_exact(v), a.clamp_lo), a.clamp_hi);
  ((float*)a.y)[(long)m * a.out_stride + n] = v;
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
int launch_halo_split(const ConvArgs& a, hipStream_t s);   // conv_igemm.hip
int launch_halo_pair(const ConvArgs& a, hipStream_t s);    // conv_igemm.hip
int launch_halo_pair2(const ConvArgs& a, hipStream_t s);   // conv_igemm.hip

static bool aligned16(const void* p) { return (((size_t)p) & 15) == 0; }

static bool f32_vector_ok(const ConvArgs& a, int bc) {
  if (a.Cin % 4 || a.in_stride % 4 || a.Kpad % 16 || a.Kpad < a.K || a.Cout_pad % bc || a.Cout_pad < a.Cout) return false;
  if (a.Cout % 4 || a.out_stride % 4 || !aligned16(a.y) || !aligned16(a.w)) return false;
  if (a.res && (a.res_stride % 4 || !aligned16(a.res))) return false;
  if (a.R * a.S > 64) return false;
  if (a.nsrc > 1) {
    for (int j = 0; j < a.nsrc; ++j)
      if (a.xs_cend[j] % 4 || a.xs_stride[j] % 4 || !aligned16(a.xs[j])) return false;
    return true;
  }
  return aligned16(a.x);
}

template <int BP, int BC, int WP, int WC_, bool SP>
static int launch_f32_mfma(const ConvArgs& a, int kind, hipStream_t s) {   // kind: 0 generic, 1 uniform-K
  const int nbx = (a.M + BP - 1) / BP, nby = a.Cout_pad / BC;
  dim3 grid(8 * ((nbx + 7) / 8) * nby);
  if (kind == 1 && a.nsrc > 1)
    hipLaunchKernelGGL((conv_f32_uk_kernel<BP, BC, WP, WC_, true, SP>), grid, dim3(256), 0, s, a);
  else if (kind == 1)
    hipLaunchKernelGGL((conv_f32_uk_kernel<BP, BC, WP, WC_, false, SP>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv_f32_mfma_kernel<BP, BC, WP, WC_, SP>), grid, dim3(256), 0, s, a);
  CTDET_LAUNCH_CHECK();
  return 0;
}

template <int BP, int BC, int WP, int WC_, bool SP>
static int launch_f32_dcn(const ConvArgs& a, hipStream_t s) {
  const int nbx = (a.M + BP - 1) / BP, nby = a.Cout_pad / BC;
  dim3 grid(8 * ((nbx + 7) / 8) * nby);
  hipLaunchKernelGGL((dcn_f32_mfma_kernel<BP, BC, WP, WC_, SP>), grid, dim3(256), 0, s, a);
  CTDET_LAUNCH_CHECK();
  return 0;
}

static bool f32_uniform_k_ok(const ConvArgs& a) {
  if (a.R * a.S > 32 || a.in_dil != 1 || a.Kpad != a.K) return false;
  if (a.nsrc > 1) {
    int prev = 0;
    for (int j = 0; j < a.nsrc; ++j) {
      if ((a.xs_cend[j] - prev) % 16) return false;
      prev = a.xs_cend[j];
    }
    return true;
  }
  return a.Cin % 16 == 0;
}

template <bool SP>
static int launch_conv_f32_t(const ConvArgs& a, bool deform, hipStream_t s) {
  CTDET_CHECK((long)a.B * a.H * a.W * a.in_stride < (1L << 31), "conv: input too large for 32-bit element offsets");
  CTDET_CHECK(a.in_dil >= 1 && (!deform || a.in_dil == 1), "conv: bad in_dil %d", a.in_dil);
  CTDET_CHECK((a.Kpad >= a.K || (SP && a.korder >= 2)) && a.Cout_pad >= a.Cout,
              "conv(f32): packed weights [%d][%d] too small for Cout=%d K=%d", a.Cout_pad, a.Kpad, a.Cout, a.K);
  CTDET_CHECK(a.korder == 0 || (SP && (a.korder == 2 || a.korder == 3)), "conv(f32 / f16x3): korder %d", a.korder);
  const int bc = pick_bc(a.Cout);
  const bool vec = f32_vector_ok(a, bc) && (!deform || (a.Cin % 16 == 0 && a.nsrc <= 1));
  if (vec && !deform && (a.Cin == 4 || a.Cin == 8 || a.Cin == 16) && a.in_dil == 1 && a.nsrc <= 1 && a.R == a.S && a.dil == 1 &&
      a.in_stride == a.Cin && a.Cout_pad == bc && !a.res && a.Kpad == ((a.K + 15) & ~15)) {
    // the three narrow DLA base layers on tile-divisible maps
    if (a.R == 7 && a.Cin == 8 && bc == 16 && a.stride == 1 && a.Ho % 8 == 0 && a.Wo % 64 == 0 && (a.pad == 0 || a.pad == 3))
      return launch_f32_win<7, 8, 1, 1, 8, SP>(a, s);
    // the same stem on 4-channel pixels (3 used): K = 196 instead of 392, a 16-k step = four taps
    if (a.R == 7 && a.Cin == 4 && bc == 16 && a.stride == 1 && a.Ho % 8 == 0 && a.Wo % 64 == 0 && (a.pad == 0 || a.pad == 3))
      return launch_f32_win<7, 4, 1, 1, 8, SP>(a, s);
    if (a.R == 3 && a.Cin == 16 && bc == 16 && a.stride == 1 && a.Ho % 8 == 0 && a.Wo % 64 == 0 && a.pad == 1)
      return launch_f32_win<3, 16, 1, 1, 8, SP>(a, s);
    if (a.R == 3 && a.Cin == 16 && bc == 32 && a.stride == 2 && a.Ho % 4 == 0 && a.Wo % 32 == 0 && a.pad == 1)
      return launch_f32_win<3, 16, 2, 2, 4, SP>(a, s);
  }
  if constexpr (SP) {
    if (a.korder >= 2) {                  // pair-packed weights: only the halo pair kernels read them
      CTDET_CHECK(!deform && f32_vector_ok(a, bc), "conv(f16x3, pair weights): unaligned operands or deformable conv");
      return a.korder == 3 ? launch_halo_pair2(a, s) : launch_halo_pair(a, s);
    }
    if (vec && !deform) {                 // 3x3 / s1 / p1: the halo-resident kernel of conv_igemm.hip on f32 activations
      const int rc = launch_halo_split(a, s);
      if (rc <= 0) return rc;
    }
  }
  if (vec) {
    const bool big = ((long)a.M + 255) / 256 * (a.Cout_pad / bc) >= 512;
    if (deform && a.R == 3 && a.S == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 && a.H % 8 == 0 && a.W % 16 == 0 &&
        a.H <= 4094 && a.W <= 4094 && a.Kpad == a.K && a.Cout_pad % 64 == 0 && !(ctdet_tuning_flags() & CTDET_TUNE_NO_F32_DCN_WINDOW)) {
      // 64 couts per workgroup (128 would spill under two workgroups per CU); wider layers sample the window once per cout tile
      const int nbx = a.B * (a.H / 8) * (a.W / 16);
      dim3 grid(8 * ((nbx + 7) / 8) * (a.Cout_pad / 64));
      if (SP && !(ctdet_tuning_flags() & CTDET_TUNE_DCN_WINDOW_V1)) {
        // the sampling (conflict-laden LDS gathers, blend, split) is this kernel's larger half and is repeated for every cout
        // tile: the 128- and 256-cout layers use 128-cout tiles -- eight 16-pixel waves, one workgroup per CU, the same
        // eight waves per CU with half the gathers per MFMA (CTDET_TUNE_DCN_SPLIT_4W: 64-cout tiles everywhere)
        if (a.Cout_pad % 128 == 0 && !(ctdet_tuning_flags() & CTDET_TUNE_DCN_SPLIT_4W)) {
          dim3 grid128(8 * ((nbx + 7) / 8) * (a.Cout_pad / 128));
          if (a.cols_out) hipLaunchKernelGGL((dcn_split_window_kernel<1, 128, true>), grid128, dim3(512), 0, s, a);
          else hipLaunchKernelGGL((dcn_split_window_kernel<1, 128>), grid128, dim3(512), 0, s, a);
        } else {
          if (a.cols_out) hipLaunchKernelGGL((dcn_split_window_kernel<2, 64, true>), grid, dim3(256), 0, s, a);
          else if (ctdet_tuning_flags() & CTDET_TUNE_DCN_SPLIT_8W64)
            hipLaunchKernelGGL((dcn_split_window_kernel<1, 64>), grid, dim3(512), 0, s, a);
          else hipLaunchKernelGGL((dcn_split_window_kernel<2, 64>), grid, dim3(256), 0, s, a);
        }
      }
      else
        hipLaunchKernelGGL((dcn_f32_window_kernel<64, SP>), grid, dim3(256), 0, s, a);
      CTDET_LAUNCH_CHECK();
      return 0;
    }
    if (deform) {   // 128-pixel tiles: two or more workgroups per CU cover each other's gather latency
      switch (bc) {
        case 16: return launch_f32_dcn<128, 16, 4, 1, SP>(a, s);
        case 32: return launch_f32_dcn<128, 32, 4, 1, SP>(a, s);
        case 64: return launch_f32_dcn<64, 64, 2, 2, SP>(a, s);
        case 128: return launch_f32_dcn<128, 128, 2, 2, SP>(a, s);
      }
    }
    const int kind = f32_uniform_k_ok(a) ? 1 : 0;
    switch (bc) {
      case 16: return launch_f32_mfma<256, 16, 4, 1, SP>(a, kind, s);
      case 32: return big ? launch_f32_mfma<256, 32, 4, 1, SP>(a, kind, s) : launch_f32_mfma<128, 32, 4, 1, SP>(a, kind, s);
      case 64: return big ? launch_f32_mfma<256, 64, 4, 1, SP>(a, kind, s) : launch_f32_mfma<128, 64, 2, 2, SP>(a, kind, s);
      case 128: return big ? launch_f32_mfma<256, 128, 2, 2, SP>(a, kind, s) : launch_f32_mfma<128, 128, 2, 2, SP>(a, kind, s);
    }
  }
  const int CP = (a.Cout + 3) & ~3;
  const long total = (long)a.M * CP;
  dim3 grid((unsigned)((total + 255) / 256));
  if (deform)
    hipLaunchKernelGGL((conv_direct_f32_kernel<true, SP>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv_direct_f32_kernel<false, SP>), grid, dim3(256), 0, s, a);
  CTDET_LAUNCH_CHECK();
  return 0;
}

// does launch_conv_f32(deform, split) take the LDS-window kernel that can also write the columns (ConvArgs::cols_out)?
bool dcn_split_window_ok(const ConvArgs& a) {
  const int bc = pick_bc(a.Cout);
  return f32_vector_ok(a, bc) && a.Cin % 16 == 0 && a.nsrc <= 1 && a.R == 3 && a.S == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 &&
         a.H % 8 == 0 && a.W % 16 == 0 && a.H <= 4094 && a.W <= 4094 && a.Kpad == a.K && a.Cout_pad % 64 == 0 && a.korder == 0 &&
         !(ctdet_tuning_flags() & (CTDET_TUNE_NO_F32_DCN_WINDOW | CTDET_TUNE_DCN_WINDOW_V1));
}

int launch_conv_f32(const ConvArgs& a, bool deform, bool split, hipStream_t s) {
  return split ? launch_conv_f32_t<true>(a, deform, s) : launch_conv_f32_t<false>(a, deform, s);
}

// f32 packed weights [rows][Kpad] -> the split image of the same size: per group of 4 k {w_hi[4], w_lo[4]} (f16)
__global__ void __launch_bounds__(256) split_weights_kernel(const f32x4* __restrict__ src, f16x8* __restrict__ dst, long groups) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= groups) return;
  const f32x4 w = src[i];
  f16x8 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f16 h = (f16)w[j];
    o[j] = h;
    o[4 + j] = (f16)(w[j] - (float)h);
  }
  dst[i] = o;
}

int launch_split_weights(const float* src, void* dst, long n, hipStream_t s) {
  CTDET_CHECK(n % 4 == 0 && (((size_t)src | (size_t)dst) & 15) == 0, "split_weights: needs 16-byte aligned buffers of a multiple of 4 floats");
  if (n == 0) return 0;
  const long groups = n / 4;
  hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, (const f32x4*)src, (f16x8*)dst,
                     groups);
  CTDET_LAUNCH_CHECK();
  return 0;
}
