// Batched CenterNet decode: 3x3 max-pool peak test + exact top-K + box assembly.
// Reference: detectron2/modeling/meta_arch/centernet.py:399-405 (_nms, `hmax == heat` exact equality,
// plateaus all kept), :408-424 (_topk: per-class top-K then top-K of C*K == global top-K),
// :426-458 (ctdet_decode).  The reference handles batch==1 only and loops over images in Python
// (:224-233); here the whole batch is two launches with no host round trip.
//
// Layout: heat is f32 NHWC [B,H,W,heat_stride] (the layout the head conv writes; the first C channels count).
// Ordering contract ("canonical order"): score descending, ties by the reference's flat NCHW index
// canon = c*H*W + y*W + x ascending.  torch.topk's own tie order is unspecified, so ties are
// where a difference is permitted; fixtures are tie-free or assert the canonical rule.
//
// One pass over the heat map (dec_tile_kernel): a workgroup brings an 8x16-pixel tile (+1 pixel halo, all channels) into
// LDS once, zeroes everything that is not a positive 3x3 peak, and selects the tile's own K best peaks exactly -- MSD
// radix select on the key (score bits, ~canon), 4096 bins per level, the further levels only while the bin that holds
// the tile's K-th key is crowded (near-constant or tied maps: every level re-scans LDS, not memory).  A member of
// the image's top K is a member of its tile's top K, so the <= K-1 + DEC_TILE_SLACK survivors per tile form a small
// candidate list per image (tens of KB) in which dec_final_kernel (one workgroup per image) repeats the same select,
// sorts the <= K-1 + DEC_CAP finalists and assembles the boxes.  HBM traffic: the heat map once (halo rows come from
// L2) + ~1 % for the candidates; the first version read it two to three times in 15 launches.
//
// Clamp floor (DecArgs::floor_bits, ctdet_decode's heat_floor): the map `_sigmoid` hands over (centernet.py:13-15) is
// clamped to [1e-4, 1 - 1e-4], and a trained network's background sits exactly ON the lower clamp -- a plateau of several
// hundred thousand tied "peaks" per image that rank below every other peak and among themselves by flat index.  With the
// floor given, the tile pass leaves them out (its lists hold the few real peaks only) and dec_final_kernel, if an image
// has fewer than K peaks above the floor, takes the floor peaks of lowest flat index straight from the map -- the same
// entries in the same order the canonical rule selects.  A positive value below the floor breaks the caller's promise
// and is reported through the status word (ctdet_decode_status: -EINVAL).
#include "common.h"

#define DEC_CAP 2048            // max uncertain candidates carried to the final sort
#define DEC_NCAND 4096          // sort width (>= K-1 + DEC_CAP)
#define DEC_HIST 4096           // bins per level
#define DEC_LEVELS 6            // 1 + ceil(56 / 12)
#define DEC_TILE_SLACK 128      // a tile stops refining once its threshold bin holds at most this many keys
#define DEC_TW 16
#define DEC_LDS_TILE (60 * 1024)
// per-image workspace (uint32 words): [0] candidate count, [1] overflow flag, [16..) candidates (u64)
#define DEC_ST_WORDS 16
enum { ST_NCAND = 0, ST_OVERFLOW = 1, ST_BELOW_FLOOR = 2 };

__device__ __forceinline__ int dec_d0(uint32_t bits) {
  const int d = ((int)bits - 0x38000000) >> 15;
  return d < 0 ? 0 : (d > 4095 ? 4095 : d);
}
__device__ __forceinline__ uint64_t dec_key(uint32_t bits, uint32_t canon) {
  return ((uint64_t)bits << 24) | (uint64_t)(0xFFFFFFu - canon);     // 56 bits: larger key = better rank
}

// Exact "K-th largest key" search shared by both kernels: MSD radix select, 12-bit digits.
// Level 0 bins the score bits finely over the sigmoid range [2^-15, 2) (dec_d0: no pre-pass needed, resolves every map
// whose values are spread out).  While the bin that holds the K-th key has more than `slack` keys, further levels split
// it -- on the 12 key bits from the most significant position in which its keys DIFFER downwards (AND / OR of the keys: the
// common leading bits are skipped, so a plateau of equal scores -- the clamped background of a trained network -- is split
// by its index bits at once instead of walking through 15 identical score bits first).  All keys of the group agree above
// the window and the windows of successive levels descend, so "key restricted to the cared bits" orders the group like
// its digits do.
// The caller provides visit(f): calls f(bits, canon) for every key of its set (each thread its own share; the set must not
// change between calls).  Result (uniform across the workgroup): a key is selected iff  d0 > p0  or  (d0 == p0 and
// (key & care) >= want); the selected set holds `above` keys strictly above the threshold group plus the group's `inbin`
// keys, above < need <= above + inbin, and inbin <= slack unless the keys cannot be split further (unique keys: never).
struct DecSel { uint32_t p0; uint64_t care, want; uint32_t above, inbin; bool takeall; };

template <int NT, typename V>
__device__ __forceinline__ DecSel dec_select(V visit, uint32_t* lh, uint32_t need, uint32_t slack) {
  constexpr int NW = NT / 64;
  constexpr int per = DEC_HIST / NT;
  __shared__ uint32_t wtot[NW];
  __shared__ uint32_t sh_T, sh_above, sh_cnt;
  __shared__ unsigned long long sh_and, sh_or;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  DecSel r;
  r.p0 = 0; r.care = 0; r.want = 0; r.above = 0; r.inbin = 0; r.takeall = false;
  int dshift = 0;         // the current level's digit is key bits [dshift, dshift + 12) (level >= 1)
  for (int level = 0; level < 8; ++level) {
    for (int i = t; i < DEC_HIST; i += NT) lh[i] = 0;
    __syncthreads();
    // run-length accumulation: consecutive keys of a thread usually share a bin (heat maps cluster around
    // sigmoid(bias)); one LDS atomic per run removes most of the same-address serialisation
    uint32_t run_bin = 0xFFFFFFFFu, run_cnt = 0;
    auto count = [&](uint32_t bin) {
      if (bin == run_bin) { ++run_cnt; return; }
      if (run_cnt) atomicAdd(&lh[run_bin], run_cnt);
      run_bin = bin; run_cnt = 1;
    };
    visit([&](uint32_t bits, uint32_t canon) {
      const uint32_t d0 = (uint32_t)dec_d0(bits);
      if (level == 0) { count(d0); return; }
      if (d0 != r.p0) return;
      const uint64_t key = dec_key(bits, canon);
      if ((key & r.care) != r.want) return;
      const uint32_t dig = (uint32_t)(key >> dshift) & 0xFFFu;
      count(dig);
    });
    if (run_cnt) atomicAdd(&lh[run_bin], run_cnt);
    __syncthreads();
    // suffix sums over the threads' bin ranges (wave shuffles + one LDS step), then the owner of the crossing bin
    // publishes it
    uint32_t loc[per], s = 0;
#pragma unroll
    for (int j = 0; j < per; ++j) { loc[j] = lh[t * per + j]; s += loc[j]; }
    uint32_t suf = s;     // inclusive suffix sum within the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = __shfl_down(suf, o, 64);
      if (lane + o < 64) suf += v;
    }
    if (lane == 0) wtot[wv] = suf;
    __syncthreads();
    uint32_t higher = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { const uint32_t v = wtot[w]; total += v; if (w > wv) higher += v; }
    const uint32_t incl = suf + higher;          // keys in the bins of threads >= t
    const uint32_t want_n = need - r.above;
    if (total < want_n) {   // only possible at level 0: fewer keys than needed
      r.takeall = true;
      r.inbin = total;
      __syncthreads();
      return r;
    }
    const uint32_t above_me = incl - s;
    if (above_me < want_n && incl >= want_n) {
      uint32_t cum = above_me;
#pragma unroll
      for (int j = per - 1; j >= 0; --j) {
        if (cum + loc[j] >= want_n) { sh_T = t * per + j; sh_above = cum; sh_cnt = loc[j]; break; }
        cum += loc[j];
      }
    }
    if (t == 0) { sh_and = ~0ull; sh_or = 0ull; }
    __syncthreads();
    const uint32_t T = sh_T, cnt = sh_cnt;
    r.above += sh_above;
    r.inbin = cnt;
    if (level == 0) r.p0 = T;
    else {
      r.care |= 0xFFFull << dshift;
      r.want |= (uint64_t)T << dshift;
    }
    if (cnt <= slack) break;
    // the positions in which the keys of the threshold group differ: its 12 highest ones are the next digit
    uint64_t a_and = ~0ull, a_or = 0ull;
    visit([&](uint32_t bits, uint32_t canon) {
      if ((uint32_t)dec_d0(bits) != r.p0) return;
      const uint64_t key = dec_key(bits, canon);
      if ((key & r.care) != r.want) return;
      a_and &= key; a_or |= key;
    });
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      a_and &= __shfl_down((unsigned long long)a_and, o, 64);
      a_or |= __shfl_down((unsigned long long)a_or, o, 64);
    }
    if (lane == 0) { atomicAnd(&sh_and, (unsigned long long)a_and); atomicOr(&sh_or, (unsigned long long)a_or); }
    __syncthreads();
    uint64_t diff = (sh_and ^ sh_or) & ~r.care & ((1ull << 56) - 1);
    if (!diff) break;     // identical keys (impossible: canon is unique)
    const int hi = 63 - __builtin_clzll(diff);
    dshift = hi >= 11 ? hi - 11 : 0;
    __syncthreads();
  }
  return r;
}

__device__ __forceinline__ bool dec_selected(const DecSel& r, uint32_t bits, uint32_t canon) {
  if (r.takeall) return true;
  const uint32_t d0 = (uint32_t)dec_d0(bits);
  if (d0 != r.p0) return d0 > r.p0;
  return (dec_key(bits, canon) & r.care) >= r.want;
}

// geometry of the tile pass: TW = 16 columns, TH rows and CW channels per workgroup.  One thread owns one (tile column
// incl. halo, 4-channel vector): 18 * CW/4 <= DEC_TNT, and both LDS images -- the (TH+2) x 18 x CW f32 tile and, after
// the peak test, the compact peak list (u32 score bits + u16 position per possible peak) -- fit DEC_LDS_TILE
// (C = 80: 8 rows x all channels = 60 KB + 16 KB histogram -> two workgroups per CU)
#define DEC_TNT 512
struct DecGeom { int TH, CW, tiles_x, tiles_y, cchunks; };
static inline DecGeom dec_geom(int H, int W, int C) {
  DecGeom g;
  const int C4 = (C + 3) & ~3;
  g.CW = C4 < 112 ? C4 : 112;
  g.TH = 8;
  while (g.TH > 1 && ((g.TH + 2) * (DEC_TW + 2) * g.CW * 4 > DEC_LDS_TILE || g.TH * DEC_TW * g.CW * 6 > DEC_LDS_TILE)) g.TH >>= 1;
  g.tiles_x = (W + DEC_TW - 1) / DEC_TW;
  g.tiles_y = (H + g.TH - 1) / g.TH;
  g.cchunks = (C4 + g.CW - 1) / g.CW;
  return g;
}

__global__ void __launch_bounds__(256) dec_init_kernel(DecArgs a, long ws_words) {
  uint32_t* ws = a.ws + (long)blockIdx.x * ws_words;
  if (threadIdx.x < DEC_ST_WORDS) ws[threadIdx.x] = 0;
}

template <int TH>
__global__ void __launch_bounds__(DEC_TNT, 4) dec_tile_kernel(DecArgs a, int CW, int tiles_x, int tiles_y, long ws_words, int cap) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  uint32_t* lh = (uint32_t*)dsm;                    // DEC_HIST words
  float* tile = (float*)(dsm + DEC_HIST * 4);       // [(TH+2)*(TW+2)][CW]
  __shared__ uint32_t sh_base, sh_slot, sh_npk;
  const int b = blockIdx.y;
  const int tx = blockIdx.x % tiles_x, t2 = blockIdx.x / tiles_x;
  const int ty = t2 % tiles_y, cz = t2 / tiles_y;
  const int x0 = tx * DEC_TW, y0 = ty * TH, c0 = cz * CW;
  const int CV = CW >> 2;
  constexpr int PW = DEC_TW + 2;
  const int HW = a.H * a.W;
  const float* hb = a.heat + (long)b * HW * a.heat_stride;
  const bool vec = (a.heat_stride & 3) == 0;
  const int t = threadIdx.x, lane = t & 63;
  // this thread's column of the tile (px = 0 and 17 are the halo columns) and its 4 channels
  const int px = t / CV, cv = t - px * CV;
  const bool has_col = px < PW;
  const int x = x0 + px - 1, c = c0 + cv * 4;
  if (t == 0) sh_npk = 0;

  // ---- the column, TH + 2 rows, -> LDS; all loads of a thread in flight before the first store (pixels outside the
  // image and channels >= C read as -1: below every candidate)
  if (has_col) {
    f32x4 v[TH + 2];
    const bool col_ok = x >= 0 && x < a.W && c < a.C;
#pragma unroll
    for (int r = 0; r < TH + 2; ++r) {
      const int y = y0 + r - 1;
      v[r] = (f32x4){-1.f, -1.f, -1.f, -1.f};
      if (col_ok && y >= 0 && y < a.H) {
        const float* src = hb + (long)(y * a.W + x) * a.heat_stride + c;
        if (vec && c + 4 <= a.C) v[r] = *(const f32x4*)src;
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (c + e < a.C) v[r][e] = src[e];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < TH + 2; ++r) *(f32x4*)(tile + (long)(r * PW + px) * CW + cv * 4) = v[r];
  }
  __syncthreads();
  // ---- 3x3 peak test, separable: horizontal maxima of the TH + 2 rows, then the vertical maximum of three of them;
  // exact equality with the window maximum, positive values only.  The centre values stay in registers.
  f32x4 ctr[TH];
  uint32_t pk = 0;
  bool below = false;
  const float floorv = __uint_as_float(a.floor_bits);     // 0.0f when no floor is promised: excludes nothing new
  const bool interior = has_col && px >= 1 && px <= DEC_TW;
  if (interior) {
    f32x4 h[TH + 2];
    const float* colp = tile + (long)px * CW + cv * 4;
#pragma unroll
    for (int r = 0; r < TH + 2; ++r) {
      const float* p = colp + (long)r * PW * CW;
      const f32x4 cc = *(const f32x4*)p, l = *(const f32x4*)(p - CW), rr = *(const f32x4*)(p + CW);
      if (r >= 1 && r <= TH) ctr[r - 1] = cc;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[r][e] = fmaxf(fmaxf(l[e], cc[e]), rr[e]);
    }
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float m = fmaxf(fmaxf(h[r][e], h[r + 1][e]), h[r + 2][e]);
        if (ctr[r][e] == m && ctr[r][e] > floorv) pk |= 1u << (r * 4 + e);
        below |= ctr[r][e] > 0.f && ctr[r][e] < floorv;
      }
  }
  if (below) a.ws[(long)b * ws_words + ST_BELOW_FLOOR] = 1;
  __syncthreads();     // every thread has read its neighbours: the tile image may be overwritten
  // ---- compact list of the tile's peaks in LDS (over the tile image): score bits + position (local channel << 7 | row
  // << 4 | column).  Later passes cost per PEAK, not per element.
  uint32_t* pbits = (uint32_t*)tile;
  uint16_t* ppos = (uint16_t*)(pbits + TH * DEC_TW * CW);
  {
    // wave-level stream compaction, slot by slot: the lanes that hold a peak in register slot (r, e) write consecutive list
    // entries (ballot + prefix count), so the LDS stores of a wave never collide (a per-lane running position strides by
    // the lane's own peak count: 32-way bank conflicts on a plateau tile)
    uint32_t total = 0;
#pragma unroll
    for (int sl = 0; sl < 4 * TH; ++sl) total += (uint32_t)__popcll(__ballot((pk >> sl) & 1u));
    uint32_t run = 0;
    if (lane == 0 && total) run = atomicAdd(&sh_npk, total);
    run = __shfl(run, 0, 64);
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (total) {
#pragma unroll
      for (int r = 0; r < TH; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool on = (pk >> (r * 4 + e)) & 1u;
          const unsigned long long m = __ballot(on);
          if (on) {
            const uint32_t pos = run + (uint32_t)__popcll(m & lt);
            pbits[pos] = __float_as_uint(ctr[r][e]);
            ppos[pos] = (uint16_t)(((cv * 4 + e) << 7) | (r << 4) | (px - 1));
          }
          run += (uint32_t)__popcll(m);
        }
    }
  }
  __syncthreads();
  const uint32_t npk = sh_npk;
  // ---- the tile's own top K (exact superset), every level a scan of the compact list
  auto visit = [&](auto f) {
    for (uint32_t i = t; i < npk; i += DEC_TNT) {
      const uint32_t lp = ppos[i];
      const uint32_t pos = (uint32_t)((y0 + (int)((lp >> 4) & 7u)) * a.W + x0 + (int)(lp & 15u));
      f(pbits[i], (uint32_t)(c0 + (int)(lp >> 7)) * (uint32_t)HW + pos);
    }
  };
  const DecSel sel = dec_select<DEC_TNT>(visit, lh, (uint32_t)a.K, DEC_TILE_SLACK);
  // ---- survivors -> the image's candidate list: one global atomic per tile reserves the range, an LDS counter hands out
  // the slots
  uint32_t* ws = a.ws + (long)b * ws_words;
  if (t == 0) {
    const uint32_t n = sel.above + sel.inbin;       // takeall: every key of the tile (above = 0, inbin = their number)
    sh_slot = 0;
    sh_base = n ? atomicAdd(&ws[ST_NCAND], n) : 0u;
  }
  __syncthreads();
  uint64_t* cand = (uint64_t*)(ws + DEC_ST_WORDS);
  const uint32_t base = sh_base;
  visit([&](uint32_t bits, uint32_t canon) {
    if (!dec_selected(sel, bits, canon)) return;
    const uint32_t slot = base + atomicAdd(&sh_slot, 1u);
    if (slot < (uint32_t)cap) cand[slot] = ((uint64_t)bits << 32) | (uint64_t)(0xFFFFFFFFu - canon);
    else ws[ST_OVERFLOW] = 1;
  });
}

// one workgroup per image: exact top K of the candidate list, bitonic sort (descending) of the finalists, boxes
__global__ void __launch_bounds__(1024) dec_final_kernel(DecArgs a, long ws_words, int cap) {
  __shared__ uint64_t keys[DEC_NCAND];
  __shared__ uint32_t lh[DEC_HIST];
  __shared__ uint32_t sh_n;
  const int b = blockIdx.x;
  uint32_t* ws = a.ws + (long)b * ws_words;
  const uint64_t* cand = (const uint64_t*)(ws + DEC_ST_WORDS);
  uint32_t ncand = ws[ST_NCAND];
  if (ncand > (uint32_t)cap) ncand = (uint32_t)cap;
  const int t = threadIdx.x;
  auto visit = [&](auto f) {
    for (uint32_t i = t; i < ncand; i += 1024) {
      const uint64_t k = cand[i];
      f((uint32_t)(k >> 32), 0xFFFFFFFFu - (uint32_t)k);
    }
  };
  const DecSel sel = dec_select<1024>(visit, lh, (uint32_t)a.K, DEC_CAP);
  if (t == 0) sh_n = 0;
  __syncthreads();
  visit([&](uint32_t bits, uint32_t canon) {
    if (!dec_selected(sel, bits, canon)) return;
    const uint32_t slot = atomicAdd(&sh_n, 1u);
    if (slot < DEC_NCAND) keys[slot] = ((uint64_t)bits << 32) | (uint64_t)(0xFFFFFFFFu - canon);
    else ws[ST_OVERFLOW] = 1;     // cannot happen: <= K-1 + DEC_CAP by construction
  });
  __syncthreads();
  uint32_t n = sh_n;
  if (n > DEC_NCAND) n = DEC_NCAND;
  // sort only as wide as needed: next power of two >= n (typically a few hundred finalists, not 4096)
  int NS = 256;
  while ((uint32_t)NS < n) NS <<= 1;
  for (int i = t; i < NS; i += 1024) if ((uint32_t)i >= n) keys[i] = 0ull;
  __syncthreads();
  for (int k = 2; k <= NS; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < NS; i += 1024) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const uint64_t x = keys[i], y = keys[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? (x < y) : (x > y)) { keys[i] = y; keys[ixj] = x; }
        }
      }
      __syncthreads();
    }
  }
  const int HW = a.H * a.W;
  // fewer peaks (above the floor) than K.  What the reference's topk (centernet.py:408-424) returns next, in canonical
  // order: first the peaks that sit on the clamp floor, lowest flat index first (a trained map's background: the first
  // chunk of class 0 has them all); then, if the map runs out of those too (tiny maps), entries of the NMS-ed map that
  // are exactly 0 -- the non-peak positions of lowest flat index.  Both straight from the heat map.
  uint32_t* fill = lh;                      // the histogram is free now: [0, 2K) flags, [2048, 2048 + K) the fill list
  uint32_t nfloor = 0;                      // leading entries of the fill list that are floor peaks
  if (n < (uint32_t)a.K) {
    __shared__ uint32_t sh_got, wcnt[16];
    const float* hb = a.heat + (long)b * HW * a.heat_stride;
    const uint32_t total = (uint32_t)a.C * (uint32_t)HW, need = (uint32_t)a.K - n;
    const int lane = t & 63, wv = t >> 6;
    // value at canon position c and whether it is a 3x3 peak (positive, equal to the window maximum)
    auto probe = [&](uint32_t c, float& v) -> bool {
      const int cls = (int)(c / (uint32_t)HW), pix = (int)(c % (uint32_t)HW);
      const int x = pix % a.W, y = pix / a.W;
      v = hb[(long)pix * a.heat_stride + cls];
      if (!(v > 0.f)) return false;
      bool peak = true;
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int yy = y + dy, xx = x + dx;
          if (yy < 0 || yy >= a.H || xx < 0 || xx >= a.W) continue;
          peak &= hb[(long)(yy * a.W + xx) * a.heat_stride + cls] <= v;
        }
      return peak;
    };
    if (t == 0) sh_got = 0;
    __syncthreads();
    if (a.floor_bits) {
      for (uint32_t base = 0; base < total; base += 1024) {
        const uint32_t c = base + (uint32_t)t;
        bool on = false;
        if (c < total) {
          const int cls = (int)(c / (uint32_t)HW), pix = (int)(c % (uint32_t)HW);
          float v = hb[(long)pix * a.heat_stride + cls];
          if (__float_as_uint(v) == a.floor_bits) on = probe(c, v);
        }
        const unsigned long long m = __ballot(on);
        if (lane == 0) wcnt[wv] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = sh_got, all = 0;
        for (int w = 0; w < 16; ++w) { const uint32_t q = wcnt[w]; all += q; if (w < wv) before += q; }
        if (on) {
          const uint32_t pos = before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
          if (pos < need) fill[2048 + pos] = c;
        }
        __syncthreads();
        if (t == 0) sh_got += all;
        __syncthreads();
        if (sh_got >= need) break;
      }
    }
    const uint32_t got_f = sh_got < need ? sh_got : need;
    nfloor = got_f;
    if (got_f < need) {
      // fewer than K peaks in the whole map: of its first 2K positions at least K are not peaks
      const uint32_t span = 2u * (uint32_t)a.K < total ? 2u * (uint32_t)a.K : total;
      for (uint32_t c = t; c < span; c += 1024) { float v; fill[c] = probe(c, v) ? 1u : 0u; }
      __syncthreads();
      if (t == 0) {
        uint32_t got = got_f;
        for (uint32_t c = 0; c < span && got < need; ++c)
          if (!fill[c]) fill[2048 + got++] = c;
        for (; got < need; ++got) fill[2048 + got] = 0;      // map smaller than K positions
      }
      __syncthreads();
    }
  }
  for (int k = t; k < a.K; k += 1024) {
    const uint64_t key = k < NS ? keys[k] : 0ull;
    float score = 0.f; int cls = 0, ind = 0;
    if ((uint32_t)k < n) {
      score = __uint_as_float((uint32_t)(key >> 32));
      const uint32_t canon = 0xFFFFFFFFu - (uint32_t)key;
      cls = (int)(canon / HW);
      ind = (int)(canon % HW);
    } else {
      const uint32_t canon = fill[2048 + k - n];
      cls = (int)(canon / HW);
      ind = (int)(canon % HW);
      if ((uint32_t)k - n < nfloor) score = __uint_as_float(a.floor_bits);
    }
    const int x = ind % a.W, y = ind / a.W;
    const long pix = (long)b * HW + ind;
    float xs = (float)x, ys = (float)y;
    if (a.reg) { xs = xs + a.reg[pix * a.reg_stride]; ys = ys + a.reg[pix * a.reg_stride + 1]; }
    else { xs += 0.5f; ys += 0.5f; }
    const float w = a.wh[pix * a.wh_stride], h = a.wh[pix * a.wh_stride + 1];
    const long o = (long)b * a.K + k;
    a.boxes[o * 4 + 0] = (xs - w / 2) * a.down_ratio;
    a.boxes[o * 4 + 1] = (ys - h / 2) * a.down_ratio;
    a.boxes[o * 4 + 2] = (xs + w / 2) * a.down_ratio;
    a.boxes[o * 4 + 3] = (ys + h / 2) * a.down_ratio;
    a.scores[o] = score;
    a.classes[o] = cls;
    if (a.inds) a.inds[o] = ind;
  }
}

// inference_single_image (centernet.py:251-261: slice to max detections, score > threshold) followed by
// detector_postprocess (detectron2/modeling/postprocessing.py:11-72: scale, clip, drop empty boxes), for the
// whole batch: one wave per image, order-preserving compaction.  img_params[b] = {scale_x, scale_y, out_w, out_h}.
__global__ void __launch_bounds__(64) dec_postprocess_kernel(const float* __restrict__ boxes,
                                                            const float* __restrict__ scores,
                                                            const int* __restrict__ classes, int K, int max_det,
                                                            float thresh, const float* __restrict__ img_params,
                                                            float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                            int* __restrict__ out_classes, int* __restrict__ counts) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float sx = img_params[b * 4 + 0], sy = img_params[b * 4 + 1];
  const float ow = img_params[b * 4 + 2], oh = img_params[b * 4 + 3];
  int base = 0;
  for (int k0 = 0; k0 < K; k0 += 64) {
    const int k = k0 + lane;
    bool keep = false;
    float x1 = 0, y1 = 0, x2 = 0, y2 = 0, sc = 0;
    int cl = 0;
    if (k < K && k < max_det) {
      const long o = (long)b * K + k;
      sc = scores[o];
      cl = classes[o];
      x1 = fminf(fmaxf(boxes[o * 4 + 0] * sx, 0.f), ow);
      y1 = fminf(fmaxf(boxes[o * 4 + 1] * sy, 0.f), oh);
      x2 = fminf(fmaxf(boxes[o * 4 + 2] * sx, 0.f), ow);
      y2 = fminf(fmaxf(boxes[o * 4 + 3] * sy, 0.f), oh);
      keep = sc > thresh && (x2 - x1) > 0.f && (y2 - y1) > 0.f;
    }
    const unsigned long long m = __ballot(keep);
    if (keep) {
      const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
      const long o = (long)b * K + pos;
      out_boxes[o * 4 + 0] = x1; out_boxes[o * 4 + 1] = y1; out_boxes[o * 4 + 2] = x2; out_boxes[o * 4 + 3] = y2;
      out_scores[o] = sc;
      out_classes[o] = cl;
    }
    base += __popcll(m);
  }
  if (lane == 0) counts[b] = base;
}

int launch_postprocess(const float* boxes, const float* scores, const int* classes, int B, int K, int max_det,
                       float thresh, const float* img_params, float* out_boxes, float* out_scores, int* out_classes,
                       int* counts, hipStream_t s) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(dec_postprocess_kernel, dim3(B), dim3(64), 0, s, boxes, scores, classes, K, max_det, thresh,
                     img_params, out_boxes, out_scores, out_classes, counts);
  CTDET_LAUNCH_CHECK();
  return 0;
}

// candidates an image can produce: every tile keeps at most K-1 + DEC_TILE_SLACK
static long dec_cap(int H, int W, int C, int K) {
  const DecGeom g = dec_geom(H, W, C);
  return (long)g.tiles_x * g.tiles_y * g.cchunks * (K - 1 + DEC_TILE_SLACK);
}
static long dec_ws_words(int H, int W, int C, int K) { return DEC_ST_WORDS + 2 * dec_cap(H, W, C, K); }

size_t decode_workspace_bytes(int B, int H, int W, int C, int K) {
  return (size_t)B * dec_ws_words(H, W, C, K) * sizeof(uint32_t);
}
int decode_status_words(int H, int W, int C, int K, long* ws_words, int* below_word) {
  *ws_words = dec_ws_words(H, W, C, K);
  *below_word = ST_BELOW_FLOOR;
  return ST_OVERFLOW;
}

int launch_decode(const DecArgs& a, hipStream_t s) {
  CTDET_CHECK(a.C >= 1 && a.heat_stride >= a.C, "decode: C=%d / heat_stride=%d invalid", a.C, a.heat_stride);
  CTDET_CHECK(a.K >= 1 && a.K <= 1024 && a.K - 1 + DEC_CAP <= DEC_NCAND, "decode: K=%d out of range", a.K);
  CTDET_CHECK((long)a.C * a.H * a.W < (1L << 24), "decode: C*H*W too large for the 24-bit index field");
  CTDET_CHECK(((uintptr_t)a.heat & 15) == 0, "decode: heat must be 16-byte aligned");
  if (a.B == 0) return 0;
  const DecGeom g = dec_geom(a.H, a.W, a.C);
  const long words = dec_ws_words(a.H, a.W, a.C, a.K);
  const long cap = dec_cap(a.H, a.W, a.C, a.K);
  CTDET_CHECK(cap < (1L << 31), "decode: candidate list too large");
  const size_t tile_b = (size_t)(g.TH + 2) * (DEC_TW + 2) * g.CW * 4, list_b = (size_t)g.TH * DEC_TW * g.CW * 6;
  const size_t lds = DEC_HIST * 4 + (tile_b > list_b ? tile_b : list_b);
  CTDET_CHECK((DEC_TW + 2) * (g.CW / 4) <= DEC_TNT && g.CW <= 112, "decode: tile geometry");
  hipLaunchKernelGGL(dec_init_kernel, dim3(a.B), dim3(256), 0, s, a, words);
  const dim3 grid(g.tiles_x * g.tiles_y * g.cchunks, a.B);
  switch (g.TH) {
    case 8: hipLaunchKernelGGL(dec_tile_kernel<8>, grid, dim3(DEC_TNT), lds, s, a, g.CW, g.tiles_x, g.tiles_y, words, (int)cap); break;
    case 4: hipLaunchKernelGGL(dec_tile_kernel<4>, grid, dim3(DEC_TNT), lds, s, a, g.CW, g.tiles_x, g.tiles_y, words, (int)cap); break;
    case 2: hipLaunchKernelGGL(dec_tile_kernel<2>, grid, dim3(DEC_TNT), lds, s, a, g.CW, g.tiles_x, g.tiles_y, words, (int)cap); break;
    default: hipLaunchKernelGGL(dec_tile_kernel<1>, grid, dim3(DEC_TNT), lds, s, a, g.CW, g.tiles_x, g.tiles_y, words, (int)cap); break;
  }
  hipLaunchKernelGGL(dec_final_kernel, dim3(a.B), dim3(1024), 0, s, a, words, (int)cap);
  CTDET_LAUNCH_CHECK();
  return 0;
}
