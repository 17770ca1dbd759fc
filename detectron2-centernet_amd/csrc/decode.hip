// Batched CenterNet decode: 3x3 max-pool peak test + exact top-K + box assembly.
// Reference: detectron2/modeling/meta_arch/centernet.py:399-405 (_nms, `hmax == heat` exact equality,
// plateaus all kept), :408-424 (_topk: per-class top-K then top-K of C*K == global top-K),
// :426-458 (ctdet_decode).  The reference handles batch==1 only and loops over images in Python
// (:224-233); here the whole batch is one set of launches with no host round trip.
//
// Layout: heat is f32 NHWC [B,H,W,C] (the layout the head conv writes).  Ordering contract
// ("canonical order"): score descending, ties by the reference's flat NCHW index
// canon = c*H*W + y*W + x ascending.  torch.topk's own tie order is unspecified, so ties are
// where a difference is permitted; fixtures are tie-free or assert the canonical rule.
//
// Selection = MSD radix select on the key (score bits, ~canon), 4096 bins per level: level 0 uses fine bins
// over the sigmoid range [2^-15, 2); further levels run only while the bin holding the K-th key has more than
// DEC_CAP entries (near-constant or tied maps) and early-exit otherwise.  They split what level 0 left: for an
// interior level-0 bin the low 15 score bits + 24 index bits, for the two clamped end bins the full 32 + 24
// bits, 12 bits per level.  Then the <= K-1 certain + <= CAP uncertain candidates are sorted by one workgroup
// per image.
// HBM traffic: heat is read twice (histogram pass, collect pass), each pass tile by tile so that the 3x3
// neighbours come from L1.
#include "common.h"
#include <stdlib.h>

#define DEC_CAP 2048            // max uncertain candidates carried to the final sort
#define DEC_NCAND 4096          // sort width (>= K-1 + DEC_CAP)
#define DEC_HIST 4096           // bins per level
#define DEC_LEVELS 6            // 1 + ceil(56 / 12)
// per-image workspace (uint32 words): [0..31] state, [32..32+DEC_HIST) histogram, then candidates (u64)
#define DEC_ST_WORDS 32
#define DEC_WS_WORDS (DEC_ST_WORDS + DEC_HIST + 2 * DEC_NCAND)
enum { ST_RESOLVED = 0, ST_NABOVE = 1, ST_LEVEL = 2, ST_P0 = 3, ST_PR_LO = 4, ST_PR_HI = 5, ST_NCAND = 6,
       ST_OVERFLOW = 7, ST_TAKEALL = 8 };

__device__ __forceinline__ int dec_d0(uint32_t bits) {
  const int d = ((int)bits - 0x38000000) >> 15;
  return d < 0 ? 0 : (d > 4095 ? 4095 : d);
}
// what is left to order after level 0, left-aligned in 60 bits (5 digits of 12)
__device__ __forceinline__ uint64_t dec_rest(uint32_t bits, uint32_t canon, bool interior) {
  const uint64_t inv = (uint64_t)(0xFFFFFFu - canon);
  return interior ? ((((uint64_t)(bits & 0x7FFFu) << 24) | inv) << 21)   // 39 significant bits
                  : ((((uint64_t)bits << 24) | inv) << 4);                // 56 significant bits
}

// visits every positive peak of image b handled by this block: f(bits, canon).
// A block owns a DEC_TH x DEC_TW pixel tile (all channels).  A thread walks a (pixel column, 4-channel vector) down
// the tile rows with the separable form of the 3x3 max: per row it loads the three horizontal neighbours once
// (consecutive threads -> consecutive 16-byte vectors), keeps the horizontal maxima of the last three rows in
// registers, and tests the middle one.  3 (TH+2)/TH loads per element instead of 9, every heat element fetched from
// L2/HBM about (TH+2)/TH times per pass.  (The first version strode the flat index space over the whole grid with 9
// loads per element: rocprof showed 7x the heat map in FETCH_SIZE per pass.)
#define DEC_TH 16
#define DEC_TW 16
template <typename F>
__device__ __forceinline__ void for_each_peak(const DecArgs& a, int b, F f) {
  const int CV = a.C >> 2;
  const int strips = (a.W + DEC_TW - 1) / DEC_TW;
  const int x0 = (blockIdx.x % strips) * DEC_TW, y0 = (blockIdx.x / strips) * DEC_TH;
  const float* hb = a.heat + (long)b * a.H * a.W * a.C;
  const int HW = a.H * a.W;
  const int ncol = DEC_TW * CV;
  const f32x4 neg = {-1.f, -1.f, -1.f, -1.f};   // below every candidate (only v > 0 can be a peak)
  for (int i = threadIdx.x; i < ncol; i += blockDim.x) {
    const int cv = i % CV;
    const int x = x0 + i / CV;
    if (x >= a.W) continue;
    const bool hasl = x > 0, hasr = x + 1 < a.W;
    const float* col = hb + (long)x * a.C + cv * 4;
    // horizontal 3-max of a row (and its centre value); rows outside the image contribute nothing
    auto hrow = [&](int y, f32x4& ctr) {
      if (y < 0 || y >= a.H) { ctr = neg; return neg; }
      const float* r = col + (long)y * a.W * a.C;
      ctr = *(const f32x4*)r;
      f32x4 m = ctr;
      if (hasl) { const f32x4 n = *(const f32x4*)(r - a.C);
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = n[e] > m[e] ? n[e] : m[e]; }
      if (hasr) { const f32x4 n = *(const f32x4*)(r + a.C);
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = n[e] > m[e] ? n[e] : m[e]; }
      return m;
    };
    f32x4 c_prev, c_cur, c_next;
    f32x4 h_prev = hrow(y0 - 1, c_prev);
    f32x4 h_cur = hrow(y0, c_cur);
    for (int yl = 0; yl < DEC_TH; ++yl) {
      const int y = y0 + yl;
      if (y >= a.H) break;
      const f32x4 h_next = hrow(y + 1, c_next);
      const int p = y * a.W + x;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float mx = h_cur[e];
        mx = h_prev[e] > mx ? h_prev[e] : mx;
        mx = h_next[e] > mx ? h_next[e] : mx;
        const float v = c_cur[e];
        if (v == mx && v > 0.f) f(__float_as_uint(v), (uint32_t)((cv * 4 + e) * HW + p));
      }
      h_prev = h_cur; h_cur = h_next; c_cur = c_next;
    }
  }
}

__global__ void __launch_bounds__(256) dec_init_kernel(DecArgs a) {
  uint32_t* ws = a.ws + (long)blockIdx.x * DEC_WS_WORDS;
  for (int i = threadIdx.x; i < DEC_ST_WORDS + DEC_HIST; i += 256) ws[i] = 0;
}

__global__ void __launch_bounds__(256) dec_hist_kernel(DecArgs a, int level, int dbg) {
  extern __shared__ uint32_t lh[];
  const int b = blockIdx.y;
  uint32_t* ws = a.ws + (long)b * DEC_WS_WORDS;
  if (ws[ST_RESOLVED]) return;
  const int nb = DEC_HIST;
  for (int i = threadIdx.x; i < nb; i += 256) lh[i] = 0;
  __syncthreads();
  const uint32_t p0 = ws[ST_P0];
  const bool interior = p0 > 0 && p0 < 4095;
  const uint64_t pr = ((uint64_t)ws[ST_PR_HI] << 32) | ws[ST_PR_LO];
  // run-length accumulation: a thread walks a pixel column, and real heat maps cluster (most values sit within a few
  // fine bins around sigmoid(bias)), so consecutive peaks of a thread usually share a bin -- counting the run and issuing
  // one LDS atomic per run removes most of the same-address serialisation
  uint32_t run_bin = 0xFFFFFFFFu, run_cnt = 0;
  auto count = [&](uint32_t bin) {
    if (bin == run_bin) { ++run_cnt; return; }
    if (run_cnt) atomicAdd(&lh[run_bin], run_cnt);
    run_bin = bin; run_cnt = 1;
  };
  for_each_peak(a, b, [&](uint32_t bits, uint32_t canon) {
    if (level == 0) {
      if (!(dbg & 1)) count((uint32_t)dec_d0(bits));
    } else {
      if ((uint32_t)dec_d0(bits) != p0) return;
      const uint64_t r = dec_rest(bits, canon, interior);
      if (level > 1 && (r >> (60 - 12 * (level - 1))) != pr) return;
      count((uint32_t)(r >> (60 - 12 * level)) & 0xFFFu);
    }
  });
  if (run_cnt) atomicAdd(&lh[run_bin], run_cnt);
  __syncthreads();
  uint32_t* gh = ws + DEC_ST_WORDS;
  if (dbg & 2) return;
  for (int i = threadIdx.x; i < nb; i += 256)
    if (lh[i]) atomicAdd(&gh[i], lh[i]);
}

// one workgroup per image: find the digit holding the K-th largest key at this level
__global__ void __launch_bounds__(1024) dec_scan_kernel(DecArgs a, int level) {
  __shared__ uint32_t part[1024];
  __shared__ uint32_t sh_T, sh_above;
  const int b = blockIdx.x;
  uint32_t* ws = a.ws + (long)b * DEC_WS_WORDS;
  if (ws[ST_RESOLVED]) return;
  uint32_t* gh = ws + DEC_ST_WORDS;
  const int nb = DEC_HIST;
  const int per = nb / 1024;  // 4 bins per thread, thread t owns bins [t*per, (t+1)*per)
  const int t = threadIdx.x;
  uint32_t loc[4];
  uint32_t s = 0;
  for (int j = 0; j < per; ++j) { loc[j] = gh[t * per + j]; s += loc[j]; }
  part[t] = s;
  __syncthreads();
  // inclusive suffix sum over threads (sum of bins owned by threads >= t)
  for (int off = 1; off < 1024; off <<= 1) {
    const uint32_t add = (t + off < 1024) ? part[t + off] : 0;
    __syncthreads();
    part[t] += add;
    __syncthreads();
  }
  const uint32_t need = (uint32_t)a.K - ws[ST_NABOVE];
  const uint32_t total = part[0];
  if (t == 0) { sh_T = 0xFFFFFFFFu; sh_above = 0; }
  __syncthreads();
  if (total >= need) {
    const uint32_t above_me = (t + 1 < 1024) ? part[t + 1] : 0;  // keys in bins of higher threads
    if (above_me < need && part[t] >= need) {
      uint32_t cum = above_me;
      for (int j = per - 1; j >= 0; --j) {
        if (cum + loc[j] >= need) { sh_T = t * per + j; sh_above = cum; break; }
        cum += loc[j];
      }
    }
  }
  __syncthreads();
  for (int j = 0; j < per; ++j) gh[t * per + j] = 0;  // ready for the next level
  if (t == 0) {
    if (total < need) {
      // fewer positive peaks than K (only possible at level 0): take them all
      ws[ST_TAKEALL] = 1;
      ws[ST_RESOLVED] = 1;
      ws[ST_LEVEL] = 0;
    } else {
      const uint32_t T = sh_T;
      ws[ST_NABOVE] += sh_above;
      ws[ST_LEVEL] = level;
      if (level == 0) ws[ST_P0] = T;
      else {
        const uint64_t pr = (((uint64_t)ws[ST_PR_HI] << 32) | ws[ST_PR_LO]);
        const uint64_t npr = (level == 1 ? 0ull : (pr << 12)) | T;
        ws[ST_PR_LO] = (uint32_t)npr;
        ws[ST_PR_HI] = (uint32_t)(npr >> 32);
      }
    }
  }
  __syncthreads();
  // the owner of bin T decides whether the uncertain bin fits the final sort
  if (total >= need && sh_T != 0xFFFFFFFFu && (int)(sh_T / per) == t) {
    const uint32_t cnt = loc[sh_T % per];
    if (cnt <= DEC_CAP || level == DEC_LEVELS - 1) {
      ws[ST_RESOLVED] = 1;
      if (cnt > DEC_CAP) ws[ST_OVERFLOW] = 1;  // cannot happen: last level keys are unique
    }
  }
}

__global__ void __launch_bounds__(256) dec_collect_kernel(DecArgs a) {
  const int b = blockIdx.y;
  uint32_t* ws = a.ws + (long)b * DEC_WS_WORDS;
  const int level = (int)ws[ST_LEVEL];
  const uint32_t p0 = ws[ST_P0];
  const uint64_t pr = ((uint64_t)ws[ST_PR_HI] << 32) | ws[ST_PR_LO];
  const bool takeall = ws[ST_TAKEALL] != 0;
  uint64_t* cand = (uint64_t*)(ws + DEC_ST_WORDS + DEC_HIST);
  for_each_peak(a, b, [&](uint32_t bits, uint32_t canon) {
    if (!takeall) {
      const uint32_t d0 = (uint32_t)dec_d0(bits);
      if (d0 < p0) return;
      if (d0 == p0 && level > 0) {
        const uint64_t r = dec_rest(bits, canon, p0 > 0 && p0 < 4095) >> (60 - 12 * level);
        if (r < pr) return;
      }
    }
    const uint32_t slot = atomicAdd(&ws[ST_NCAND], 1u);
    if (slot < DEC_NCAND) cand[slot] = ((uint64_t)bits << 32) | (uint64_t)(0xFFFFFFFFu - canon);
    else ws[ST_OVERFLOW] = 1;
  });
}

// one workgroup per image: bitonic sort (descending) of the candidates, emit top K
__global__ void __launch_bounds__(1024) dec_final_kernel(DecArgs a) {
  __shared__ uint64_t keys[DEC_NCAND];
  const int b = blockIdx.x;
  uint32_t* ws = a.ws + (long)b * DEC_WS_WORDS;
  const uint64_t* cand = (const uint64_t*)(ws + DEC_ST_WORDS + DEC_HIST);
  uint32_t n = ws[ST_NCAND];
  if (n > DEC_NCAND) n = DEC_NCAND;
  const int t = threadIdx.x;
  // sort only as wide as needed: next power of two >= n (typically a few hundred candidates, not 4096)
  int NS = 256;
  while ((uint32_t)NS < n) NS <<= 1;
  for (int i = t; i < NS; i += 1024) keys[i] = (uint32_t)i < n ? cand[i] : 0ull;
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
  for (int k = t; k < a.K; k += 1024) {
    const uint64_t key = k < NS ? keys[k] : 0ull;
    float score = 0.f; int cls = 0, ind = 0;
    if ((uint32_t)k < n) {
      score = __uint_as_float((uint32_t)(key >> 32));
      const uint32_t canon = 0xFFFFFFFFu - (uint32_t)key;
      cls = (int)(canon / HW);
      ind = (int)(canon % HW);
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

size_t decode_workspace_bytes(int B) { return (size_t)B * DEC_WS_WORDS * sizeof(uint32_t); }

int launch_decode(const DecArgs& a, hipStream_t s) {
  CTDET_CHECK(a.C % 4 == 0, "decode: C=%d must be a multiple of 4", a.C);
  CTDET_CHECK(a.K >= 1 && a.K <= 1024 && a.K - 1 + DEC_CAP <= DEC_NCAND, "decode: K=%d out of range", a.K);
  CTDET_CHECK((long)a.C * a.H * a.W < (1L << 24), "decode: C*H*W too large for the 24-bit index field");
  CTDET_CHECK(((uintptr_t)a.heat & 15) == 0, "decode: heat must be 16-byte aligned");
  if (a.B == 0) return 0;
  const int chunks = ((a.H + DEC_TH - 1) / DEC_TH) * ((a.W + DEC_TW - 1) / DEC_TW);  // pixel tiles per image
  hipLaunchKernelGGL(dec_init_kernel, dim3(a.B), dim3(256), 0, s, a);
  for (int level = 0; level < DEC_LEVELS; ++level) {
    const size_t lds = DEC_HIST * sizeof(uint32_t);
    static const int dbg = getenv("CTDET_DEC_DEBUG") ? atoi(getenv("CTDET_DEC_DEBUG")) : 0;
    hipLaunchKernelGGL(dec_hist_kernel, dim3(chunks, a.B), dim3(256), lds, s, a, level, dbg);
    hipLaunchKernelGGL(dec_scan_kernel, dim3(a.B), dim3(1024), 0, s, a, level);
  }
  hipLaunchKernelGGL(dec_collect_kernel, dim3(chunks, a.B), dim3(256), 0, s, a);
  hipLaunchKernelGGL(dec_final_kernel, dim3(a.B), dim3(1024), 0, s, a);
  CTDET_LAUNCH_CHECK();
  return 0;
}
