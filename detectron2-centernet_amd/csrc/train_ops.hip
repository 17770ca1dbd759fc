// Training-side HBM-bound kernels: gaussian target splat, focal loss, masked L1 loss, SGD.
//   gaussian targets : detectron2/data/detection_utils.py:600-705 (gen_heatmap, gaussian_radius, gaussian2D,
//                      draw_umich_gaussian) -- the reference runs this per image / per object in numpy on the
//                      host inside model.forward (centernet.py:188); here one launch covers B x 128 objects.
//   focal loss       : detectron2/modeling/meta_arch/centernet.py:204, 333-369 (_neg_loss)
//   reg L1 loss      : centernet.py:372-397 (RegL1Loss, _transpose_and_gather_feat)
//   SGD              : detectron2/solver/build.py:93-137 -> torch.optim.SGD(momentum, weight_decay)
#include "common.h"

// ------------------------------------------------------------------------------------------------
// gaussian_radius: evaluated in f64 in exactly the reference's operation order; contraction is off so
// no FMA changes a rounding (the int() truncation downstream makes the last ulp matter).
// ------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
__host__ __device__ inline double ctdet_gaussian_radius_f64(int height, int width) {
  const double min_overlap = 0.7;
  const double hw = (double)(width * height);
  const double b1 = (double)(height + width);
  const double c1 = hw * (1 - min_overlap) / (1 + min_overlap);
  const double sq1 = sqrt(b1 * b1 - 4 * c1);
  const double r1 = (b1 + sq1) / 2;

  const double b2 = (double)(2 * (height + width));
  const double c2 = (1 - min_overlap) * (double)width * (double)height;
  const double sq2 = sqrt(b2 * b2 - 16 * c2);
  const double r2 = (b2 + sq2) / 2;

  const double a3 = 4 * min_overlap;
  const double b3 = -2 * min_overlap * (double)(height + width);
  const double c3 = (min_overlap - 1) * (double)width * (double)height;
  const double sq3 = sqrt(b3 * b3 - 4 * a3 * c3);
  const double r3 = (b3 + sq3) / 2;
  double r = r1 < r2 ? r1 : r2;
  r = r3 < r ? r3 : r;
  return r;
}

__global__ void gaussian_radius_kernel(const int* __restrict__ hw, int n, double* __restrict__ out_r,
                                       int* __restrict__ out_i) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double r = ctdet_gaussian_radius_f64(hw[2 * i], hw[2 * i + 1]);
  if (out_r) out_r[i] = r;
  if (out_i) { const int ri = (int)r; out_i[i] = ri > 0 ? ri : 0; }
}

// one workgroup per (object k, image b)
__global__ void __launch_bounds__(256) gaussian_targets_kernel(const float* __restrict__ boxes,
                                                               const int64_t* __restrict__ classes,
                                                               const int* __restrict__ counts, int Nmax, int H, int W,
                                                               int C, float* __restrict__ hm, float* __restrict__ wh,
                                                               float* __restrict__ reg, int64_t* __restrict__ ind,
                                                               uint8_t* __restrict__ reg_mask) {
  const int k = blockIdx.x, b = blockIdx.y;
  const long o = (long)b * 128 + k;
  int n = counts[b];
  n = n < 128 ? n : 128;
  n = n < Nmax ? n : Nmax;
  bool valid = k < n;
  float bx0 = 0, by0 = 0, bx1 = 0, by1 = 0, h = 0, w = 0;
  if (valid) {
    const float* bp = boxes + ((long)b * Nmax + k) * 4;
    bx0 = bp[0] / 4.f; by0 = bp[1] / 4.f; bx1 = bp[2] / 4.f; by1 = bp[3] / 4.f;
    h = by1 - by0; w = bx1 - bx0;
    valid = h > 0.f && w > 0.f;
  }
  if (!valid) {
    if (threadIdx.x == 0) {
      wh[o * 2] = 0.f; wh[o * 2 + 1] = 0.f; reg[o * 2] = 0.f; reg[o * 2 + 1] = 0.f; ind[o] = 0; reg_mask[o] = 0;
    }
    return;
  }
  const int radius_raw = (int)ctdet_gaussian_radius_f64((int)ceilf(h), (int)ceilf(w));
  const int radius = radius_raw > 0 ? radius_raw : 0;
  const float ctx = (bx0 + bx1) / 2.f, cty = (by0 + by1) / 2.f;
  const int cx = (int)ctx, cy = (int)cty;  // astype(int32): truncation toward zero
  if (cx < 0 || cx >= W || cy < 0 || cy >= H || ctx < 0.f || cty < 0.f) {
    // a box whose centre lies outside the map (unclipped annotations): the reference's gather raises an index error for
    // it (centernet.py:392-397); here the object is dropped instead of writing through an out-of-range index
    if (threadIdx.x == 0) {
      wh[o * 2] = 0.f; wh[o * 2 + 1] = 0.f; reg[o * 2] = 0.f; reg[o * 2 + 1] = 0.f; ind[o] = 0; reg_mask[o] = 0;
    }
    return;
  }
  if (threadIdx.x == 0) {
    wh[o * 2] = w; wh[o * 2 + 1] = h;
    ind[o] = (int64_t)cy * W + cx;
    reg[o * 2] = ctx - (float)cx; reg[o * 2 + 1] = cty - (float)cy;
    reg_mask[o] = 1;
  }
  const int cls = (int)classes[(long)b * Nmax + k];
  if (cls < 0 || cls >= C) return;
  const int diameter = 2 * radius + 1;
  const double sigma = (double)diameter / 6;
  const double denom = 2 * sigma * sigma;
  const double eps = 2.220446049250313e-16;  // np.finfo(float64).eps; h.max() == 1 (the centre)
  unsigned int* hmu = (unsigned int*)hm + (long)b * H * W * C + cls;
  for (int i = threadIdx.x; i < diameter * diameter; i += 256) {
    const int gy = i / diameter, gx = i % diameter;
    const int py = cy - radius + gy, px = cx - radius + gx;
    if (py < 0 || py >= H || px < 0 || px >= W) continue;
    const double yy = (double)(gy - radius), xx = (double)(gx - radius);
    double g = exp(-(xx * xx + yy * yy) / denom);
    if (g < eps) g = 0;
    const float gf = (float)g;
    atomicMax(hmu + ((long)py * W + px) * C, __float_as_uint(gf));  // all values >= 0: uint order == float order
  }
}
#pragma clang fp contract(fast)

int launch_gaussian_radius(const int* hw, int n, double* out_r, int* out_i, hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(gaussian_radius_kernel, dim3((n + 255) / 256), dim3(256), 0, s, hw, n, out_r, out_i);
  CTDET_LAUNCH_CHECK();
  return 0;
}

// the target map is cleared by a KERNEL, not hipMemsetAsync: inside a captured training step the memset became a memset node,
// and a replay launched on an idle stream ran it out of order with the splat kernel after it (two ranks sharing one GPU, the
// data-parallel step whose SGD launch is eager: hm_loss 93.7 -> 127 ... inf within 12 steps, profiles/r04_graph_memset_node.txt)
__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ p, long n) {
  const long n4 = n >> 2, stride = (long)gridDim.x * 256;
  float4* p4 = (float4*)p;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) p4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[(n4 << 2) + threadIdx.x] = 0.f;
}

int launch_gaussian_targets(const float* boxes, const int64_t* classes, const int* counts, int B, int Nmax, int H,
                            int W, int C, float* hm, float* wh, float* reg, int64_t* ind, uint8_t* reg_mask,
                            hipStream_t s) {
  if (B == 0) return 0;
  CTDET_CHECK(((uintptr_t)hm & 15) == 0, "gaussian_targets: hm must be 16-byte aligned");
  const long n = (long)B * H * W * C;
  if (ctdet_tuning_flags() & CTDET_TUNE_TARGETS_MEMSET) {
    const hipError_t e = hipMemsetAsync(hm, 0, (size_t)n * sizeof(float), s);
    CTDET_CHECK(e == hipSuccess, "gaussian_targets: memset failed: %s", hipGetErrorString(e));
  } else
  hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)std::min<long>((n / 4 + 255) / 256 + 1, 2048)), dim3(256), 0, s, hm, n);
  CTDET_LAUNCH_CHECK();
  hipLaunchKernelGGL(gaussian_targets_kernel, dim3(128, B), dim3(256), 0, s, boxes, classes, counts, Nmax, H, W, C, hm,
                     wh, reg, ind, reg_mask);
  CTDET_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// focal loss
// ------------------------------------------------------------------------------------------------
#define FL_VEC_PER_BLOCK (256 * 8)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

template <bool GRAD>
__global__ void __launch_bounds__(256) focal_main_kernel(const float* __restrict__ logits, const float* __restrict__ gt,
                                                         const float* __restrict__ alpha, long nvec, int C,
                                                         float grad_scale, float* __restrict__ grad,
                                                         float* __restrict__ partial) {
  __shared__ float red[3][4];
  float pos = 0.f, neg = 0.f, npos = 0.f;
  const long base = (long)blockIdx.x * FL_VEC_PER_BLOCK;
  for (int it = 0; it < 8; ++it) {
    const long i = base + it * 256 + threadIdx.x;
    if (i >= nvec) break;
    const f32x4 x = *(const f32x4*)(logits + i * 4);
    const f32x4 g = *(const f32x4*)(gt + i * 4);
    const int c0 = (int)((i * 4) % C);
    f32x4 gr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ce = c0 + e < C ? c0 + e : c0 + e - C;   // any class count: a vector may straddle two pixels
      const float s = ctdet_sigmoid_exact(x[e]);
      const float p = fminf(fmaxf(s, 1e-4f), 1.f - 1e-4f);
      const float dpdx = (s >= 1e-4f && s <= 1.f - 1e-4f) ? s * (1.f - s) : 0.f;
      float dLdp = 0.f;
      if (g[e] == 1.f) {
        const float a = alpha[ce % C], lp = logf(p), q = 1.f - p;
        pos += a * (lp * (q * q));
        npos += 1.f;
        if (GRAD) dLdp = a * (q * q / p - 2.f * q * lp);
      } else if (g[e] < 1.f) {
        const float om = 1.f - g[e];
        const float nw = (om * om) * (om * om);
        const float l1p = logf(1.f - p);
        neg += l1p * (p * p) * nw;
        if (GRAD) dLdp = nw * (2.f * p * l1p - p * p / (1.f - p));
      }
      gr[e] = -dLdp * dpdx * grad_scale;
    }
    if (GRAD) *(f32x4*)(grad + i * 4) = gr;
  }
  pos = wave_sum(pos); neg = wave_sum(neg); npos = wave_sum(npos);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { red[0][wv] = pos; red[1][wv] = neg; red[2][wv] = npos; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[(long)blockIdx.x * 3 + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partial[(long)blockIdx.x * 3 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    partial[(long)blockIdx.x * 3 + 2] = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
  }
}

// deterministic fixed-order reduction of the block partials (f64), one workgroup
__global__ void __launch_bounds__(256) focal_finalize_kernel(const float* __restrict__ partial, int nblocks,
                                                             float* __restrict__ loss, float* __restrict__ stats) {
  __shared__ double red[3][256];
  double a = 0, b = 0, c = 0;
  for (int i = threadIdx.x; i < nblocks; i += 256) { a += partial[i * 3]; b += partial[i * 3 + 1]; c += partial[i * 3 + 2]; }
  red[0][threadIdx.x] = a; red[1][threadIdx.x] = b; red[2][threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
      red[2][threadIdx.x] += red[2][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double pos = red[0][0], neg = red[1][0], np = red[2][0];
    const double l = np == 0 ? -neg : -(pos + neg) / np;
    loss[0] = (float)l;
    stats[0] = (float)pos; stats[1] = (float)neg; stats[2] = (float)np;
    stats[3] = np == 0 ? 1.f : (float)(1.0 / np);
  }
}

__global__ void __launch_bounds__(256) scale_by_dev_kernel(float* __restrict__ g, long nvec, const float* __restrict__ sc) {
  const float s = sc[0];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
    f32x4 v = *(f32x4*)(g + i * 4);
    v = v * s;
    *(f32x4*)(g + i * 4) = v;
  }
}

size_t focal_workspace_bytes(long numel) {
  const long nvec = (numel + 3) / 4;
  const long nblocks = (nvec + FL_VEC_PER_BLOCK - 1) / FL_VEC_PER_BLOCK;
  return (size_t)(nblocks * 3 + 4) * sizeof(float);
}

int launch_focal_loss(const float* logits, const float* gt, const float* alpha, int B, int H, int W, int C,
                      float grad_scale, void* workspace, float* loss, float* stats, float* grad, hipStream_t s) {
  const long numel = (long)B * H * W * C;
  CTDET_CHECK(numel % 4 == 0, "focal_loss: B*H*W*C=%ld must be a multiple of 4 (any C; maps are multiples of 4 pixels)", numel);
  CTDET_CHECK(numel > 0, "focal_loss: empty input");
  const long nvec = numel / 4;
  const int nblocks = (int)((nvec + FL_VEC_PER_BLOCK - 1) / FL_VEC_PER_BLOCK);
  float* partial = (float*)workspace;
  if (grad)
    hipLaunchKernelGGL((focal_main_kernel<true>), dim3(nblocks), dim3(256), 0, s, logits, gt, alpha, nvec, C, grad_scale,
                       grad, partial);
  else
    hipLaunchKernelGGL((focal_main_kernel<false>), dim3(nblocks), dim3(256), 0, s, logits, gt, alpha, nvec, C,
                       grad_scale, grad, partial);
  hipLaunchKernelGGL(focal_finalize_kernel, dim3(1), dim3(256), 0, s, partial, nblocks, loss, stats);
  if (grad) {
    int gb = (int)((nvec + 255) / 256);
    if (gb > 4096) gb = 4096;
    hipLaunchKernelGGL(scale_by_dev_kernel, dim3(gb), dim3(256), 0, s, grad, nvec, stats + 3);
  }
  CTDET_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// RegL1Loss: one workgroup; B*N entries (<= a few thousand)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) reg_l1_kernel(const float* __restrict__ pred, int pred_stride,
                                                     const uint8_t* __restrict__ mask, const int64_t* __restrict__ ind,
                                                     const float* __restrict__ target, int B, int N, int HW,
                                                     float grad_scale, float* __restrict__ loss,
                                                     float* __restrict__ grad, int grad_stride) {
  __shared__ double red[2][256];
  double sum = 0, msum = 0;
  const int total = B * N;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int b = i / N;
    const long id = ind[i];
    const float m = (id >= 0 && id < HW) ? (float)mask[i] : 0.f;   // an index outside the map never reaches memory
    const float* p = pred + ((long)b * HW + (m != 0.f ? id : 0)) * pred_stride;
    const float d0 = p[0] * m - target[i * 2] * m, d1 = p[1] * m - target[i * 2 + 1] * m;
    sum += (double)fabsf(d0) + (double)fabsf(d1);
    msum += 2.0 * m;
  }
  red[0][threadIdx.x] = sum; red[1][threadIdx.x] = msum;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
    __syncthreads();
  }
  const float denom = (float)red[1][0] + 1e-4f;
  if (threadIdx.x == 0) loss[0] = (float)red[0][0] / denom;
  if (!grad) return;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int b = i / N;
    const float m = (ind[i] >= 0 && ind[i] < HW) ? (float)mask[i] : 0.f;
    if (m == 0.f) continue;
    const float* p = pred + ((long)b * HW + ind[i]) * pred_stride;
    float* gp = grad + ((long)b * HW + ind[i]) * grad_stride;
    const float d0 = p[0] * m - target[i * 2] * m, d1 = p[1] * m - target[i * 2 + 1] * m;
    const float s0 = d0 > 0.f ? 1.f : (d0 < 0.f ? -1.f : 0.f), s1 = d1 > 0.f ? 1.f : (d1 < 0.f ? -1.f : 0.f);
    atomicAdd(gp, s0 * m / denom * grad_scale);
    atomicAdd(gp + 1, s1 * m / denom * grad_scale);
  }
}

int launch_reg_l1(const float* pred, int pred_stride, const uint8_t* mask, const int64_t* ind, const float* target,
                  int B, int N, int HW, float grad_scale, float* loss, float* grad, int grad_stride, hipStream_t s) {
  hipLaunchKernelGGL(reg_l1_kernel, dim3(1), dim3(256), 0, s, pred, pred_stride, mask, ind, target, B, N, HW, grad_scale,
                     loss, grad, grad_stride);
  CTDET_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// SGD with momentum + weight decay
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ m, long n, const float* __restrict__ lr_dev,
                                                  float mom, float wd, int first) {
  const float lr = lr_dev[0];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float gi = g[i];
    const float pi = p[i];
    if (wd != 0.f) gi = gi + wd * pi;
    const float bi = first ? gi : mom * m[i] + gi;
    m[i] = bi;
    p[i] = pi - lr * bi;
  }
}

// the same update over a flat buffer cut into `nruns` consecutive runs with their own (lr, weight decay): run r covers
// [run_end[r-1], run_end[r]); lr = lr_table[run_lr_index[r]] is read from device memory.  One launch instead of one per run
// (DLA-34: 109 runs, weights / norm / bias parameters alternate in backward order).
__global__ void __launch_bounds__(256) sgd_runs_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       long n, const long* __restrict__ run_end,
                                                       const int* __restrict__ run_lr_index, const float* __restrict__ run_wd,
                                                       const float* __restrict__ lr_table, int nruns, float mom, int first) {
  // A workgroup owns a contiguous range and a thread walks it in steps of 1024 elements, four at a time: a thread stays
  // inside one run for many iterations, so the run lookup (a binary search over dependent global loads) is rare.  With a
  // grid-wide stride every iteration landed in another run and searched again (204 us for 18.6 M parameters).
  const long per = (((n + gridDim.x - 1) / gridDim.x) + 1023) & ~1023L;
  const long beg = (long)blockIdx.x * per, end = beg + per < n ? beg + per : n;
  long cur_end = -1;
  float lr = 0.f, wd = 0.f;
  auto lookup = [&](long i) {
    int lo = 0, hi = nruns - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (run_end[mid] > i) hi = mid; else lo = mid + 1;
    }
    cur_end = run_end[lo];
    lr = lr_table[run_lr_index[lo]];
    wd = run_wd[lo];
  };
  for (long i = beg + (long)threadIdx.x * 4; i < end; i += 1024) {
    if (i >= cur_end) lookup(i);
    if (i + 4 <= cur_end && i + 4 <= end) {
      const float4 gv = *(const float4*)(g + i), pv = *(const float4*)(p + i);
      float4 mv = {0.f, 0.f, 0.f, 0.f};
      if (!first) mv = *(const float4*)(m + i);
      const float ge[4] = {gv.x, gv.y, gv.z, gv.w}, pe[4] = {pv.x, pv.y, pv.z, pv.w}, me[4] = {mv.x, mv.y, mv.z, mv.w};
      float bo[4], po[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float gi = ge[k];
        if (wd != 0.f) gi = gi + wd * pe[k];
        bo[k] = first ? gi : mom * me[k] + gi;
        po[k] = pe[k] - lr * bo[k];
      }
      *(float4*)(m + i) = make_float4(bo[0], bo[1], bo[2], bo[3]);
      *(float4*)(p + i) = make_float4(po[0], po[1], po[2], po[3]);
    } else {
      for (long j = i; j < i + 4 && j < end; ++j) {
        if (j >= cur_end) lookup(j);
        float gi = g[j];
        const float pi = p[j];
        if (wd != 0.f) gi = gi + wd * pi;
        const float bi = first ? gi : mom * m[j] + gi;
        m[j] = bi;
        p[j] = pi - lr * bi;
      }
    }
  }
}

int launch_sgd_runs(float* p, const float* g, float* m, long n, const long* run_end, const int* run_lr_index,
                    const float* run_wd, const float* lr_table, int nruns, float mom, int first, hipStream_t s) {
  if (n == 0 || nruns == 0) return 0;
  long nb = (n + 1023) / 1024;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(sgd_runs_kernel, dim3((unsigned)nb), dim3(256), 0, s, p, g, m, n, run_end, run_lr_index, run_wd, lr_table,
                     nruns, mom, first);
  CTDET_LAUNCH_CHECK();
  return 0;
}

int launch_sgd(float* p, const float* g, float* m, long n, const float* lr_dev, float mom, float wd, int first,
               hipStream_t s) {
  if (n == 0) return 0;
  long nb = (n + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)nb), dim3(256), 0, s, p, g, m, n, lr_dev, mom, wd, first);
  CTDET_LAUNCH_CHECK();
  return 0;
}
