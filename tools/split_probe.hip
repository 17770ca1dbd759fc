// Dev tool for the f16x3 ("split") arithmetic of the conv kernels: an f32 operand x is carried as hi + lo with
// hi = f16(x), lo = f16(x - hi) and a product a*w is evaluated as a_hi*w_hi + a_lo*w_hi + a_hi*w_lo on the f16 matrix pipe
// with f32 accumulation (the dropped a_lo*w_lo term is ~2^-22 relative).
//   part 1: rate of the MFMA sequences that can carry the three products of a 16-channel K step (see rate_loop), random
//           operands, 1/2/4 waves per SIMD
//   part 2: numerics of one 16x16 output tile, K = 256, against an f64 reference (see tile_kernel): plain f16 operands,
//           an f32 fma chain and the candidate split sequences; operand scales 1, 1e-3 and 1e-6 (f16 subnormal range
//           for the lo parts)
//   hipcc --offload-arch=gfx950 -O3 tools/split_probe.hip -o /tmp/split_probe && /tmp/split_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16;
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: 8x 16x16x32 (independent accumulators)      1: 8x 16x16x16
//      2: per accumulator 32 then 16, back to back (dependent pair of DIFFERENT opcodes)
//      3: all eight 32s, then all eight 16s (same accumulators, eight instructions apart)
//      4: per accumulator three 16x16x16 back to back     5: three rounds of eight 16x16x16
//      6: per accumulator two 16x16x32 back to back       7: two rounds of eight 16x16x32
template <int MODE>
__global__ void __launch_bounds__(256) rate_loop(const f16x8* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  f16x8 a[2], b[4];
  for (int i = 0; i < 2; ++i) a[i] = in[(tid * 6 + i) & 4095];
  for (int i = 0; i < 4; ++i) b[i] = in[(tid * 6 + 2 + i) & 4095];
  f16x4 a4[2], b4[4], a4b[2], b4b[4];
  for (int i = 0; i < 2; ++i) { a4[i] = __builtin_shufflevector(a[i], a[i], 4, 5, 6, 7); a4b[i] = __builtin_shufflevector(a[i], a[i], 0, 1, 2, 3); }
  for (int i = 0; i < 4; ++i) { b4[i] = __builtin_shufflevector(b[i], b[i], 0, 1, 2, 3); b4b[i] = __builtin_shufflevector(b[i], b[i], 4, 5, 6, 7); }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define M32(i, j) acc[(i) * 4 + (j)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[(i) * 4 + (j)], 0, 0, 0)
#define M32B(i, j) acc[(i) * 4 + (j)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1 - (i)], b[j], acc[(i) * 4 + (j)], 0, 0, 0)
#define M16(i, j) acc[(i) * 4 + (j)] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4[i], b4[j], acc[(i) * 4 + (j)], 0, 0, 0)
#define M16B(i, j) acc[(i) * 4 + (j)] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4b[i], b4[j], acc[(i) * 4 + (j)], 0, 0, 0)
#define M16C(i, j) acc[(i) * 4 + (j)] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4b[i], b4b[j], acc[(i) * 4 + (j)], 0, 0, 0)
#define ALL(X) _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) { X; }
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) { ALL(M32(i, j)) }
    if constexpr (MODE == 1) { ALL(M16(i, j)) }
    if constexpr (MODE == 2) { ALL(M32(i, j); M16(i, j)) }
    if constexpr (MODE == 3) { ALL(M32(i, j)) ALL(M16(i, j)) }
    if constexpr (MODE == 4) { ALL(M16(i, j); M16B(i, j); M16C(i, j)) }
    if constexpr (MODE == 5) { ALL(M16(i, j)) ALL(M16B(i, j)) ALL(M16C(i, j)) }
    if constexpr (MODE == 6) { ALL(M32(i, j); M32B(i, j)) }
    if constexpr (MODE == 7) { ALL(M32(i, j)) ALL(M32B(i, j)) }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[tid] = s;
}

__device__ __forceinline__ f16x8 split4(const f32x4 x) {
  f16x8 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f16 h = (f16)x[j];
    r[j] = h;
    r[4 + j] = (f16)(x[j] - (float)h);
  }
  return r;
}

// one wave: D[16 couts][16 pixels] = sum_k W[cout][k] A[pixel][k], K in steps of 16 k.
// mode 0: plain f16 operands; 2: v_mfma_f32_16x16x4_f32; split schemes:
//   1: 16x16x32 then 16x16x16 on the same accumulator, back to back    3: the same with s_nop 15 x2 between them
//   4: the 32s and the 16s into two accumulators, summed at the end      5: three 16x16x16 on one accumulator
//   6: two 16x16x32 on one accumulator ([whi|whi].[hi|lo] + [wlo|0].[hi|lo])
__global__ void __launch_bounds__(64) tile_kernel(const float* __restrict__ W, const float* __restrict__ A, float* __restrict__ D,
                                                 int K, int mode) {
  const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 16) {
    const f32x4 w = *(const f32x4*)(W + r * K + k0 + 4 * q);
    const f32x4 a = *(const f32x4*)(A + r * K + k0 + 4 * q);
    if (mode == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], a[e], acc, 0, 0, 0);
    } else if (mode == 0) {
      f16x4 wh, ah;
#pragma unroll
      for (int j = 0; j < 4; ++j) { wh[j] = (f16)w[j]; ah[j] = (f16)a[j]; }
      acc = __builtin_amdgcn_mfma_f32_16x16x16f16(wh, ah, acc, 0, 0, 0);
    } else {
      const f16x8 ws = split4(w), as = split4(a);
      const f16x8 a1 = __builtin_shufflevector(ws, ws, 0, 1, 2, 3, 0, 1, 2, 3);
      const f16x4 whi = __builtin_shufflevector(ws, ws, 0, 1, 2, 3), wlo = __builtin_shufflevector(ws, ws, 4, 5, 6, 7);
      const f16x4 ahi = __builtin_shufflevector(as, as, 0, 1, 2, 3), alo = __builtin_shufflevector(as, as, 4, 5, 6, 7);
      if (mode == 1) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, as, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(wlo, ahi, acc, 0, 0, 0);
      } else if (mode == 3) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, as, acc, 0, 0, 0);
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc));
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(wlo, ahi, acc, 0, 0, 0);
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc));
      } else if (mode == 4) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, as, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x16f16(wlo, ahi, acc2, 0, 0, 0);
      } else if (mode == 5) {
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(whi, ahi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(whi, alo, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(wlo, ahi, acc, 0, 0, 0);
      } else {
        f16x8 a2 = ws;
        a2[0] = ws[4]; a2[1] = ws[5]; a2[2] = ws[6]; a2[3] = ws[7];
        a2[4] = a2[5] = a2[6] = a2[7] = (f16)0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, as, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, as, acc, 0, 0, 0);
      }
    }
  }
  acc = acc + acc2;
#pragma unroll
  for (int i = 0; i < 4; ++i) D[(4 * q + i) * 16 + r] = acc[i];   // D[cout][pixel]
}

template <int MODE>
static void run_rate(const char* name, double flop_per_iter_per_wave, const f16x8* din, int cus) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 20000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int blocks = cus * wps;
    float* dout;
    (void)hipMalloc(&dout, (size_t)blocks * 256 * sizeof(float));
    hipLaunchKernelGGL(rate_loop<MODE>, dim3(blocks), dim3(256), 0, 0, din, dout, 2000);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(rate_loop<MODE>, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * flop_per_iter_per_wave;
    printf("%-34s %d wave(s)/SIMD: %7.2f ms  %6.0f TFLOP/s  %.2f ns per loop body per SIMD\n", name, wps, ms, flops / ms / 1e9,
           ms * 1e6 / ((double)iters * wps));
    (void)hipFree(dout);
  }
}

int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  std::vector<f16> h(4096 * 8);
  srand(1);
  for (auto& v : h) v = (f16)((rand() % 2001 - 1000) / 4000.0f);
  f16x8* din;
  (void)hipMalloc(&din, h.size() * sizeof(f16));
  (void)hipMemcpy(din, h.data(), h.size() * sizeof(f16), hipMemcpyHostToDevice);
  const double T = 2.0 * 16 * 16;   // FLOP per unit of K of one 16x16 tile
  run_rate<0>("8x 16x16x32", 8 * T * 32, din, cus);
  run_rate<1>("8x 16x16x16", 8 * T * 16, din, cus);
  run_rate<2>("8x (32,16 back to back)", 8 * T * 48, din, cus);
  run_rate<3>("8x 32 then 8x 16", 8 * T * 48, din, cus);
  run_rate<4>("8x (16,16,16 back to back)", 8 * T * 48, din, cus);
  run_rate<5>("3 rounds of 8x 16x16x16", 8 * T * 48, din, cus);
  run_rate<6>("8x (32,32 back to back)", 8 * T * 64, din, cus);
  run_rate<7>("2 rounds of 8x 16x16x32", 8 * T * 64, din, cus);

  const int K = 256;
  float *dW, *dA, *dD;
  (void)hipMalloc(&dW, 16 * K * 4);
  (void)hipMalloc(&dA, 16 * K * 4);
  (void)hipMalloc(&dD, 256 * 4);
  const float scales[3] = {1.f, 1e-3f, 1e-6f};
  for (int si = 0; si < 3; ++si)
    for (int sj = 0; sj < 3; ++sj) {
      std::vector<float> W(16 * K), A(16 * K), D(256);
      for (auto& v : W) v = scales[si] * ((rand() % 20001 - 10000) / 10000.0f);
      for (auto& v : A) v = scales[sj] * ((rand() % 20001 - 10000) / 10000.0f);
      (void)hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
      (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
      printf("|w| ~ %g, |a| ~ %g:", scales[si], scales[sj]);
      for (int mode = 0; mode < 7; ++mode) {
        hipLaunchKernelGGL(tile_kernel, dim3(1), dim3(64), 0, 0, dW, dA, dD, K, mode);
        (void)hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
        double worst = 0.0;
        for (int c = 0; c < 16; ++c)
          for (int p = 0; p < 16; ++p) {
            double ref = 0.0, mag = 0.0;
            for (int k = 0; k < K; ++k) { ref += (double)W[c * K + k] * A[p * K + k]; mag += fabs((double)W[c * K + k] * A[p * K + k]); }
            worst = fmax(worst, fabs(D[c * 16 + p] - ref) / mag);
          }
        printf("  m%d %.1e", mode, worst);
      }
      printf("   (max |err| / sum|w a|)\n");
    }
  return 0;
}
