// Dev tool: what v_mfma_f32_16x16x32_f16 sustains on this part with nothing else in the loop -- the practical ceiling the
// conv kernels' "fraction of 2.5 PFLOP/s" should be read against.  Every wave keeps 8 independent accumulator tiles and
// issues MFMAs back to back from registers (no LDS, no memory); operands are either zeros or random f16 (the clock the
// chip holds under matrix load depends on the data toggling).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) mfma_loop(const f16x8* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  f16x8 a[2], b[4];
  for (int i = 0; i < 2; ++i) a[i] = in[(tid * 6 + i) & 4095];
  for (int i = 0; i < 4; ++i) b[i] = in[(tid * 6 + 2 + i) & 4095];
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[tid] = s;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) mfma32_loop(const f16x8* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  f16x8 a[2], b[2];
  for (int i = 0; i < 2; ++i) a[i] = in[(tid * 6 + i) & 4095];
  for (int i = 0; i < 2; ++i) b[i] = in[(tid * 6 + 2 + i) & 4095];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i * 2 + j], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[tid] = s;
}

int main() {
  int dev = 0;
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, dev);
  const int cus = prop.multiProcessorCount;
  std::vector<f16> h(4096 * 8);
  f16x8* din;
  float* dout;
  (void)hipMalloc(&din, h.size() * sizeof(f16));
  const int iters = 40000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    srand(1);
    for (auto& v : h) v = mode ? (f16)((rand() % 2001 - 1000) / 4000.0f) : (f16)0.f;
    (void)hipMemcpy(din, h.data(), h.size() * sizeof(f16), hipMemcpyHostToDevice);
    for (int wps = 1; wps <= 8; wps *= 2) {   // waves per SIMD
      const int blocks = cus * wps;        // 256 threads = 4 waves = one per SIMD
      (void)hipMalloc(&dout, (size_t)blocks * 256 * sizeof(float));
      hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, din, dout, 2000);   // warm-up / clock ramp
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)blocks * 4 * iters * 8 * (2.0 * 16 * 16 * 32);
      const double per_mfma_ns = ms * 1e6 / ((double)iters * 8 * wps);
      printf("%s operands, %d wave(s) per SIMD, %d CUs: %.1f ms, %.0f TFLOP/s, %.2f ns per MFMA per SIMD\n",
             mode ? "random" : "zero", wps, cus, ms, flops / ms / 1e9, per_mfma_ns);
      // the 32x32x16 form: 4 independent accumulator tiles, 4 MFMAs (2 * 32*32*16 FLOP each) per iteration
      hipLaunchKernelGGL(mfma32_loop, dim3(blocks), dim3(256), 0, 0, din, dout, 2000);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(mfma32_loop, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double flops32 = (double)blocks * 4 * iters * 4 * (2.0 * 32 * 32 * 16);
      printf("   32x32x16: %.1f ms, %.0f TFLOP/s, %.2f ns per MFMA per SIMD\n", ms, flops32 / ms / 1e9,
             ms * 1e6 / ((double)iters * 4 * wps));
      (void)hipFree(dout);
    }
  }
  return 0;
}
