// Dev tool: do vector-ALU instructions of one wave overlap the MFMAs of ANOTHER wave on the same SIMD?  512-thread workgroups
// (two waves per SIMD, one workgroup per CU), per loop trip either 16 x v_mfma_f32_16x16x32_f16 (role M, ~256 matrix-pipe
// cycles) or 64 x v_pk_fma_f32 on independent accumulators (role V, ~256 VALU cycles).  Configurations: every wave M, every
// wave V, and waves 0-3 M / 4-7 V (a workgroup's waves go round the four SIMDs, so every SIMD gets one of each).  If the two
// pipes overlap, the mixed run takes max(M, V) of ONE wave's work; if they share the issue port, the sum.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_mfma_probe.hip -o /tmp/vmp && /tmp/vmp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// roles: bit 0 = waves 0-3 run M (else V), bit 1 = waves 4-7 run M (else V)
__global__ void __launch_bounds__(512) probe(const f16x8* __restrict__ in, float* __restrict__ out, int iters, int roles) {
  const int tid = blockIdx.x * 512 + threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool mrole = (roles >> (wave >> 2)) & 1;
  float s = 0.f;
  if (mrole) {
    f16x8 a[2], b[4];
    for (int i = 0; i < 2; ++i) a[i] = in[(tid * 6 + i) & 4095];
    for (int i = 0; i < 4; ++i) b[i] = in[(tid * 6 + 2 + i) & 4095];
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + r) & 1], b[i & 3], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    f32x2 v[16], m, c;
    const f16x8 x = in[tid & 4095];
    m = (f32x2){1.0f + 1e-7f * (float)x[0], 1.0f - 1e-7f * (float)x[1]};
    c = (f32x2){1e-3f * (float)x[2], 1e-3f * (float)x[3]};
    for (int i = 0; i < 16; ++i) v[i] = (f32x2){(float)x[i & 7], (float)x[(i + 1) & 7]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(c));
    }
    for (int i = 0; i < 16; ++i) s += v[i][0] + v[i][1];
  }
  out[tid] = s;
}

int main() {
  const int ncu = 256, iters = 20000;
  std::vector<f16x8> h(4096);
  srand(1);
  for (auto& v : h) for (int e = 0; e < 8; ++e) v[e] = (f16)((rand() % 2001 - 1000) / 1000.0f);
  f16x8* din; float* dout;
  hipMalloc(&din, 4096 * sizeof(f16x8)); hipMalloc(&dout, ncu * 512 * sizeof(float));
  hipMemcpy(din, h.data(), 4096 * sizeof(f16x8), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[4] = {"all V (2 V waves per SIMD)", "waves 0-3 M, 4-7 V", "waves 0-3 V, 4-7 M", "all M (2 M waves per SIMD)"};
  for (int warm = 0; warm < 2; ++warm)
    for (int roles = 0; roles < 4; ++roles) {
      hipLaunchKernelGGL(probe, dim3(ncu), dim3(512), 0, 0, din, dout, iters, roles);   // warm-up
      hipEventRecord(e0);
      for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe, dim3(ncu), dim3(512), 0, 0, din, dout, iters, roles);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (warm) printf("%-32s %8.3f ms per launch  = %6.1f ns per loop trip\n", names[roles], ms / 5, ms / 5 * 1e6 / iters);
    }
  return 0;
}
