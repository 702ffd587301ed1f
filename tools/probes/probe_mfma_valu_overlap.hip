// Does ordinary VALU work hide under matrix instructions on gfx950?  Per loop iteration a wave issues 4 independent MFMAs and
// NV independent v_fma_f32; 3 waves per SIMD (768-thread blocks, one per CU).  Reported: SIMD cycles per iteration at the
// measured wall time and an assumed 2.35 GHz, next to the two models  ADD = 3 x (4 x T_mfma + 4 NV)  and
// OVERLAP = max(3 x 4 x T_mfma, 3 x (4 + NV) x 4)   (T = 64 cycles for v_mfma_f32_32x32x2_f32, 32 for v_mfma_f32_32x32x16_bf16).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ab/probe_overlap tools/probes/probe_mfma_valu_overlap.hip && tools/ab/probe_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <bool BF16, int NV>
__global__ void __launch_bounds__(768, 1) k(float* out, int iters, float seed) {
  f32x16 acc[4] = {};
  float v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = seed + i + threadIdx.x;
  const float a = seed, b = seed * 0.5f;
  bf16x8_t ab, bb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(seed + i); bb[i] = (__bf16)(seed - i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (BF16) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[m], 0, 0, 0);
      else acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NV / 4; ++q) {
        const int i = (m * (NV / 4) + q) & 31;
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += v[i];
#pragma unroll
  for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][7];
  if (s == 12345.678f) out[0] = s;
}

template <bool BF16, int NV>
static void run(float* d) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<BF16, NV>), dim3(256), dim3(768), 0, 0, d, 100, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<BF16, NV>), dim3(256), dim3(768), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double cyc = ms * 1e-3 * 2.35e9 / iters;
  const int T = BF16 ? 32 : 64;
  const double add = 3.0 * (4 * T + 4 * NV), ovl = 3.0 * (4 * T) > 3.0 * (4 + NV) * 4 ? 3.0 * (4 * T) : 3.0 * (4 + NV) * 4;
  printf("%-5s 4 MFMA + %2d VALU per wave-iteration: %7.0f cycles per SIMD-iteration   (ADD model %5.0f, OVERLAP model %5.0f)\n",
         BF16 ? "bf16" : "fp32", NV, cyc, add, ovl);
}

int main() {
  float* d; hipMalloc(&d, 4);
  run<false, 0>(d); run<false, 8>(d); run<false, 16>(d); run<false, 32>(d); run<false, 64>(d);
  run<true, 0>(d); run<true, 8>(d); run<true, 16>(d); run<true, 24>(d); run<true, 32>(d); run<true, 64>(d);
  return 0;
}
