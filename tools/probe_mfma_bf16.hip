// Probe for the next kernel generation (NOTES_NEXT_ROUND.md): v_mfma_f32_32x32x16_bf16 on gfx950.
//   1. which k index does element e (0..7) of a lane's A / B operand carry?   (asymmetric integer matrices, exact)
//   2. fp32 products from bf16 pieces: error of 3 / 6 / 9 piece pairs against fp64, next to the fp32 MFMA (32x32x2)
//   3. issue rate: cycles per 32x32 x K=16 block with 6 bf16 MFMAs vs 8 fp32 MFMAs (one wave per SIMD, back to back)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_build/probe_mfma_bf16 tools/probe_mfma_bf16.hip     run: on the GPU box
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ unsigned short f2bf(float f) {  // round to nearest even
  unsigned u = __builtin_bit_cast(unsigned, f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }

// ---- 1. layout probe.  layout 0: k = 8*(lane>>5) + e;  layout 1: k = 4*(lane>>5) + (e&3) + 8*(e>>2)
__global__ void probe_layout(const float* A, const float* B, float* D, int layout) {  // A[32][16], B[16][32], D[32][32]
  const int lane = threadIdx.x, l31 = lane & 31, hh = lane >> 5;
  unsigned short a[8], b[8];
  for (int e = 0; e < 8; ++e) {
    const int k = layout == 0 ? 8 * hh + e : 4 * hh + (e & 3) + 8 * (e >> 2);
    a[e] = f2bf(A[l31 * 16 + k]);
    b[e] = f2bf(B[k * 32 + l31]);
  }
  bf16x8 av, bv;
  memcpy(&av, a, 16);
  memcpy(&bv, b, 16);
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + l31] = acc[r];
}

// ---- 2. accuracy: D = A[32][K] * B[K][32], K = 512, fp32 inputs.  mode 0: fp32 MFMA; 3/6/9: bf16 piece pairs
__device__ __forceinline__ void split3(float v, unsigned short* p) {
  p[0] = f2bf(v);
  float r = v - bf2f(p[0]);
  p[1] = f2bf(r);
  r -= bf2f(p[1]);
  p[2] = f2bf(r);
}

__global__ void gemm_probe(const float* A, const float* B, float* D, int K, int mode, int layout) {
  const int lane = threadIdx.x, l31 = lane & 31, hh = lane >> 5;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (mode == 0) {
    for (int k = 0; k < K; k += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + k + hh], B[(k + hh) * 32 + l31], acc, 0, 0, 0);
  } else {
    const int lim = mode == 9 ? 4 : (mode == 6 ? 2 : 1);
    for (int k0 = 0; k0 < K; k0 += 16) {
      unsigned short a[3][8], b[3][8];
      for (int e = 0; e < 8; ++e) {
        const int k = k0 + (layout == 0 ? 8 * hh + e : 4 * hh + (e & 3) + 8 * (e >> 2));
        unsigned short pa[3], pb[3];
        split3(A[l31 * K + k], pa);
        split3(B[k * 32 + l31], pb);
        for (int p = 0; p < 3; ++p) { a[p][e] = pa[p]; b[p][e] = pb[p]; }
      }
      // smallest terms first
      for (int s = lim; s >= 0; --s)
        for (int i = 0; i < 3; ++i) {
          const int j = s - i;
          if (j < 0 || j > 2) continue;
          bf16x8 av, bv;
          memcpy(&av, a[i], 16);
          memcpy(&bv, b[j], 16);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
        }
    }
  }
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + l31] = acc[r];
}

// ---- 3. issue rate, one wave per SIMD (256 threads), operands in registers
__global__ void __launch_bounds__(256) rate_probe(float* out, long long* cycles, int iters, int mode) {
  f32x16 acc[4];
  for (int m = 0; m < 4; ++m)
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  bf16x8 av, bv;
  unsigned short t[8];
  for (int e = 0; e < 8; ++e) t[e] = f2bf(1.0f + threadIdx.x * 0.001f + e);
  memcpy(&av, t, 16);
  memcpy(&bv, t, 16);
  const float fa = 1.0f + threadIdx.x * 0.001f, fb = 0.5f;
  const long long t0 = __builtin_readcyclecounter();
  if (mode == 0) {
    for (int it = 0; it < iters; ++it)  // 8 fp32 MFMAs = one 32x32 x K=16 block; four independent accumulators
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[q & 3], 0, 0, 0);
  } else {
    for (int it = 0; it < iters; ++it)  // 6 bf16 MFMAs = the same block from bf16 pieces
#pragma unroll
      for (int q = 0; q < 6; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[q & 3], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int m = 0; m < 4; ++m)
    for (int r = 0; r < 16; ++r) s += acc[m][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
  // 1. layout
  std::vector<float> A(32 * 16), B(16 * 32), D(32 * 32);
  for (int i = 0; i < 32; ++i)
    for (int k = 0; k < 16; ++k) A[i * 16 + k] = (float)((i * 7 + k * 3) % 11 - 5);
  for (int k = 0; k < 16; ++k)
    for (int j = 0; j < 32; ++j) B[k * 32 + j] = (float)((k * 5 + j * 2 + k * j) % 13 - 6);
  float *dA, *dB, *dD;
  CK(hipMalloc(&dA, 512 * 32 * 4)); CK(hipMalloc(&dB, 512 * 32 * 4)); CK(hipMalloc(&dD, 32 * 32 * 4));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  int good_layout = -1;
  for (int layout = 0; layout < 2; ++layout) {
    hipLaunchKernelGGL(probe_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD, layout);
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    double err = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double ref = 0;
        for (int k = 0; k < 16; ++k) ref += (double)A[i * 16 + k] * B[k * 32 + j];
        err = fmax(err, fabs(ref - D[i * 32 + j]));
      }
    printf("layout %d (%s): max |D - ref| = %g\n", layout,
           layout == 0 ? "k = 8*(lane>>5) + e" : "k = 4*(lane>>5) + (e&3) + 8*(e>>2)", err);
    if (err == 0 && good_layout < 0) good_layout = layout;
  }
  printf("NOTE: both layouts give the right PRODUCT whenever A and B use the SAME k permutation (a dot product does not care);\n"
         "      what matters for a kernel is only that A and B agree.  good_layout = %d\n", good_layout);
  // 2. accuracy
  const int K = 512;
  std::vector<float> A2(32 * K), B2(K * 32);
  srand(1);
  for (auto& v : A2) v = (float)(rand() % 20001) / 10000.f - 1.f;
  for (auto& v : B2) v = ((float)(rand() % 20001) / 10000.f - 1.f) / 22.6f;
  CK(hipMemcpy(dA, A2.data(), A2.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B2.data(), B2.size() * 4, hipMemcpyHostToDevice));
  std::vector<double> ref(32 * 32);
  double refmax = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += (double)A2[i * K + k] * B2[k * 32 + j];
      ref[i * 32 + j] = s;
      refmax = fmax(refmax, fabs(s));
    }
  const int modes[4] = {0, 3, 6, 9};
  for (int mi = 0; mi < 4; ++mi) {
    hipLaunchKernelGGL(gemm_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, modes[mi], good_layout < 0 ? 0 : good_layout);
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    double err = 0;
    for (int i = 0; i < 32 * 32; ++i) err = fmax(err, fabs(ref[i] - D[i]));
    printf("K=512 GEMM, %-22s max-norm relative error vs fp64: %.3e\n",
           modes[mi] == 0 ? "fp32 MFMA 32x32x2" : (modes[mi] == 3 ? "bf16 pieces, 3 pairs" : (modes[mi] == 6 ? "bf16 pieces, 6 pairs" : "bf16 pieces, 9 pairs")),
           err / refmax);
  }
  // 3. rate
  float* dO;
  long long* dC;
  CK(hipMalloc(&dO, 1024 * 256 * 4));
  CK(hipMalloc(&dC, 8));
  for (int mode = 0; mode < 2; ++mode) {
    const int iters = 2000;
    hipLaunchKernelGGL(rate_probe, dim3(256), dim3(256), 0, 0, dO, dC, iters, mode);  // warm-up
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(rate_probe, dim3(256), dim3(256), 0, 0, dO, dC, iters, mode);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    long long cyc = 0;
    CK(hipMemcpy(&cyc, dC, 8, hipMemcpyDeviceToHost));
    const double blocks = 256.0 * 4 * iters;  // 32x32 x K=16 blocks computed chip-wide (1 block per CU, 4 waves)
    const double flop = blocks * 2.0 * 32 * 32 * 16;
    printf("%s: %.3f ms, %.1f counter ticks per 32x32xK16 block per wave, %.1f TFLOP/s fp32-equivalent (256 blocks x 4 waves)\n",
           mode == 0 ? "8 x fp32 MFMA 32x32x2   " : "6 x bf16 MFMA 32x32x16  ", ms, (double)cyc / iters, flop / ms / 1e9);
  }
  return 0;
}
