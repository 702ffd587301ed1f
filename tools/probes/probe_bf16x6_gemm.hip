// Can six bf16 MFMAs stand in for one fp32 product?  x = x1 + x2 + x3 (three 8-bit-mantissa pieces, EXACT by truncation),
// a*b ~ a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1): the dropped terms are <= 2^-24 |a||b|.  This probe measures, on one
// MI355X, (1) the error of that scheme against fp64 next to the fp32 MFMA's and to the 3- / 1-product forms, (2) the rate of
// a register-fed 64x64-per-wave GEMM with v_mfma_f32_32x32x16_bf16 x 6 against v_mfma_f32_32x32x2_f32, operands pre-split
// (the Winograd kernels would split ONCE per block into LDS) and with B split on the fly in the consuming wave.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ab/probe_bf16x6 tools/probes/probe_bf16x6_gemm.hip && tools/ab/probe_bf16x6
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

static const int M = 2048, N = 2048, K = 1024;

// ---- fp32 MFMA: A4[tm][k8][lane] = float4 {A[32tm + (lane&31)][8k8 + 2s + (lane>>5)], s = 0..3}; B4 likewise with n
__global__ void __launch_bounds__(256) gemm_f32(const float4* __restrict__ A4, const float4* __restrict__ B4,
                                                float* __restrict__ C, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int w = blockIdx.x * 4 + wave;  // 64x64 tile index
  const int tm = (w / (N / 64)) * 2, tn = (w % (N / 64)) * 2;
  f32x16 acc[2][2] = {};
  for (int rep = 0; rep < reps; ++rep)
    for (int k8 = 0; k8 < K / 8; ++k8) {
      float4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = A4[((size_t)(tm + i) * (K / 8) + k8) * 64 + lane];
        b[i] = B4[((size_t)(tn + i) * (K / 8) + k8) * 64 + lane];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) C[(((size_t)w * 4 + i * 2 + j) * 16 + r) * 64 + lane] = acc[i][j][r];
}

// ---- bf16 pieces: P[piece][t][ks][lane] = 8 bf16 of X[32t + (lane&31)][16ks + 8(lane>>5) + 0..7]
// NPROD = 1, 3, 6 products;  FLY: B arrives as fp32 (BF[tn][ks][lane][2] float4) and is split here
__device__ __forceinline__ void split8(const float4 lo, const float4 hi, u32x4_t* p1, u32x4_t* p2, u32x4_t* p3) {
  const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  unsigned a1[8], a2[8], a3[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned u1 = __builtin_bit_cast(unsigned, v[i]) & 0xffff0000u;
    const float r1 = v[i] - __builtin_bit_cast(float, u1);
    const unsigned u2 = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
    const float r2 = r1 - __builtin_bit_cast(float, u2);
    a1[i] = u1; a2[i] = u2; a3[i] = __builtin_bit_cast(unsigned, r2);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    (*p1)[i] = (a1[2 * i] >> 16) | (a1[2 * i + 1] & 0xffff0000u);
    (*p2)[i] = (a2[2 * i] >> 16) | (a2[2 * i + 1] & 0xffff0000u);
    (*p3)[i] = (a3[2 * i] >> 16) | (a3[2 * i + 1] & 0xffff0000u);
  }
}

template <int NPROD, bool FLY>
__global__ void __launch_bounds__(256) gemm_bf16s(const u32x4_t* __restrict__ AP, const u32x4_t* __restrict__ BP,
                                                  const float4* __restrict__ BF, float* __restrict__ C, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int w = blockIdx.x * 4 + wave;
  const int tm = (w / (N / 64)) * 2, tn = (w % (N / 64)) * 2;
  const size_t pa = (size_t)(M / 32) * (K / 16) * 64, pb = (size_t)(N / 32) * (K / 16) * 64;
  constexpr int NP = NPROD == 1 ? 1 : (NPROD == 3 ? 2 : 3);  // pieces used
  f32x16 acc[2][2] = {};
  for (int rep = 0; rep < reps; ++rep)
    for (int ks = 0; ks < K / 16; ++ks) {
      u32x4_t a[2][3], b[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int p = 0; p < NP; ++p) a[i][p] = AP[p * pa + ((size_t)(tm + i) * (K / 16) + ks) * 64 + lane];
        if (FLY) {
          const size_t o = (((size_t)(tn + i) * (K / 16) + ks) * 64 + lane) * 2;
          split8(BF[o], BF[o + 1], &b[i][0], &b[i][1], &b[i][2]);
        } else {
#pragma unroll
          for (int p = 0; p < NP; ++p) b[i][p] = BP[p * pb + ((size_t)(tn + i) * (K / 16) + ks) * 64 + lane];
        }
      }
#define MF(I, J, PA, PB)                                                                                           \
  acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[I][PA]),                       \
                                                      __builtin_bit_cast(bf16x8_t, b[J][PB]), acc[I][J], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (NPROD == 6) { MF(i, j, 0, 2) MF(i, j, 1, 1) MF(i, j, 2, 0) }  // smallest terms first
          if (NPROD >= 3) { MF(i, j, 0, 1) MF(i, j, 1, 0) }
          MF(i, j, 0, 0)
        }
#undef MF
    }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) C[(((size_t)w * 4 + i * 2 + j) * 16 + r) * 64 + lane] = acc[i][j][r];
}

static float trunc16(float v) { uint32_t u; memcpy(&u, &v, 4); u &= 0xffff0000u; float r; memcpy(&r, &u, 4); return r; }
static uint16_t hi16(float v) { uint32_t u; memcpy(&u, &v, 4); return (uint16_t)(u >> 16); }

int main() {
  std::vector<float> A((size_t)M * K), B((size_t)K * N);  // A[m][k], B[k][n]
  srand(1);
  auto rnd = []() { float s = 0; for (int i = 0; i < 6; ++i) s += rand() / (float)RAND_MAX - 0.5f; return s * 1.4f; };
  for (auto& v : A) v = rnd();
  for (auto& v : B) v = rnd();
  // fp32 layouts
  std::vector<float> A4((size_t)M * K), B4((size_t)N * K);
  for (int t = 0; t < M / 32; ++t) for (int k8 = 0; k8 < K / 8; ++k8) for (int l = 0; l < 64; ++l) for (int s = 0; s < 4; ++s)
    A4[(((size_t)t * (K / 8) + k8) * 64 + l) * 4 + s] = A[(size_t)(32 * t + (l & 31)) * K + 8 * k8 + 2 * s + (l >> 5)];
  for (int t = 0; t < N / 32; ++t) for (int k8 = 0; k8 < K / 8; ++k8) for (int l = 0; l < 64; ++l) for (int s = 0; s < 4; ++s)
    B4[(((size_t)t * (K / 8) + k8) * 64 + l) * 4 + s] = B[(size_t)(8 * k8 + 2 * s + (l >> 5)) * N + 32 * t + (l & 31)];
  // bf16 pieces + fp32 B in the bf16 lane layout
  const size_t pa = (size_t)(M / 32) * (K / 16) * 64 * 8, pb = (size_t)(N / 32) * (K / 16) * 64 * 8;
  std::vector<uint16_t> AP(3 * pa), BP(3 * pb);
  std::vector<float> BF(pb);
  auto split = [](float v, uint16_t* o) { float x1 = trunc16(v), r1 = v - x1, x2 = trunc16(r1), r2 = r1 - x2; o[0] = hi16(x1); o[1] = hi16(x2); o[2] = hi16(r2); };
  for (int t = 0; t < M / 32; ++t) for (int ks = 0; ks < K / 16; ++ks) for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) {
    uint16_t o[3]; split(A[(size_t)(32 * t + (l & 31)) * K + 16 * ks + 8 * (l >> 5) + e], o);
    const size_t idx = (((size_t)t * (K / 16) + ks) * 64 + l) * 8 + e;
    for (int p = 0; p < 3; ++p) AP[p * pa + idx] = o[p];
  }
  for (int t = 0; t < N / 32; ++t) for (int ks = 0; ks < K / 16; ++ks) for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) {
    const float v = B[(size_t)(16 * ks + 8 * (l >> 5) + e) * N + 32 * t + (l & 31)];
    uint16_t o[3]; split(v, o);
    const size_t idx = (((size_t)t * (K / 16) + ks) * 64 + l) * 8 + e;
    for (int p = 0; p < 3; ++p) BP[p * pb + idx] = o[p];
    BF[idx] = v;
  }
  float *dA4, *dB4, *dBF, *dC; uint16_t *dAP, *dBP;
  hipMalloc(&dA4, A4.size() * 4); hipMalloc(&dB4, B4.size() * 4); hipMalloc(&dBF, BF.size() * 4); hipMalloc(&dC, (size_t)M * N * 4);
  hipMalloc(&dAP, AP.size() * 2); hipMalloc(&dBP, BP.size() * 2);
  hipMemcpy(dA4, A4.data(), A4.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB4, B4.data(), B4.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dBF, BF.data(), BF.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dAP, AP.data(), AP.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dBP, BP.data(), BP.size() * 2, hipMemcpyHostToDevice);
  const int grid = (M / 64) * (N / 64) / 4;
  // fp64 reference of a sample of outputs: the first 64x64 tile (w = 0) and the last one
  std::vector<float> Ch((size_t)M * N);
  auto check = [&](const char* name, double ms, int reps) {
    hipMemcpy(Ch.data(), dC, (size_t)M * N * 4, hipMemcpyDeviceToHost);
    double worst = 0, sumsq = 0; long cnt = 0;
    for (int w : {0, grid * 4 - 1}) {
      const int tm = (w / (N / 64)) * 2, tn = (w % (N / 64)) * 2;
      for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) for (int l = 0; l < 64; ++l) {
        const int m = 32 * (tm + i) + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), n = 32 * (tn + j) + (l & 31);
        double ref = 0, mag = 0;
        for (int k = 0; k < K; ++k) { const double p = (double)A[(size_t)m * K + k] * B[(size_t)k * N + n]; ref += p; mag += fabs(p); }
        const double got = Ch[(((size_t)w * 4 + i * 2 + j) * 16 + r) * 64 + l] / (double)reps;
        const double e = fabs(got - ref) / mag;
        worst = e > worst ? e : worst; sumsq += e * e; ++cnt;
      }
    }
    printf("%-34s %8.3f ms %8.1f TF/s (2MNK)   err/sum|ab|: max %.2e rms %.2e\n", name, ms / reps, 2.0 * M * N * K * reps / ms / 1e9,
           worst, sqrt(sumsq / cnt));
  };
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
#define RUN(NAME, LAUNCH)                                                     \
  { const int reps = 1; LAUNCH; hipDeviceSynchronize();                       \
    hipEventRecord(e0); LAUNCH; hipEventRecord(e1); hipEventSynchronize(e1);  \
    hipEventElapsedTime(&ms, e0, e1); check(NAME, ms, 1); }
  RUN("fp32 MFMA 32x32x2", hipLaunchKernelGGL(gemm_f32, dim3(grid), dim3(256), 0, 0, (const float4*)dA4, (const float4*)dB4, dC, reps))
  RUN("bf16 x1 (a1 b1)", hipLaunchKernelGGL((gemm_bf16s<1, false>), dim3(grid), dim3(256), 0, 0, (const u32x4_t*)dAP, (const u32x4_t*)dBP, (const float4*)dBF, dC, reps))
  RUN("bf16 x3 (+ a1b2 + a2b1)", hipLaunchKernelGGL((gemm_bf16s<3, false>), dim3(grid), dim3(256), 0, 0, (const u32x4_t*)dAP, (const u32x4_t*)dBP, (const float4*)dBF, dC, reps))
  RUN("bf16 x6 pre-split", hipLaunchKernelGGL((gemm_bf16s<6, false>), dim3(grid), dim3(256), 0, 0, (const u32x4_t*)dAP, (const u32x4_t*)dBP, (const float4*)dBF, dC, reps))
  RUN("bf16 x6, B split in the wave", hipLaunchKernelGGL((gemm_bf16s<6, true>), dim3(grid), dim3(256), 0, 0, (const u32x4_t*)dAP, (const u32x4_t*)dBP, (const float4*)dBF, dC, reps))
  return 0;
}
