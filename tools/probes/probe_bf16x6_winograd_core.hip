// The K loop of a Winograd F(4x4,3x3) convolution with six bf16 MFMAs per fp32 product, as a stand-alone experiment
// (round 4 -> plan of round 5, NOTES_NEXT_ROUND.md):
//     M[f][co][t] = sum_ci U[f][co][ci] * V[f][ci][t]          f = 36 frequencies, t = tiles
// V arrives TRANSFORMED in fp32 (what conv_wino4's transform role produces); a block = 64 co x 32 tiles x 36 frequencies,
// 12 waves, wave (j, s) owns frequencies i*6 + j (i = 0..5) of the 32-channel subtile s — the split of conv_wino4.hip.
// Per 16 input channels ("step"), SERIAL form:
//   phase T: every thread takes three (frequency, tile, 8-channel group) items: 8 fp32 loads, the exact three-way split by
//            truncation (x = x1 + x2 + x3, 8 mantissa bits each), three 16-byte LDS writes into Vp[piece][f][tile][16 ch];
//   phase M: per owned frequency 3 x ds_read_b128 (B pieces), 3 x 16-byte global loads (A pieces of U, pre-split and
//            stored MFMA-ready), 6 x v_mfma_f32_32x32x16_bf16 (a1b3 a2b2 a3b1 a1b2 a2b1 a1b1).
// Checks M against fp64 on sampled outputs and prints the time per step.  The fp32-MFMA form of the same loop (conv_wino4's
// K loop without its transform) costs 2 x 4 608 = 9 216 SIMD cycles per step = 3.9 us at 2.35 GHz.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ab/probe_b6core tools/probes/probe_bf16x6_winograd_core.hip && tools/ab/probe_b6core
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

#define NF 36
#define TCO 64
#define TT 32

// Up: [piece 3][f 36][step Ci/16][co-subtile Co/32][lane 64] x 16 bytes: lane (m = lane & 31, kg = lane >> 5) holds
// U[f][co = 32 sub + m][ci = 16 step + 8 kg + 0..7] as 8 bf16
__global__ void __launch_bounds__(768, 1) wino_core_b6(const float* __restrict__ V, const u32x4_t* __restrict__ Up,
                                                       float* __restrict__ M, int Ci, int Co, int T) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4_t* Vp = reinterpret_cast<u32x4_t*>(smem);  // [3][36][32 tiles][2 kg] x 16 bytes = 110 592 bytes
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wj = wave % 6, ws = wave / 6;
  const int l31 = lane & 31, kg = lane >> 5;
  const int n_co_tiles = Co / TCO;
  const int co_tile = blockIdx.x % n_co_tiles, tt = blockIdx.x / n_co_tiles;
  const int t0 = tt * TT, sub = co_tile * 2 + ws;
  const int nsteps = Ci / 16, nsub = Co / 32;
  const size_t up_piece = (size_t)NF * nsteps * nsub * 64;
  f32x16 acc[6] = {};
  for (int st = 0; st < nsteps; ++st) {
    // ---- phase T: items (f, tile, kgroup): 36 * 32 * 2 = 2304 = 3 per thread
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int item = tid + q * 768;
      const int tile = item & 31, g = (item >> 5) & 1, f = item >> 6;
      const float* src = V + ((size_t)f * Ci + st * 16 + g * 8) * T + t0 + tile;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = src[(size_t)e * T];
      unsigned a1[8], a2[8], a3[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned u1 = __builtin_bit_cast(unsigned, v[e]) & 0xffff0000u;
        const float r1 = v[e] - __builtin_bit_cast(float, u1);
        const unsigned u2 = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
        const float r2 = r1 - __builtin_bit_cast(float, u2);
        a1[e] = u1; a2[e] = u2; a3[e] = __builtin_bit_cast(unsigned, r2);
      }
      u32x4_t p1, p2, p3;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        p1[e] = (a1[2 * e] >> 16) | (a1[2 * e + 1] & 0xffff0000u);
        p2[e] = (a2[2 * e] >> 16) | (a2[2 * e + 1] & 0xffff0000u);
        p3[e] = (a3[2 * e] >> 16) | (a3[2 * e + 1] & 0xffff0000u);
      }
      // LDS slot of (f, tile, g): the two kgroups of a tile are 16 bytes apart... a ds_read_b128 of a half-wave reads 32
      // consecutive tiles of one kgroup: keep [kg][tile] so that those 32 x 16 bytes are contiguous (conflict-free)
      const int slot = (f * 2 + g) * 32 + tile;
      Vp[slot] = p1;
      Vp[NF * 64 + slot] = p2;
      Vp[2 * NF * 64 + slot] = p3;
    }
    __syncthreads();
    // ---- phase M
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int f = i * 6 + wj;
      const size_t ua = ((size_t)(f * nsteps + st) * nsub + sub) * 64 + lane;
      const u32x4_t a1 = Up[ua], a2 = Up[up_piece + ua], a3 = Up[2 * up_piece + ua];
      const int slot = (f * 2 + kg) * 32 + l31;
      const u32x4_t b1 = Vp[slot], b2 = Vp[NF * 64 + slot], b3 = Vp[2 * NF * 64 + slot];
#define MF(A, B) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A), __builtin_bit_cast(bf16x8_t, B), acc[i], 0, 0, 0);
      MF(a1, b3) MF(a2, b2) MF(a3, b1) MF(a1, b2) MF(a2, b1) MF(a1, b1)
#undef MF
    }
    __syncthreads();
  }
  // acc[i][r]: co = 32 sub + (r & 3) + 8 (r >> 2) + 4 kg, tile = t0 + l31
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = 32 * sub + (r & 3) + 8 * (r >> 2) + 4 * kg;
      M[((size_t)(i * 6 + wj) * Co + co) * T + t0 + l31] = acc[i][r];
    }
}

static float trunc16(float v) { uint32_t u; memcpy(&u, &v, 4); u &= 0xffff0000u; float r; memcpy(&r, &u, 4); return r; }
static uint16_t hi16(float v) { uint32_t u; memcpy(&u, &v, 4); return (uint16_t)(u >> 16); }

int main(int argc, char** argv) {
  const int Ci = argc > 1 ? atoi(argv[1]) : 256, Co = argc > 2 ? atoi(argv[2]) : 256, T = argc > 3 ? atoi(argv[3]) : 32 * 1024;
  printf("Ci %d Co %d tiles %d (= %d images of 64x64)\n", Ci, Co, T, T / 256);
  std::vector<float> V((size_t)NF * Ci * T), U((size_t)NF * Co * Ci);
  srand(2);
  auto rnd = []() { float s = 0; for (int i = 0; i < 4; ++i) s += rand() / (float)RAND_MAX - 0.5f; return s * 1.7f; };
  for (auto& v : V) v = rnd();
  for (auto& v : U) v = rnd() * 0.1f;
  const int nsteps = Ci / 16, nsub = Co / 32;
  const size_t up_piece = (size_t)NF * nsteps * nsub * 64 * 8;
  std::vector<uint16_t> Up(3 * up_piece);
  for (int f = 0; f < NF; ++f) for (int st = 0; st < nsteps; ++st) for (int sb = 0; sb < nsub; ++sb) for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) {
    const float v = U[((size_t)f * Co + 32 * sb + (l & 31)) * Ci + 16 * st + 8 * (l >> 5) + e];
    const float x1 = trunc16(v), r1 = v - x1, x2 = trunc16(r1), r2 = r1 - x2;
    const size_t idx = ((((size_t)f * nsteps + st) * nsub + sb) * 64 + l) * 8 + e;
    Up[idx] = hi16(x1); Up[up_piece + idx] = hi16(x2); Up[2 * up_piece + idx] = hi16(r2);
  }
  float *dV, *dM; uint16_t* dUp;
  hipMalloc(&dV, V.size() * 4); hipMalloc(&dM, (size_t)NF * Co * T * 4); hipMalloc(&dUp, Up.size() * 2);
  hipMemcpy(dV, V.data(), V.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dUp, Up.data(), Up.size() * 2, hipMemcpyHostToDevice);
  const size_t lds = 3 * NF * 64 * 16;
  hipFuncSetAttribute((const void*)wino_core_b6, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int grid = (T / TT) * (Co / TCO);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(wino_core_b6, dim3(grid), dim3(768), lds, 0, dV, (const u32x4_t*)dUp, dM, Ci, Co, T);
  hipError_t err = hipDeviceSynchronize();
  if (err != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(err)); return 1; }
  hipEventRecord(e0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(wino_core_b6, dim3(grid), dim3(768), lds, 0, dV, (const u32x4_t*)dUp, dM, Ci, Co, T);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  const double items_per_cu = (double)grid / 256.0;
  printf("%.3f ms per launch; %d work items (%.1f per CU); %.2f us per 16-channel step and item (fp32-MFMA form: 3.9 us); "
         "%.1f TF/s of fp32-equivalent multiply-adds\n", ms, grid, items_per_cu, ms * 1e3 / items_per_cu / nsteps,
         2.0 * NF * Co * (double)Ci * T / ms / 1e9);
  std::vector<float> Mh((size_t)NF * Co * T);
  hipMemcpy(Mh.data(), dM, Mh.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0, sumsq = 0; long cnt = 0;
  for (int s = 0; s < 4000; ++s) {
    const int f = rand() % NF, co = rand() % Co, t = rand() % T;
    double ref = 0, mag = 0;
    for (int c = 0; c < Ci; ++c) { const double p = (double)U[((size_t)f * Co + co) * Ci + c] * V[((size_t)f * Ci + c) * T + t]; ref += p; mag += fabs(p); }
    const double e = fabs(Mh[((size_t)f * Co + co) * T + t] - ref) / mag;
    worst = e > worst ? e : worst; sumsq += e * e; ++cnt;
  }
  printf("error vs fp64 over %ld sampled outputs, relative to sum|uv|: max %.2e rms %.2e\n", cnt, worst, sqrt(sumsq / cnt));
  return 0;
}
