// Second experiment on the six-product bf16 K loop of F(4x4,3x3) (see probe_bf16x6_winograd_core.hip for the first, serial
// one: correct, but 9.5 us per 16-channel step — every load exposed).  Here ONE FREQUENCY-COLUMN PAIR per work item
// ("pass" p: columns (1,2), (3,4) or (0,5): 12 of the 36 frequencies), the form NOTES_NEXT_ROUND.md proposes for the LDS
// capacity problem:
//   * block = 64 co x 32 tiles x 12 frequencies, 12 waves; wave w owns frequency w of the pass for BOTH 32-channel
//     subtiles: 2 accumulators (32 registers), one B operand serves two MFMA groups;
//   * per 16 input channels ("step"): a thread splits ONE (frequency, tile, 8-channel group) item into three bf16 pieces
//     (8 loads, ~55 VALU, three 16-byte LDS writes) — V pieces of a step are 37 KB, double-buffered; a wave reads
//     3 x 16 bytes of B per lane, 6 x 16 bytes of A (U pieces, pre-split, MFMA-ready in global memory) and issues 12 MFMAs;
//   * loads run two steps ahead (3-slot rings in registers), one barrier per step.
// Prints time per step and the error against fp64.  Budget: 12 MFMAs x 32 cycles x 3 waves = 1 152 SIMD cycles = 0.49 us per
// step and pass, 1.47 us for the three passes of a 16-channel step — against 3.9 us (model) / 5.1 us (measured) of the fp32 form.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ab/probe_b6pass tools/probes/probe_bf16x6_winograd_pass.hip && tools/ab/probe_b6pass
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
#ifndef HOT
#define HOT 0  // 1: every step re-reads the operands of steps 0 / 1 (results wrong; isolates the loop from HBM / L2 misses)
#endif
#ifndef RING
#define RING 2  // register ring depth: loads run RING - 1 steps ahead
#endif

__device__ __forceinline__ void split8(const float* v, u32x4_t* p1, u32x4_t* p2, u32x4_t* p3) {
  unsigned a1[8], a2[8], a3[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const unsigned u1 = __builtin_bit_cast(unsigned, v[e]) & 0xffff0000u;
    const float r1 = v[e] - __builtin_bit_cast(float, u1);
    const unsigned u2 = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
    const float r2 = r1 - __builtin_bit_cast(float, u2);
    a1[e] = u1; a2[e] = u2; a3[e] = __builtin_bit_cast(unsigned, r2);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    (*p1)[e] = (a1[2 * e] >> 16) | (a1[2 * e + 1] & 0xffff0000u);
    (*p2)[e] = (a2[2 * e] >> 16) | (a2[2 * e + 1] & 0xffff0000u);
    (*p3)[e] = (a3[2 * e] >> 16) | (a3[2 * e + 1] & 0xffff0000u);
  }
}

// Up: [piece 3][f 36][step][co-subtile][lane] x 16 bytes (as in probe_bf16x6_winograd_core.hip)
__global__ void __launch_bounds__(768, 1) wino_pass_b6(const float* __restrict__ V, const u32x4_t* __restrict__ Up,
                                                       float* __restrict__ M, int Ci, int Co, int T) {
  __shared__ __attribute__((aligned(16))) u32x4_t Vp[2][3][12 * 64];  // [buffer][piece][(freq, kgroup, tile)]: 73 728 bytes
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kg = lane >> 5;
  const int n_co_tiles = Co / 64;
  int b = blockIdx.x;
  const int pass = b % 3; b /= 3;
  const int co_tile = b % n_co_tiles, tt = b / n_co_tiles;
  const int t0 = tt * 32;
  const int nsteps = Ci / 16, nsub = Co / 32;
  const size_t up_piece = (size_t)NF * nsteps * nsub * 64;
  // frequency of wave w: row i = w % 6, column = the pass's pair member w / 6
  const int jcol = pass == 0 ? (wave / 6 ? 2 : 1) : (pass == 1 ? (wave / 6 ? 4 : 3) : (wave / 6 ? 5 : 0));
  const int f_own = (wave % 6) * 6 + jcol;
  // transform item of this thread: (frequency slot fs = tid / 64, kgroup g, tile)
  const int it_tile = tid & 31, it_g = (tid >> 5) & 1, it_fs = tid >> 6;
  const int it_j = pass == 0 ? (it_fs / 6 ? 2 : 1) : (pass == 1 ? (it_fs / 6 ? 4 : 3) : (it_fs / 6 ? 5 : 0));
  const int it_f = (it_fs % 6) * 6 + it_j;
  const float* vsrc = V + ((size_t)it_f * Ci + it_g * 8) * T + t0 + it_tile;
  const int it_slot = (it_fs * 2 + it_g) * 32 + it_tile;
  const int rd_slot = (wave * 2 + kg) * 32 + l31;

  f32x16 acc[2] = {};
  float v[RING][8];
  u32x4_t A[RING][2][3];
#define LOADV(ST, S)                                                                       \
  {                                                                                        \
    const float* p_ = vsrc + (size_t)(HOT ? 0 : (ST)) * 16 * T;  /* HOT: operands stay cache-hot (timing only) */ \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) v[S][e] = p_[(size_t)e * T];             \
  }
#define LOADA(ST, S)                                                                       \
  {                                                                                        \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                     \
      const size_t ua = ((size_t)(f_own * nsteps + (HOT ? ((ST) & 1) : (ST))) * nsub + co_tile * 2 + s2) * 64 + lane; \
      A[S][s2][0] = Up[ua]; A[S][s2][1] = Up[up_piece + ua]; A[S][s2][2] = Up[2 * up_piece + ua]; \
    }                                                                                      \
  }
#define TRANSFORM(S, BUF)                                                                  \
  {                                                                                        \
    u32x4_t p1, p2, p3;                                                                    \
    split8(v[S], &p1, &p2, &p3);                                                           \
    Vp[BUF][0][it_slot] = p1; Vp[BUF][1][it_slot] = p2; Vp[BUF][2][it_slot] = p3;          \
  }
#define MF(AA, BB, S2) acc[S2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, AA), __builtin_bit_cast(bf16x8_t, BB), acc[S2], 0, 0, 0);
#define MMA(S, BUF)                                                                        \
  {                                                                                        \
    const u32x4_t b1 = Vp[BUF][0][rd_slot], b2 = Vp[BUF][1][rd_slot], b3 = Vp[BUF][2][rd_slot]; \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                     \
      MF(A[S][s2][0], b3, s2) MF(A[S][s2][1], b2, s2) MF(A[S][s2][2], b1, s2)              \
      MF(A[S][s2][0], b2, s2) MF(A[S][s2][1], b1, s2) MF(A[S][s2][0], b1, s2)              \
    }                                                                                      \
  }
  // prologue: the first RING - 1 steps in flight, step 0 transformed
  LOADV(0, 0) LOADA(0, 0)
#if RING == 3
  if (nsteps > 1) { LOADV(1, 1) LOADA(1, 1) }
#endif
  TRANSFORM(0, 0)
  __syncthreads();
  // main loop, unrolled so that ring slots (mod RING) and LDS buffers (mod 2) are compile-time; nsteps % (2 * RING... 6) == 0
  // is NOT required: loads and transforms past the end are clamped repeats of the last step (no conditional definitions of
  // register arrays: hipcc carries conditionally defined arrays as undefined values through every join and spills them)
  constexpr int UNR = RING == 2 ? 2 : 6;
  for (int st0 = 0; st0 < nsteps; st0 += UNR) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int st = st0 + u;
      const int ld = st + RING - 1 < nsteps ? st + RING - 1 : nsteps - 1;
      LOADV(ld, (u + RING - 1) % RING) LOADA(ld, (u + RING - 1) % RING)
      if (st < nsteps) MMA(u % RING, u % 2)
      TRANSFORM((u + 1) % RING, (u + 1) % 2)
      // (LDS-only barrier: __syncthreads() also waits vmcnt(0), i.e. for the loads just issued two steps ahead)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = 64 * co_tile + 32 * s2 + (r & 3) + 8 * (r >> 2) + 4 * kg;
      M[((size_t)f_own * Co + co) * T + t0 + l31] = acc[s2][r];
    }
}

static float trunc16(float v) { uint32_t u; memcpy(&u, &v, 4); u &= 0xffff0000u; float r; memcpy(&r, &u, 4); return r; }
static uint16_t hi16(float v) { uint32_t u; memcpy(&u, &v, 4); return (uint16_t)(u >> 16); }

int main(int argc, char** argv) {
  const int Ci = argc > 1 ? atoi(argv[1]) : 256, Co = argc > 2 ? atoi(argv[2]) : 256, T = argc > 3 ? atoi(argv[3]) : 8192;
  printf("Ci %d Co %d tiles %d\n", Ci, Co, T);
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
  const int grid = (T / 32) * (Co / 64) * 3;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(wino_pass_b6, dim3(grid), dim3(768), 0, 0, dV, (const u32x4_t*)dUp, dM, Ci, Co, T);
  hipError_t err = hipDeviceSynchronize();
  if (err != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(err)); return 1; }
  hipEventRecord(e0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(wino_pass_b6, dim3(grid), dim3(768), 0, 0, dV, (const u32x4_t*)dUp, dM, Ci, Co, T);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  const double items_per_cu = (double)grid / 256.0;
  printf("%.3f ms per launch; %d pass-items (%.1f per CU); %.2f us per 16-channel step and pass = %.2f us for the three passes "
         "(fp32-MFMA form: 3.9 us model, 5.1 us measured); %.1f TF/s of fp32-equivalent multiply-adds\n", ms, grid, items_per_cu,
         ms * 1e3 / items_per_cu / nsteps, 3 * ms * 1e3 / items_per_cu / nsteps, 2.0 * NF * Co * (double)Ci * T / ms / 1e9);
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
