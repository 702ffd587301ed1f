// EXPERIMENT v3 (hand-pipelined K loop; see conv_wino_bf16x6.hip for the first form) (not part of libsivae_hip.so; see NOTES_NEXT_ROUND.md): Winograd F(2x2,3x3) forward with the fp32 products
// built from bf16 pieces on v_mfma_f32_32x32x16_bf16 — every fp32 operand value a = a0 + a1 + a2 (three bf16 pieces,
// exact), the six piece products with p + q <= 2 accumulated in fp32: fp32-level accuracy (profiles/r1_probe_mfma_bf16.txt,
// profiles/r1_wino_numerics.txt) at 6 x 32 = 192 matrix-pipe cycles per 32x32 x 16-channel block instead of 8 x 64 = 512.
//
// Same block structure as conv_wino.hip (4 waves = the 4 frequency columns, 64 output channels x 32 tiles, persistent
// items, 16-channel chunks, double-buffered raw halo in LDS, one barrier per chunk); what changes is the K loop:
//   * one MFMA consumes a whole 16-channel chunk: lane (tile l31, half hh) supplies channels 2e + hh, e = 0..7, so it
//     transforms 8 channels (64 ds_read_b32 + 64 VALU), then per frequency splits its 8 values into three packed pieces;
//   * U is packed as bf16 pieces [j][chunk][i][piece][hh][co][8]: one 16-byte load per (frequency, co-subtile, piece),
//     ring of two frequencies (48 VGPRs) refilled two frequency-steps (768 MFMA cycles) ahead.
// Plain forward only (no prologue / statistics / accumulate): this file exists to settle registers, layout and speed.
//
// build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o tools/_build/wino_bf16x6 <this file> && tools/_build/wino_bf16x6
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define OOB 0xFFFFFFFFu
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned long long bytes) {
  const unsigned n = bytes > 0xFFFFFFFEull ? 0xFFFFFFFEu : (unsigned)bytes;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)n, 0x00020000);
}
__device__ __forceinline__ float buf_load_f32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ u32x4 buf_load_b128(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void buf_store_f32x2(__amdgpu_buffer_rsrc_t r, float a, float b, unsigned voff) {
  f32x2 v;
  v[0] = a;
  v[1] = b;
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, (int)voff, 0, 0);
}
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

struct B6Args {
  const float* x;
  const void* up;  // bf16 pieces [4 j][nch][4 i][3 p][2 hh][Co_pad][8]
  float* y;
  int B, Ci, Co, H, W;
  int Ci_pad, Co_pad;
  int nbh, nbw, n_co_tiles, n_items;
};


// ---------------------------------------------------------------------------------------------------------------------
// v3: ONE wave per SIMD, 4 waves = the 4 frequency columns, WM 32-channel co-subtiles per wave (WM = 4: 128 output
// channels x 32 tiles per block, 256 accumulators), software-pipelined by hand:
//   * halo: LDS-direct loads (buffer_load ... lds, no staging registers) two chunks ahead into a ring of three buffers;
//   * the input transform of chunk c+1 and the piece split of its frequency 0 run under the MFMAs of chunk c's last
//     frequency step, the split of frequency i+1 under the MFMAs of frequency i (sched_group_barrier pins the
//     interleave: 1 MFMA, then a few VALU / one load);
//   * U ring of two frequency steps, refilled unconditionally (clamped index) so that waits know the loads in flight.
#ifndef B6_WM
#define B6_WM 4
#endif
#define B6_CK 16
#define B6_EX_FLOATS (2 * 4 * 2 * 16 * 64)  // output-transform exchange area: two co-subtiles at a time (64 KB)
// VALU per MFMA inside a region: split slice = 11 VALU, transform slice = 8 VALU, over WM MFMAs
#define B6_NVS ((11 + B6_WM - 1) / B6_WM)
#define B6_NVT ((8 + B6_WM - 1) / B6_WM)
#define B6_NDR ((8 + B6_WM - 1) / B6_WM)
#ifndef B6_NDS
#define B6_NDS 8  // ds_read2st64_b32 instructions of one channel pair's 16 halo values
#endif
#ifndef B6_VPM
#define B6_VPM 3  // VALU instructions scheduled after each MFMA in the steady state
#endif

template <int TTH_L2, int TTW_L2>
__global__ void __launch_bounds__(256, 1) conv_wino_b6_kernel(B6Args a) {
  constexpr int TTH = 1 << TTH_L2, TTW = 1 << TTW_L2;
  static_assert(TTH * TTW == 32, "one image per 32-tile block in this experiment");
  constexpr int WM = B6_WM, NW = 4;
  constexpr int PXH = 2 * TTH, PXW = 2 * TTW;
  constexpr int LH = PXH + 2, LWU = PXW + 2;
  constexpr int PH = TTW + TTW / 4, RS = 2 * PH, PLANE = 256;
  static_assert(LH * RS <= 256, "halo plane fits one slot per thread");
  constexpr int CK = B6_CK, XBUF = CK * PLANE;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;  // [3][CK][PLANE]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int wj = wave;
  const int H = a.H, W = a.W, HW = H * W;
  const int nch = a.Ci_pad / CK;
  const int nfsteps = nch * 4;

  int item = blockIdx.x;
  int pt, b, r0, c0, co0;
  __amdgpu_buffer_rsrc_t xrsrc;
  const __amdgpu_buffer_rsrc_t ursrc = make_rsrc(a.up, 96ull * a.Ci_pad * a.Co_pad);
  unsigned xo, ua_base;
  // thread t stages plane slot t = xrr*RS + parity*PH + col/2 (slots outside the halo load nothing: OOB -> 0)
  const int xrr = tid / RS, xcc = 2 * ((tid % RS) % PH) + (tid % RS) / PH;
  const bool xslot = xrr < LH && xcc < LWU;
#define B6_SETUP(ITEM)                                                   \
  {                                                                      \
    const int co_tile = (ITEM) % a.n_co_tiles;                           \
    pt = (ITEM) / a.n_co_tiles;                                          \
    const int tbx = pt % a.nbw;                                          \
    const int t2 = pt / a.nbw;                                           \
    const int tby = t2 % a.nbh;                                          \
    b = t2 / a.nbh;                                                      \
    r0 = tby * PXH;                                                      \
    c0 = tbx * PXW;                                                      \
    co0 = co_tile * 32 * WM;                                             \
    xrsrc = make_rsrc(a.x + (size_t)b * a.Ci * HW, (unsigned long long)a.Ci * HW * 4ull); \
    const int r = r0 + xrr - 1, c = c0 + xcc - 1;                        \
    xo = OOB;                                                            \
    if (xslot && r >= 0 && r < H && c >= 0 && c < W) xo = (unsigned)(r * W + c) * 4u; \
    ua_base = (unsigned)(wj * nch * 24) * (unsigned)a.Co_pad * 16u + (unsigned)co0 * 16u; \
  }
  const unsigned va0 = (unsigned)(hh * a.Co_pad + l31) * 16u;
  const unsigned ua_step = (unsigned)a.Co_pad * 32u;  // bytes per (chunk, i, piece) record = 2 hh x Co_pad x 16

  const int tx = l31 & (TTW - 1), ty = l31 >> TTW_L2;
  const int ca = (wj == 0) ? 0 : ((wj == 2) ? 2 : 1);
  const int cb = (wj == 0) ? 2 : ((wj == 1) ? 2 : ((wj == 2) ? 1 : 3));
  const float sgn = (wj == 1) ? 1.f : -1.f;
  const int bb = hh * PLANE + 2 * ty * RS + tx;
  const int base_a = bb + (ca & 1) * PH + (ca >> 1);
  const int base_b = bb + (cb & 1) * PH + (cb >> 1);

  f32x16 acc[4][WM];
  u32x4 AR[2][WM][3];
  float v0[8][4], v1[8][4];  // transformed values of the current / the next chunk (ping-pong by unrolling)
  u32x4 bpA[3], bpB[3];

#define B6_LOAD_LDS(CH, BUF)                                             \
  {                                                                      \
    const int chc = (CH) < nch ? (CH) : nch - 1;                         \
    _Pragma("unroll") for (int ck = 0; ck < CK; ++ck)                    \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(                          \
          xrsrc, (float __attribute__((address_space(3)))*)(xs + (BUF)*XBUF + ck * PLANE + wave * 64), 4, xo, \
          (unsigned)(chc * CK + ck) * (unsigned)HW * 4u, 0, 0);          \
  }
#define B6_LOAD_A(F, SLOT)                                               \
  {                                                                      \
    const int fc = (F) < nfsteps ? (F) : nfsteps - 1;                    \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) {                      \
      const unsigned so = ua_base + (unsigned)(fc * 3 + p) * ua_step;    \
      _Pragma("unroll") for (int m = 0; m < WM; ++m) AR[SLOT][m][p] = buf_load_b128(ursrc, va0 + m * 512u, so); \
    }                                                                    \
  }
  // ---- the K loop is written as small scheduling regions (sched_barrier between them): one group of WM MFMAs (one
  // piece pair of one frequency, all co-subtiles), one slice of VALU work that the NEXT regions need, two U loads.
  // halo reads of channel E (of the lane's 8: channel 2E + hh) of the chunk at float offset XOFF -> raw values DA, DB
#define B6_READ1_(XOFF, E, DA, DB)                                       \
  {                                                                      \
    const float* pa = xs + (XOFF) + 2 * (E)*PLANE + base_a;              \
    const float* pb_ = xs + (XOFF) + 2 * (E)*PLANE + base_b;             \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                      \
      DA[r] = pa[r * RS];                                                \
      DB[r] = pb_[r * RS];                                               \
    }                                                                    \
  }
  // input transform of channel E from its raw values -> V[E][0..3]   (8 VALU)
#define B6_TRANS1_(V, E, DA, DB)                                         \
  {                                                                      \
    float t[4];                                                          \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) t[r] = DA[r] + sgn * DB[r]; \
    V[E][0] = t[0] - t[2];                                               \
    V[E][1] = t[1] + t[2];                                               \
    V[E][2] = t[2] - t[1];                                               \
    V[E][3] = t[1] - t[3];                                               \
  }
  // three exact bf16 pieces (truncation: 8 + 8 + 8 significant bits) of values 2D, 2D + 1 of frequency I, packed (11 VALU)
#ifdef B6_NO_VALU
#define B6_SPLIT1(V, I, BP, D)
#define B6_TRANS1(V, E, DA, DB)
#define B6_READ1(XOFF, E, DA, DB)
#else
#define B6_SPLIT1(V, I, BP, D) B6_SPLIT1_(V, I, BP, D)
#define B6_TRANS1(V, E, DA, DB) B6_TRANS1_(V, E, DA, DB)
#define B6_READ1(XOFF, E, DA, DB) B6_READ1_(XOFF, E, DA, DB)
#endif
#define B6_SPLIT1_(V, I, BP, D)                                          \
  {                                                                      \
    const float x0 = V[2 * (D)][I], x1 = V[2 * (D) + 1][I];              \
    const unsigned h0 = __builtin_bit_cast(unsigned, x0) & 0xffff0000u, h1 = __builtin_bit_cast(unsigned, x1) & 0xffff0000u; \
    const float q0 = x0 - __builtin_bit_cast(float, h0), q1 = x1 - __builtin_bit_cast(float, h1); \
    const unsigned m0 = __builtin_bit_cast(unsigned, q0) & 0xffff0000u, m1 = __builtin_bit_cast(unsigned, q1) & 0xffff0000u; \
    const float s0 = q0 - __builtin_bit_cast(float, m0), s1 = q1 - __builtin_bit_cast(float, m1); \
    BP[0][D] = __builtin_amdgcn_perm(h1, h0, 0x07060302u);               \
    BP[1][D] = __builtin_amdgcn_perm(m1, m0, 0x07060302u);               \
    BP[2][D] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s0), 0x07060302u); \
  }
#define B6_MF(I, SLOT, P, Q, BP)                                         \
  _Pragma("unroll") for (int m = 0; m < WM; ++m)                         \
    acc[I][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, AR[SLOT][m][P]), \
                                                        __builtin_bit_cast(bf16x8, BP[Q]), acc[I][m], 0, 0, 0);
  // U piece P of frequency step F (clamped past the end: the refill is unconditional so that waits know what is in flight)
#ifdef B6_NO_U
#define B6_LOAD_A1(F, SLOT, P)
#else
#define B6_LOAD_A1(F, SLOT, P) B6_LOAD_A1_(F, SLOT, P)
#endif
#define B6_LOAD_A1_(F, SLOT, P)                                          \
  {                                                                      \
    const int fc = (F) < nfsteps ? (F) : nfsteps - 1;                    \
    const unsigned so = ua_base + (unsigned)(fc * 3 + (P)) * ua_step;    \
    _Pragma("unroll") for (int m = 0; m < WM; ++m) AR[SLOT][m][P] = buf_load_b128(ursrc, va0 + m * 512u, so); \
  }
// ablation switches for timing experiments (the results are wrong with any of them): -DB6_NO_DMA drops the halo loads of
// the K loop, -DB6_NO_U the U refills, -DB6_NO_VALU the splits / transforms (pieces and values keep their first contents)
#ifdef B6_NO_DMA
#define B6_ABL_DMA(X)
#else
#define B6_ABL_DMA(X) X
#endif
#define B6_FENCE __builtin_amdgcn_sched_barrier(0);
  // interleave inside a region: WM x { 1 MFMA, NV VALU, NL buffer loads, ND LDS reads }
#ifdef B6_NOPAT
#define B6_PAT(NV, NL, ND)
#else
#define B6_PAT(NV, NL, ND)                                               \
  _Pragma("unroll") for (int g_ = 0; g_ < WM; ++g_) {                    \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   \
    if (NV) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);          \
    if (NL) __builtin_amdgcn_sched_group_barrier(0x020, NL, 0);          \
    if (ND) __builtin_amdgcn_sched_group_barrier(0x100, ND, 0);          \
  }
#endif
  // frequency step I (0..2) of a chunk: MFMAs of frequency I from pieces BPC; under them the split of frequency I + 1 into
  // BPN and the transform of channels 2I, 2I + 1 of the next chunk (raw values read one step earlier into DA*/DB*).
  // The piece-pair order (smallest products first) frees U piece 2 after the third group, piece 1 after the fifth.
#define B6_STEP(F, I, SLOT, CUR, NXT, XN, BPC, BPN)                      \
  {                                                                      \
    B6_MF(I, SLOT, 0, 2, BPC) B6_SPLIT1(CUR, (I) + 1, BPN, 0) B6_PAT(B6_NVS, 0, 0) B6_FENCE \
    B6_MF(I, SLOT, 1, 1, BPC) B6_SPLIT1(CUR, (I) + 1, BPN, 1) B6_PAT(B6_NVS, 0, 0) B6_FENCE \
    B6_MF(I, SLOT, 2, 0, BPC) B6_SPLIT1(CUR, (I) + 1, BPN, 2) B6_LOAD_A1((F) + 2, SLOT, 2) B6_PAT(B6_NVS, 1, 0) B6_FENCE \
    B6_MF(I, SLOT, 0, 1, BPC) B6_SPLIT1(CUR, (I) + 1, BPN, 3) B6_PAT(B6_NVS, 0, 0) B6_FENCE \
    B6_MF(I, SLOT, 1, 0, BPC) B6_TRANS1(NXT, 2 * (I), da0, db0) B6_LOAD_A1((F) + 2, SLOT, 1) B6_PAT(B6_NVT, 1, 0) B6_FENCE \
    B6_MF(I, SLOT, 0, 0, BPC) B6_TRANS1(NXT, 2 * (I) + 1, da1, db1) B6_LOAD_A1((F) + 2, SLOT, 0) \
    B6_READ1(XN, 2 * (I) + 2, da0, db0) B6_READ1(XN, 2 * (I) + 3, da1, db1) B6_PAT(B6_NVT, 1, B6_NDR) B6_FENCE \
  }
  // last step of the chunk: the transform of the next chunk's channels 6, 7 first, then the split of ITS frequency 0
#define B6_STEP3(F, SLOT, NXT, BPC, BPN)                                 \
  {                                                                      \
    B6_MF(3, SLOT, 0, 2, BPC) B6_TRANS1(NXT, 6, da0, db0) B6_PAT(B6_NVT, 0, 0) B6_FENCE \
    B6_MF(3, SLOT, 1, 1, BPC) B6_TRANS1(NXT, 7, da1, db1) B6_PAT(B6_NVT, 0, 0) B6_FENCE \
    B6_MF(3, SLOT, 2, 0, BPC) B6_SPLIT1(NXT, 0, BPN, 0) B6_LOAD_A1((F) + 2, SLOT, 2) B6_PAT(B6_NVS, 1, 0) B6_FENCE \
    B6_MF(3, SLOT, 0, 1, BPC) B6_SPLIT1(NXT, 0, BPN, 1) B6_PAT(B6_NVS, 0, 0) B6_FENCE \
    B6_MF(3, SLOT, 1, 0, BPC) B6_SPLIT1(NXT, 0, BPN, 2) B6_LOAD_A1((F) + 2, SLOT, 1) B6_PAT(B6_NVS, 1, 0) B6_FENCE \
    B6_MF(3, SLOT, 0, 0, BPC) B6_SPLIT1(NXT, 0, BPN, 3) B6_LOAD_A1((F) + 2, SLOT, 0) B6_PAT(B6_NVS, 1, 0) B6_FENCE \
  }
  // one chunk.  The barrier at the top publishes the halo of chunk CH + 1 (requested a whole chunk ago: more than 63
  // loads back) and retires the readers of the buffer that chunk CH + 2's halo is about to overwrite.
#define B6_CHUNK(CH, CUR, NXT)                                           \
  {                                                                      \
    __builtin_amdgcn_s_waitcnt(0xCF7F); /* vmcnt(63) */                  \
    __syncthreads();                                                     \
    const int xn = (((CH) + 1) % 3) * XBUF;                              \
    B6_ABL_DMA(B6_LOAD_LDS((CH) + 2, ((CH) + 2) % 3))                    \
    B6_READ1(xn, 0, da0, db0)                                            \
    B6_READ1(xn, 1, da1, db1)                                            \
    B6_FENCE                                                             \
    B6_STEP((CH)*4 + 0, 0, 0, CUR, NXT, xn, bpA, bpB)                    \
    B6_STEP((CH)*4 + 1, 1, 1, CUR, NXT, xn, bpB, bpA)                    \
    B6_STEP((CH)*4 + 2, 2, 0, CUR, NXT, xn, bpA, bpB)                    \
    B6_STEP3((CH)*4 + 3, 1, NXT, bpB, bpA)                               \
  }
  float da0[4], db0[4], da1[4], db1[4];

  B6_SETUP(item)
  B6_LOAD_LDS(0, 0)
  B6_LOAD_LDS(1, 1)
  B6_LOAD_A(0, 0)
  B6_LOAD_A(1, 1)
  for (;;) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][m][r] = 0.f;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's part of halo chunks 0 and 1 has landed
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      B6_READ1_(0, e, da0, db0)
      B6_TRANS1_(v0, e, da0, db0)
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      B6_SPLIT1_(v0, 0, bpA, d)
      B6_SPLIT1_(v0, 1, bpB, d)
    }
    {
      int ch = 0;
      for (; ch + 1 < nch; ch += 2) {
        B6_CHUNK(ch, v0, v1)
        B6_CHUNK(ch + 1, v1, v0)
      }
      if (ch < nch) B6_CHUNK(ch, v0, v1)
    }
    __syncthreads();

    const int e_b = b, e_r0 = r0, e_c0 = c0, e_co0 = co0;
    const int next = item + (int)gridDim.x;
    const bool has_next = next < a.n_items;
    // the next item's first two halo chunks and U steps go in flight before the output transform (the exchange area
    // below does not alias the halo ring, and the U ring is dead here)
    if (has_next) {
      B6_SETUP(next)
      B6_LOAD_LDS(0, 0)
      B6_LOAD_LDS(1, 1)
      B6_LOAD_A(0, 0)
      B6_LOAD_A(1, 1)
    }
    // ---- output transform (as conv_wino.hip): rows in registers, columns across the four waves through LDS, two
    // co-subtiles (64 KB) at a time.  The lane index is laundered so that the address arithmetic stays inside the item
    // loop as base + immediate instead of being hoisted into ~250 loop-invariant registers (which spilled).
    {
      int lane_l = lane;
      asm volatile("" : "+v"(lane_l));
      float* ex = smem + 3 * XBUF;  // [2 ar][4 j][2 cg][16 r][64 lanes]
      constexpr int PPW = 32 / NW;  // rows (cg, r) of a half per wave
      const __amdgpu_buffer_rsrc_t yrsrc =
          make_rsrc(a.y + (size_t)e_b * a.Co * HW, (unsigned long long)a.Co * HW * 4ull);
      const int row_base = e_r0 + 2 * ty, col = e_c0 + 2 * tx;
      float* exw0 = ex + (wj * 2 * 16) * 64 + lane_l;
      float* exw1 = ex + ((4 + wj) * 2 * 16) * 64 + lane_l;
      const float* exr = ex + lane_l;
#pragma unroll
      for (int h = 0; h < WM / 2; ++h) {
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = 2 * h + mm;
            exw0[(mm * 16 + r) * 64] = acc[0][m][r] + acc[1][m][r] + acc[2][m][r];
            exw1[(mm * 16 + r) * 64] = acc[1][m][r] - acc[2][m][r] - acc[3][m][r];
            if ((r & 3) == 3) B6_FENCE  // keep the accumulator reads in small batches (all 256 at once spill)
          }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < PPW; ++rr) {
          const int p = wave * PPW + rr;  // (cg = p >> 4, r = p & 15) of this half
          const int chn = e_co0 + (2 * h + (p >> 4)) * 32 + (p & 3) + 8 * ((p & 15) >> 2) + 4 * hh;
          const unsigned yo = (chn < a.Co && col < W) ? (unsigned)((chn * H + row_base) * W + col) * 4u : OOB;
#pragma unroll
          for (int ar = 0; ar < 2; ++ar) {
            float e[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) e[jj] = exr[(((ar * 4 + jj) * 2) * 16 + p) * 64];
            const bool ok = yo != OOB && row_base + ar < H;
            buf_store_f32x2(yrsrc, e[0] + e[1] + e[2], e[1] - e[2] - e[3], ok ? yo + (unsigned)(ar * W) * 4u : OOB);
          }
        }
        __syncthreads();
      }
    }
    if (!has_next) break;
    item = next;
  }
}

// U = G g G^T split into bf16 pieces, packed [j][chunk][i][piece][hh][Co_pad][8]: element e of half hh = channel 2e + hh
__global__ void __launch_bounds__(256) pack_wino_b6_kernel(const float* __restrict__ w, unsigned short* __restrict__ up,
                                                           int Co, int Ci, int kpad, int npad) {
  const size_t total = (size_t)kpad * npad;
  const int nch = kpad / 16;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int n = (int)(idx % npad), k = (int)(idx / npad);
    float g[3][3];
    const bool ok = k < Ci && n < Co;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) g[r][c] = ok ? w[((size_t)n * Ci + k) * 9 + r * 3 + c] : 0.f;
    float gg[4][3];
    for (int c = 0; c < 3; ++c) {
      gg[0][c] = g[0][c];
      gg[1][c] = 0.5f * (g[0][c] + g[1][c] + g[2][c]);
      gg[2][c] = 0.5f * (g[0][c] - g[1][c] + g[2][c]);
      gg[3][c] = g[2][c];
    }
    float u[4][4];
    for (int i = 0; i < 4; ++i) {
      u[i][0] = gg[i][0];
      u[i][1] = 0.5f * (gg[i][0] + gg[i][1] + gg[i][2]);
      u[i][2] = 0.5f * (gg[i][0] - gg[i][1] + gg[i][2]);
      u[i][3] = gg[i][2];
    }
    const int chunk = k >> 4, kk = k & 15, hh = kk & 1, e = kk >> 1;
    for (int j = 0; j < 4; ++j)
      for (int i = 0; i < 4; ++i) {
        float val = u[i][j];
        for (int p = 0; p < 3; ++p) {
          const __bf16 h = (__bf16)val;
          val -= (float)h;
          const size_t rec = ((((size_t)j * nch + chunk) * 4 + i) * 3 + p) * 2 + hh;
          up[(rec * npad + n) * 8 + e] = __builtin_bit_cast(unsigned short, h);
        }
      }
  }
}

#define CK_(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static int run(int B, int Ci, int Co, int H, int W, bool check, int reps) {
  const int kpad = (Ci + 15) / 16 * 16, npad = (Co + 63) / 64 * 64;
  std::vector<float> x((size_t)B * Ci * H * W), w((size_t)Co * Ci * 9), y((size_t)B * Co * H * W);
  srand(7);
  for (auto& v : x) v = (float)(rand() % 20001) / 10000.f - 1.f;
  for (auto& v : w) v = ((float)(rand() % 20001) / 10000.f - 1.f) / sqrtf((float)Ci * 9.f);
  float *dx, *dw, *dy;
  unsigned short* dup;
  CK_(hipMalloc(&dx, x.size() * 4)); CK_(hipMalloc(&dw, w.size() * 4)); CK_(hipMalloc(&dy, y.size() * 4));
  CK_(hipMalloc(&dup, (size_t)96 * kpad * npad));
  CK_(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
  CK_(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
  CK_(hipMemset(dy, 0xff, y.size() * 4));
  hipLaunchKernelGGL(pack_wino_b6_kernel, dim3(cdiv((long long)kpad * npad, 256)), dim3(256), 0, 0, dw, dup, Co, Ci, kpad, npad);
  B6Args a;
  a.x = dx; a.up = dup; a.y = dy;
  a.B = B; a.Ci = Ci; a.Co = Co; a.H = H; a.W = W; a.Ci_pad = kpad; a.Co_pad = npad;
  a.nbh = cdiv(H, 4); a.nbw = cdiv(W, 32); a.n_co_tiles = cdiv(Co, 32 * B6_WM);
  a.n_items = B * a.nbh * a.nbw * a.n_co_tiles;
  const size_t lds_x = (size_t)3 * 16 * 256 * 4;  // three halo buffers
  const size_t lds = (size_t)B6_EX_FLOATS * 4 + lds_x;  // the exchange area sits behind the halo ring (no aliasing)
  auto kern = conv_wino_b6_kernel<1, 4>;
  CK_(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = a.n_items < 256 ? a.n_items : 256;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, a);
  CK_(hipDeviceSynchronize());
  if (check) {
    CK_(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
    double err = 0, ymax = 0;
    for (int b = 0; b < B; ++b)
      for (int co = 0; co < Co; ++co)
        for (int h = 0; h < H; ++h)
          for (int ww = 0; ww < W; ++ww) {
            double s = 0;
            for (int ci = 0; ci < Ci; ++ci)
              for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) {
                  const int hy = h + r - 1, wx = ww + c - 1;
                  if (hy < 0 || hy >= H || wx < 0 || wx >= W) continue;
                  s += (double)x[((size_t)(b * Ci + ci) * H + hy) * W + wx] * w[((size_t)co * Ci + ci) * 9 + r * 3 + c];
                }
            const double got = y[((size_t)(b * Co + co) * H + h) * W + ww];
            err = fmax(err, fabs(got - s));
            ymax = fmax(ymax, fabs(s));
          }
    printf("B=%d %d->%d @%dx%d: max-norm relative error vs fp64 direct conv = %.3e\n", B, Ci, Co, H, W, err / ymax);
  }
  if (reps > 0) {
    hipEvent_t e0, e1;
    CK_(hipEventCreate(&e0)); CK_(hipEventCreate(&e1));
    CK_(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, a);
    CK_(hipEventRecord(e1, 0));
    CK_(hipEventSynchronize(e1));
    float ms = 0;
    CK_(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double fl = 2.0 * B * H * W * (double)Ci * Co * 9;
    printf("B=%d %d->%d @%dx%d: %.3f ms  %.1f TF/s algorithmic (conv_wino.hip: see tools/bench_conv.py)\n", B, Ci, Co, H, W, ms,
           fl / ms / 1e9);
  }
  (void)hipFree(dx); (void)hipFree(dw); (void)hipFree(dy); (void)hipFree(dup);
  return 0;
}

int main(int argc, char** argv) {
  if (run(2, 48, 72, 8, 32, true, 0)) return 1;   // ragged output channels, three chunks
  if (run(2, 16, 40, 4, 32, true, 0)) return 1;    // a single chunk
  if (run(1, 64, 64, 12, 64, true, 0)) return 1;
  if (argc > 1) return 0;
  if (run(128, 256, 256, 64, 64, false, 5)) return 1;
  if (run(128, 512, 512, 32, 32, false, 5)) return 1;
  if (run(128, 128, 128, 128, 128, false, 5)) return 1;
  if (run(128, 64, 64, 256, 256, false, 3)) return 1;
  return 0;
}
