// Winograd F(4x4,3x3) weight gradient of the 3x3 stride-1 "same" convolution (fp32 MFMA, gfx950).
//
// With Y = A^T [ (G g G^T) . (B^T d B) ] A per 4x4 output tile (conv_wino4.hip), the filter gradient is linear in the
// same transformed operands (the weight half of aten::convolution_backward of the nn.Conv2d(k=3) layers,
// soft_intro_vae/train_soft_intro_vae.py:56-61):
//
//   dU[i][j][co][ci] = sum over tiles  Mg[i][j][co][tile] * V[i][j][ci][tile]     Mg = A dY A^T  (4x4 -> 6x6)
//   dg[co][ci]       = G^T dU[.][.][co][ci] G                                     V  = B^T d  B   (6x6 -> 6x6)
//
// 36 independent GEMMs with K = number of tiles: 36 multiplies per (tile, co, ci) instead of the 144 of the direct form
// (9 taps x 16 pixels) and the 64 of the F(2x2,3x3) form (conv_wino_wgrad.hip).
//
// fp32 "MFMA" time and ordinary VALU time ADD on a gfx950 SIMD (conv_wino4.hip), so both operands are transformed ONCE
// per block into LDS and the MFMA loop is nothing but ds_read_b32 + MFMA:
//   * a block = 12 waves = 64 output channels x 32 input channels x 36 frequencies; wave (j, s) owns frequency column j
//     of output-channel subtile s: 6 accumulators = 96 registers, three waves per SIMD, one block per CU;
//   * a stage = a 4 x 16 pixel strip = 4 tiles = two k-steps (a k-step is a pair of tiles).  Its raw operands — the
//     6 x 24 input halo of 32 channels (18 KB) and the 4 x 16 dY strip of 64 channels (16 KB) — arrive by LDS-direct
//     16-byte loads (three per wave and stage) into a ring of three slots, requested three stages ahead;
//   * transform phase: dY -> Mg by threads (tile, co) (waves 0-3), x -> V by threads (tile, ci, column pair) (waves 4-9,
//     the pairing of conv_wino4.hip) into Mg[f][s][tile][32] / V[f][tile][32], which the MFMA lanes read linearly;
//   * MFMA phase: 2 x 6 MFMAs per wave; under it the requests of stage s+3 and — with the fused BatchNorm + LeakyReLU
//     prologue — the in-place rewrite of stage s+1's raw halo (zero padding restored through the table select).
// Two LDS-only barriers per stage.  The sum over tiles is split across blocks ("slices") of whole stage triples; every
// slice writes its partial dU and a second kernel adds the slices in a fixed order and applies G^T . G (no atomics).
#include "common.h"
#include <stdlib.h>

struct Wino4WgArgs {
  const float* x;
  const float* dy;
  float* ws;  // [n_slices][36][Co_pad][Ci_pad]
  const float* pro_mean;
  const float* pro_invstd;
  const float* pro_gamma;
  const float* pro_beta;
  float pro_slope;
  int pro_seg_images, pro_nseg;  // segments (conv_wino.hip): pro_mean / pro_invstd are [pro_nseg][Ci], pro_nseg <= 2
  int B, Ci, Co, H, W;
  int Ci_pad, Co_pad;
  int nrh, nrw, nstages, sps;
  int n_co_tiles, n_ci_tiles;
  int nblk, xcd_remap;
};

#define G4_NT 768
#define G4_XS 4608  // raw x slot: [6 rows][32 ci][6 groups of 4 floats]
#define G4_YS 4096  // raw dY slot: [4 rows][64 co][4 groups, swizzled by (co >> 2) & 3]
#define G4_VS 4608  // V[36][4 tiles][32 ci]
#define G4_MS 9216  // Mg[36][2 subtiles][4 tiles][32 co]

// Timing ablations (results WRONG with any bit set): 1 no LDS-direct loads, 2 no transform phase, 4 no MFMAs
#ifndef G4_ABLATE
#define G4_ABLATE 0
#endif

template <bool PRO>
__global__ void __launch_bounds__(G4_NT, 1) wino4_wgrad_kernel(Wino4WgArgs a) {
  // every ring slot is its own static array: the compiler orders an LDS access behind each in-flight LDS-direct load
  // it cannot prove disjoint (conv_wino4.hip)
  __shared__ __attribute__((aligned(16))) float rx0[G4_XS];
  __shared__ __attribute__((aligned(16))) float rx1[G4_XS];
  __shared__ __attribute__((aligned(16))) float rx2[G4_XS];
  __shared__ __attribute__((aligned(16))) float ry0[G4_YS];
  __shared__ __attribute__((aligned(16))) float ry1[G4_YS];
  __shared__ __attribute__((aligned(16))) float ry2[G4_YS];
  __shared__ __attribute__((aligned(16))) float vs[G4_VS];
  __shared__ __attribute__((aligned(16))) float ms[G4_MS];
  __shared__ float4 pro4[PRO ? 64 : 1];  // {mean, invstd*gamma, beta, -} per (segment, channel of the tile)
#define G4_RX(K) ((K) == 0 ? rx0 : ((K) == 1 ? rx1 : rx2))
#define G4_RY(K) ((K) == 0 ? ry0 : ((K) == 1 ? ry1 : ry2))

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int wj = wave % 6, wsb = wave / 6;
  const int H = a.H, W = a.W, HW = H * W;

  const int ntiles = a.n_co_tiles * a.n_ci_tiles;
  // XCD-aware block order (conv_wino_wgrad.hip): the (co, ci) tiles of one slice read the same stages; put them on one XCD
  int lb = (int)blockIdx.x;
  if (a.xcd_remap) {
    lb = (lb & 7) * ((int)gridDim.x >> 3) + (lb >> 3);
    if (lb >= a.nblk) return;
  }
  const int tile = lb % ntiles, slice = lb / ntiles;
  const int ci0 = (tile % a.n_ci_tiles) * 32, co0 = (tile / a.n_ci_tiles) * 64;
  const int s_begin = slice * a.sps;
  const int s_end = (s_begin + a.sps < a.nstages) ? (s_begin + a.sps) : a.nstages;

  // ---- request role: wave w issues pieces 3w .. 3w+2 of a stage; pieces 0..17 are the x slot (waves 0-5), 18..33 the dY
  // slot (waves 6-11; the last two repeat piece 33).  A piece = 64 lanes x 16 bytes = 1 KB of consecutive LDS.
  const bool req_x = wave < 6;
  unsigned rq_off[3], rq_bits[3];
  int rq_lds[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int d = wave * 3 + i;
    if (d < 18) {
      const int pos = d * 64 + lane, row = pos / 192, rem = pos - row * 192, ci = rem / 6, p = rem - ci * 6;
      rq_off[i] = (ci0 + ci < a.Ci) ? (unsigned)((ci0 + ci) * HW + row * W + 4 * p) * 4u : SIVAE_OOB;
      rq_bits[i] = (row == 0 ? 1u : 0u) | (row == 5 ? 2u : 0u) | (p == 0 ? 4u : 0u) | (p == 5 ? 8u : 0u) | 16u;
      rq_lds[i] = d * 256;
    } else {
      const int dd = d - 18 < 15 ? d - 18 : 15;
      const int pos = dd * 64 + lane, row = pos >> 8, co = (pos & 255) >> 2, p = (pos & 3) ^ ((co >> 2) & 3);
      rq_off[i] = (co0 + co < a.Co) ? (unsigned)((co0 + co) * HW + row * W + 4 * p) * 4u : SIVAE_OOB;
      rq_bits[i] = 16u;
      rq_lds[i] = dd * 256;
    }
  }
  // ---- prologue role: 16-byte group tid of the x slot, and group 768 + (tid - 384) for the upper half of the block
  const int fq1 = 768 + (tid >= 384 ? tid - 384 : 0);
  const bool f_two = __builtin_amdgcn_readfirstlane(tid >= 384 ? 1 : 0) != 0;
  int f_ci0, f_ci1;
  unsigned f_bits0, f_bits1;
  {
    const int row = tid / 192, rem = tid - row * 192, ci = rem / 6, p = rem - ci * 6;
    f_ci0 = ci;
    f_bits0 = (row == 0 ? 1u : 0u) | (row == 5 ? 2u : 0u) | (p == 0 ? 4u : 0u) | (p == 5 ? 8u : 0u) | 16u;
  }
  {
    const int row = fq1 / 192, rem = fq1 - row * 192, ci = rem / 6, p = rem - ci * 6;
    f_ci1 = ci;
    f_bits1 = (row == 0 ? 1u : 0u) | (row == 5 ? 2u : 0u) | (p == 0 ? 4u : 0u) | (p == 5 ? 8u : 0u) | 16u;
  }
  // ---- transform roles
  // dY (waves 0-3): tile = wave, co = lane
  const int ty_rd = (lane * 4 + ((wave & 3) ^ ((lane >> 2) & 3))) * 4;
  const int ty_wr = (lane >> 5) * 128 + (wave & 3) * 32 + (lane & 31);
  // x (waves 4-9): column pair tp = (wave - 4) >> 1 ((1,2), (3,4), (0,5)), tile = 2 * ((wave - 4) & 1) + hh, ci = l31
  const int xw = wave >= 4 ? wave - 4 : 0;
  const int tp = xw >> 1, xt = 2 * (xw & 1) + hh;
  const int tx_rd = (l31 * 6 + xt + 1) * 4;  // patch columns 1..4 = group tile + 1
  const int tx_wr = xt * 32 + l31;
  const int jA = tp == 0 ? 1 : (tp == 1 ? 3 : 0), jB = tp == 0 ? 2 : (tp == 1 ? 4 : 5);
  const float t_al = tp == 0 ? -4.f : -1.f;  // a = d4 + al*d2
  const float t_be = tp == 0 ? 1.f : 2.f;    // b = be*d3 + ga*d1
  const float t_ga = tp == 0 ? -4.f : -2.f;
  // ---- MFMA role: A = Mg[(i*6 + wj)][wsb][2kk + hh][l31], B = V[(i*6 + wj)][2kk + hh][l31]
  const int m_rd = wj * 256 + wsb * 128 + lane;
  const int v_rd = wj * 128 + lane;

  // ---- stage to request next (scalar): image db, strip row dry, strip column drx
  int sd = s_begin;
  int db = sd / (a.nrh * a.nrw);
  int dry, drx;
  {
    const int rem = sd - db * (a.nrh * a.nrw);
    dry = rem / a.nrw;
    drx = rem - dry * a.nrw;
  }
  unsigned inf0 = 0, inf1 = 0, inf2 = 0;  // per ring slot: border / tail bits and the segment's table offset << 8
#define G4_INF(K) ((K) == 0 ? inf0 : ((K) == 1 ? inf1 : inf2))
#define G4_SETINF(K, V) { if ((K) == 0) inf0 = (V); else if ((K) == 1) inf1 = (V); else inf2 = (V); }

  // request stage sd into ring slot K (three unconditional LDS-direct loads per wave; invalid groups read as zeros)
#define G4_REQ(K)                                                                                   \
  {                                                                                                 \
    const unsigned fl_ = (dry == 0 ? 1u : 0u) | (dry == a.nrh - 1 ? 2u : 0u) | (drx == 0 ? 4u : 0u) |  \
                         (drx == a.nrw - 1 ? 8u : 0u) | (sd >= s_end ? 16u : 0u);                   \
    const float* base_ = req_x ? a.x + ((long long)db * a.Ci * HW + (long long)(dry * 4 - 1) * W + (drx * 16 - 4)) \
                               : a.dy + ((long long)db * a.Co * HW + (long long)(dry * 4) * W + drx * 16); \
    const __amdgpu_buffer_rsrc_t rs_ = make_rsrc(base_, 0xFFFFFFFEull);                             \
    float* lds_ = req_x ? G4_RX(K) : G4_RY(K);                                                      \
    unsigned o0_ = rq_off[0], o1_ = rq_off[1], o2_ = rq_off[2];                                     \
    if (fl_) {                                                                                      \
      o0_ = (rq_bits[0] & fl_) ? SIVAE_OOB : o0_;                                                   \
      o1_ = (rq_bits[1] & fl_) ? SIVAE_OOB : o1_;                                                   \
      o2_ = (rq_bits[2] & fl_) ? SIVAE_OOB : o2_;                                                   \
    }                                                                                               \
    if (!(G4_ABLATE & 1)) {                                                                         \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (float __attribute__((address_space(3)))*)(lds_ + rq_lds[0]), 16, o0_, 0, 0, 0); \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (float __attribute__((address_space(3)))*)(lds_ + rq_lds[1]), 16, o1_, 0, 0, 0); \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (float __attribute__((address_space(3)))*)(lds_ + rq_lds[2]), 16, o2_, 0, 0, 0); \
    }                                                                                               \
    if (PRO) G4_SETINF(K, fl_ | ((db >= a.pro_seg_images ? 32u : 0u) << 8))                         \
    ++sd;                                                                                           \
    if (++drx == a.nrw) {                                                                           \
      drx = 0;                                                                                      \
      if (++dry == a.nrh) {                                                                         \
        dry = 0;                                                                                    \
        ++db;                                                                                       \
      }                                                                                             \
    }                                                                                               \
  }
  // fused BatchNorm + LeakyReLU prologue, in place on the raw x slot K: x' = max(v, slope v), v = (x - mean) scale + beta;
  // groups outside the image (and channels beyond Ci: zero table rows) stay zero
#define G4_FIX1(K, Q, CI, BITS)                                                                     \
  {                                                                                                 \
    const unsigned in_ = G4_INF(K);                                                                 \
    float4 p_ = pro4[(in_ >> 8) + (CI)];                                                            \
    if ((BITS) & in_ & 31u) p_ = make_float4(0.f, 0.f, 0.f, 0.f);                                   \
    float4* q_ = reinterpret_cast<float4*>(G4_RX(K) + (Q) * 4);                                     \
    float4 v_ = *q_;                                                                                \
    v_.x = fmaf(v_.x - p_.x, p_.y, p_.z);                                                           \
    v_.y = fmaf(v_.y - p_.x, p_.y, p_.z);                                                           \
    v_.z = fmaf(v_.z - p_.x, p_.y, p_.z);                                                           \
    v_.w = fmaf(v_.w - p_.x, p_.y, p_.z);                                                           \
    v_.x = fmaxf(v_.x, v_.x * a.pro_slope);                                                         \
    v_.y = fmaxf(v_.y, v_.y * a.pro_slope);                                                         \
    v_.z = fmaxf(v_.z, v_.z * a.pro_slope);                                                         \
    v_.w = fmaxf(v_.w, v_.w * a.pro_slope);                                                         \
    *q_ = v_;                                                                                       \
  }
#define G4_FIX(K)                                                                                   \
  {                                                                                                 \
    G4_FIX1(K, tid, f_ci0, f_bits0)                                                                 \
    if (f_two) G4_FIX1(K, fq1, f_ci1, f_bits1)                                                      \
  }
#define G4_FENCE __builtin_amdgcn_sched_barrier(0);
  // workgroup barrier that orders LDS traffic only (a __syncthreads() with LDS-direct loads in flight waits vmcnt(0))
#define G4_LDS_BARRIER asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // A (4 -> 6) of one vector: {d0, e+o, e-o, e'+2o', e'-2o', d3}, e = d0+d2, o = d1+d3, e' = d0+4d2, o' = d1+4d3
#define G4_A4(D0, D1, D2, D3, M)                                                                    \
  {                                                                                                 \
    const float ae_ = (D0) + (D2), ao_ = (D1) + (D3);                                               \
    const float ae2_ = fmaf(4.f, (D2), (D0)), ao2_ = fmaf(4.f, (D3), (D1));                         \
    M[0] = (D0);                                                                                    \
    M[1] = ae_ + ao_;                                                                               \
    M[2] = ae_ - ao_;                                                                               \
    M[3] = fmaf(2.f, ao2_, ae2_);                                                                   \
    M[4] = fmaf(-2.f, ao2_, ae2_);                                                                  \
    M[5] = (D3);                                                                                    \
  }
  // transform phase of the stage in ring slot K
#define G4_TRANSFORM(K)                                                                             \
  if (!(G4_ABLATE & 2)) {                                                                           \
    if (wave < 4) {                                                                                 \
      const float* p_ = G4_RY(K) + ty_rd;                                                           \
      float4 d_[4];                                                                                 \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) d_[r] = *reinterpret_cast<const float4*>(p_ + r * 1024); \
      float m_[4][6];                                                                               \
      G4_A4(d_[0].x, d_[1].x, d_[2].x, d_[3].x, m_[0])                                              \
      G4_A4(d_[0].y, d_[1].y, d_[2].y, d_[3].y, m_[1])                                              \
      G4_A4(d_[0].z, d_[1].z, d_[2].z, d_[3].z, m_[2])                                              \
      G4_A4(d_[0].w, d_[1].w, d_[2].w, d_[3].w, m_[3])                                              \
      float* q_ = ms + ty_wr;                                                                       \
      _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                               \
        float o_[6];                                                                                \
        G4_A4(m_[0][i], m_[1][i], m_[2][i], m_[3][i], o_)                                           \
        _Pragma("unroll") for (int j = 0; j < 6; ++j) q_[(i * 6 + j) * 256] = o_[j];                \
      }                                                                                             \
    } else if (wave < 10) {                                                                         \
      const float* p_ = G4_RX(K) + tx_rd;                                                           \
      float tA_[6], tB_[6];                                                                         \
      if (tp == 2) {                                                                                \
        _Pragma("unroll") for (int r = 0; r < 6; ++r) {                                             \
          const float4 d_ = *reinterpret_cast<const float4*>(p_ + r * 768);                         \
          const float e0_ = p_[r * 768 - 1], e5_ = p_[r * 768 + 4];                                 \
          tA_[r] = fmaf(4.f, e0_, fmaf(-5.f, d_.y, d_.w));                                          \
          tB_[r] = fmaf(4.f, d_.x, fmaf(-5.f, d_.z, e5_));                                          \
        }                                                                                           \
      } else {                                                                                      \
        _Pragma("unroll") for (int r = 0; r < 6; ++r) {                                             \
          const float4 d_ = *reinterpret_cast<const float4*>(p_ + r * 768);                         \
          const float a_ = fmaf(t_al, d_.y, d_.w);                                                  \
          const float b_ = fmaf(t_ga, d_.x, t_be * d_.z);                                           \
          tA_[r] = a_ + b_;                                                                         \
          tB_[r] = a_ - b_;                                                                         \
        }                                                                                           \
      }                                                                                             \
      G4_TCOL(tA_, jA)                                                                              \
      G4_TCOL(tB_, jB)                                                                              \
    }                                                                                               \
  }
  // V[.][J] = B^T t (the row direction) for one column of the pair
#define G4_TCOL(T, J)                                                                               \
  {                                                                                                 \
    const float A_ = fmaf(-4.f, T[2], T[4]), B_ = fmaf(-4.f, T[1], T[3]);                           \
    const float C_ = T[4] - T[2], D_ = T[3] - T[1];                                                 \
    float* q_ = vs + (J) * 128 + tx_wr;                                                             \
    q_[0 * 768] = fmaf(4.f, T[0], fmaf(-5.f, T[2], T[4]));                                          \
    q_[1 * 768] = A_ + B_;                                                                          \
    q_[2 * 768] = A_ - B_;                                                                          \
    q_[3 * 768] = fmaf(2.f, D_, C_);                                                                \
    q_[4 * 768] = fmaf(-2.f, D_, C_);                                                               \
    q_[5 * 768] = fmaf(4.f, T[1], fmaf(-5.f, T[3], T[5]));                                          \
  }
#define G4_READ(KK, AV, BV)                                                                         \
  {                                                                                                 \
    const float* pa_ = ms + m_rd + (KK) * 64;                                                       \
    const float* pb_ = vs + v_rd + (KK) * 64;                                                       \
    _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                 \
      AV[i] = pa_[i * 1536];                                                                        \
      BV[i] = pb_[i * 768];                                                                         \
    }                                                                                               \
  }
#define G4_MMA(AV, BV)                                                                              \
  if (!(G4_ABLATE & 4)) {                                                                           \
    _Pragma("unroll") for (int i = 0; i < 6; ++i)                                                   \
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[i], BV[i], acc[i], 0, 0, 0);               \
  }
  // One stage s in ring slot K.  On entry: the raw operands of stages s and s+1 have landed and are visible, slot K is
  // already rewritten by the prologue, stage s+2 is in flight, V / Mg are free.
#define G4_STAGE(K)                                                                                 \
  {                                                                                                 \
    G4_TRANSFORM(K)                                                                                 \
    G4_FENCE                                                                                        \
    G4_LDS_BARRIER                                                                                  \
    G4_FENCE                                                                                        \
    G4_REQ(K)                                                                                       \
    float a0_[6], b0_[6], a1_[6], b1_[6];                                                           \
    G4_READ(0, a0_, b0_)                                                                            \
    G4_READ(1, a1_, b1_)                                                                            \
    G4_FENCE                                                                                        \
    G4_MMA(a0_, b0_)                                                                                \
    G4_FENCE                                                                                        \
    if (PRO) G4_FIX(((K) + 1) % 3)                                                                  \
    G4_FENCE                                                                                        \
    G4_MMA(a1_, b1_)                                                                                \
    G4_FENCE                                                                                        \
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                                                \
    G4_LDS_BARRIER                                                                                  \
    G4_FENCE                                                                                        \
  }

  f32x16 acc[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  if (PRO) {
    for (int idx = tid; idx < a.pro_nseg * 32; idx += G4_NT) {
      const int c = ci0 + (idx & 31), so = (idx >> 5) * a.Ci;
      pro4[idx] = c < a.Ci ? make_float4(a.pro_mean[so + c], a.pro_invstd[so + c] * a.pro_gamma[c], a.pro_beta[c], 0.f)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  if (s_begin < s_end) {
    G4_REQ(0)
    G4_REQ(1)
    G4_REQ(2)
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    __syncthreads();
    if (PRO) {
      G4_FIX(0)
      __syncthreads();
    }
    for (int s = s_begin; s < s_end; s += 3) {  // (whole triples: stages beyond s_end are all-zero operands)
      G4_STAGE(0)
      G4_STAGE(1)
      G4_STAGE(2)
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- partial dU of this slice: acc[i][r] -> frequency (i, wj), co = co0 + wsb*32 + row(r, hh), ci = ci0 + l31
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float* base = a.ws + ((size_t)(slice * 36 + i * 6 + wj) * a.Co_pad + co0 + wsb * 32) * a.Ci_pad + ci0 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
      base[(size_t)row * a.Ci_pad] = acc[i][r];
    }
  }
}

// dW[co][ci] = G^T (sum over slices dU[.][.][co][ci]) G.  Block = one co x 64 ci x 4 slice phases.
__global__ void __launch_bounds__(256) wino4_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                                 int Co, int Ci, int Co_pad, int Ci_pad, int n_slices) {
  __shared__ float red[3][36][64];
  const int cil = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int n_cic = (Ci + 63) / 64;
  const int co = blockIdx.x / n_cic, ci = (blockIdx.x % n_cic) * 64 + cil;
  float u[36];
#pragma unroll
  for (int f = 0; f < 36; ++f) u[f] = 0.f;
  if (ci < Ci) {
    for (int s = ph; s < n_slices; s += 4) {
      const float* p = ws + ((size_t)(s * 36) * Co_pad + co) * Ci_pad + ci;
#pragma unroll
      for (int f = 0; f < 36; ++f) u[f] += p[(size_t)f * Co_pad * Ci_pad];
    }
  }
  if (ph > 0) {
#pragma unroll
    for (int f = 0; f < 36; ++f) red[ph - 1][f][cil] = u[f];
  }
  __syncthreads();
  if (ph == 0 && ci < Ci) {
#pragma unroll
    for (int f = 0; f < 36; ++f) u[f] = ((u[f] + red[0][f][cil]) + red[1][f][cil]) + red[2][f][cil];
    // t[r][j] = sum_i G[i][r] u[i][j];  G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]
    float t[3][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const float u0 = u[0 * 6 + j], u1 = u[1 * 6 + j], u2 = u[2 * 6 + j], u3 = u[3 * 6 + j], u4 = u[4 * 6 + j],
                  u5 = u[5 * 6 + j];
      const float s12 = u1 + u2, d12 = u2 - u1, s34 = u3 + u4, d34 = u3 - u4;
      t[0][j] = 0.25f * u0 - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
      t[1][j] = (1.f / 6.f) * d12 + (1.f / 12.f) * d34;
      t[2][j] = -(1.f / 6.f) * s12 + (1.f / 6.f) * s34 + u5;
    }
    float* dst = dw + ((size_t)co * Ci + ci) * 9;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float s12 = t[r][1] + t[r][2], d12 = t[r][2] - t[r][1], s34 = t[r][3] + t[r][4], d34 = t[r][3] - t[r][4];
      dst[r * 3 + 0] = 0.25f * t[r][0] - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
      dst[r * 3 + 1] = (1.f / 6.f) * d12 + (1.f / 12.f) * d34;
      dst[r * 3 + 2] = -(1.f / 6.f) * s12 + (1.f / 6.f) * s34 + t[r][5];
    }
  }
}

namespace {
struct Wino4WgPlan {
  int Ci_pad, Co_pad, n_ci_tiles, n_co_tiles, nrh, nrw, nstages, sps, n_slices;
};

int wino4_wg_plan(int B, int Ci, int Co, int H, int W, Wino4WgPlan* p) {
  p->n_ci_tiles = cdiv(Ci, 32);
  p->n_co_tiles = cdiv(Co, 64);
  p->Ci_pad = p->n_ci_tiles * 32;
  p->Co_pad = p->n_co_tiles * 64;
  p->nrh = H / 4;
  p->nrw = W / 16;
  const long long ns = (long long)B * p->nrh * p->nrw;
  if (ns > 0x3fffffffLL) return SIVAE_ERR_RANGE;
  p->nstages = (int)ns;
  const int ntiles = p->n_ci_tiles * p->n_co_tiles;
  // one block per CU is resident (12 waves, ~158 KB LDS): aim at SIVAE_WG4_SLOTS blocks (default: one round of the CUs),
  // at least 24 stages per slice so that the 288 KB partial-dU write-out of a block stays small next to its MFMA work
  static int slots = 0;
  if (slots == 0) {
    const char* e = getenv("SIVAE_WG4_SLOTS");
    slots = e ? atoi(e) : sivae_num_cus();
    if (slots <= 0) slots = 256;
  }
  int n_slices = cdiv(slots, ntiles);
  const int max_slices = p->nstages / 24 > 0 ? p->nstages / 24 : 1;
  if (n_slices > max_slices) n_slices = max_slices;
  p->sps = cdiv(cdiv(p->nstages, n_slices), 3) * 3;  // whole stage triples
  p->n_slices = cdiv(p->nstages, p->sps);
  return SIVAE_OK;
}
}  // namespace

// maps the F(4x4,3x3) weight gradient takes (its stage is a 4 x 16 pixel strip of whole tiles)
extern "C" int sivae_conv2d_wino4_wgrad_supported(int H, int W) {
  return (H >= 4 && W >= 16 && (H % 4) == 0 && (W % 16) == 0) ? 1 : 0;
}

// does it beat the F(2x2,3x3) weight gradient for this launch?  One block per CU is resident: it needs (co, ci) tiles x
// slices of >= 24 stages for every CU
extern "C" int sivae_conv2d_wino4_wgrad_pays(int B, int Ci, int Co, int H, int W) {
  if (B <= 0 || Ci < 16 || Co < 16 || !sivae_conv2d_wino4_wgrad_supported(H, W)) return 0;
  const long long stages = (long long)B * (H / 4) * (W / 16);
  const long long tiles = (long long)cdiv(Ci, 32) * cdiv(Co, 64);
  return tiles * (stages / 24) >= sivae_num_cus() ? 1 : 0;
}

extern "C" size_t sivae_conv2d_wino4_wgrad_workspace_bytes(int B, int Ci, int Co, int H, int W) {
  Wino4WgPlan p;
  if (B <= 0 || Ci <= 0 || Co <= 0 || !sivae_conv2d_wino4_wgrad_supported(H, W)) return 0;
  if (wino4_wg_plan(B, Ci, Co, H, W, &p) != SIVAE_OK) return 0;
  return (size_t)p.n_slices * 36 * p.Co_pad * p.Ci_pad * sizeof(float);
}

// dw[Co][Ci][3][3] = weight gradient of y = conv3x3(x', w) given dy, x' = x or (pro_mean != NULL) LeakyReLU(BatchNorm(x))
// with per-segment statistics when seg_images > 0 (B = nseg * seg_images, nseg <= 2; sivae_conv2d_wino_wgrad_seg)
extern "C" int sivae_conv2d_wino4_wgrad(const float* x, const float* dy, float* dw, const float* pro_mean,
                                        const float* pro_invstd, const float* pro_gamma, const float* pro_beta,
                                        float pro_slope, int B, int Ci, int Co, int H, int W, int seg_images,
                                        void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!x || !dy || !dw || !workspace) return SIVAE_ERR_NULL;
  if (seg_images < 0 || (seg_images > 0 && B % seg_images != 0)) return SIVAE_ERR_SHAPE;
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  if (!sivae_conv2d_wino4_wgrad_supported(H, W)) return SIVAE_ERR_SHAPE;
  if (pro_mean && (!pro_invstd || !pro_gamma || !pro_beta)) return SIVAE_ERR_NULL;
  if (pro_mean && !(pro_slope >= 0.f && pro_slope <= 1.f)) return SIVAE_ERR_MODE;  // prologue uses max(v, v*slope)
  if ((((uintptr_t)x) & 15u) != 0 || (((uintptr_t)dy) & 15u) != 0) return SIVAE_ERR_SHAPE;  // 16-byte LDS-direct loads
  const long long hw = (long long)H * W;
  if ((long long)Ci * hw * 4 >= 0x7fffffffLL || (long long)Co * hw * 4 >= 0x7fffffffLL) return SIVAE_ERR_RANGE;
  Wino4WgPlan p;
  int rc = wino4_wg_plan(B, Ci, Co, H, W, &p);
  if (rc != SIVAE_OK) return rc;
  const size_t need = (size_t)p.n_slices * 36 * p.Co_pad * p.Ci_pad * sizeof(float);
  if (workspace_bytes < need) return SIVAE_ERR_WORKSPACE;
  Wino4WgArgs a;
  a.x = x;
  a.dy = dy;
  a.ws = static_cast<float*>(workspace);
  a.pro_mean = pro_mean;
  a.pro_invstd = pro_invstd;
  a.pro_gamma = pro_gamma;
  a.pro_beta = pro_beta;
  a.pro_slope = pro_slope;
  a.pro_seg_images = seg_images > 0 ? seg_images : B;
  a.pro_nseg = B / a.pro_seg_images;
  if (pro_mean && a.pro_nseg > 2) return SIVAE_ERR_SHAPE;
  a.B = B;
  a.Ci = Ci;
  a.Co = Co;
  a.H = H;
  a.W = W;
  a.Ci_pad = p.Ci_pad;
  a.Co_pad = p.Co_pad;
  a.nrh = p.nrh;
  a.nrw = p.nrw;
  a.nstages = p.nstages;
  a.sps = p.sps;
  a.n_co_tiles = p.n_co_tiles;
  a.n_ci_tiles = p.n_ci_tiles;
  const long long nblk = (long long)p.n_ci_tiles * p.n_co_tiles * p.n_slices;
  if (nblk > 0x7fffffffLL) return SIVAE_ERR_RANGE;
  a.nblk = (int)nblk;
  a.xcd_remap = sivae_xcd_remap();
  auto kern = pro_mean ? wino4_wgrad_kernel<true> : wino4_wgrad_kernel<false>;
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.xcd_remap ? (nblk + 7) / 8 * 8 : nblk)), dim3(G4_NT), 0, stream, a);
  rc = sivae_launch_status();
  if (rc != SIVAE_OK) return rc;
  const int n_cic = (Ci + 63) / 64;
  hipLaunchKernelGGL(wino4_wgrad_reduce_kernel, dim3((unsigned)(Co * n_cic)), dim3(256), 0, stream,
                     static_cast<const float*>(workspace), dw, Co, Ci, p.Co_pad, p.Ci_pad, p.n_slices);
  return sivae_launch_status();
}
