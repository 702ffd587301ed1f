// Winograd F(4x4, 3x3) stride-1 "same" convolution whose 36 frequency GEMMs run on the BF16 matrix pipe with FP32-exact
// products (round 5): every fp32 operand is split into three bf16 pieces by truncation (x = x1 + x2 + x3, 8 mantissa bits
// each, exact), and a product u*v is accumulated in fp32 from six of the nine piece products
//     u1 v3 + u2 v2 + u3 v1 + u1 v2 + u2 v1 + u1 v1            (the three dropped terms are <= 2^-24 |u v| together)
// on v_mfma_f32_32x32x16_bf16.  Measured against fp64 (tools/probes/probe_bf16x6_gemm.hip, profiles/r4_probe_bf16x6_accuracy.txt)
// the six-product form errs 2.4e-8 rms / 2.5e-7 max of sum|uv| — the figures of v_mfma_f32_32x32x2_f32 itself (2.7e-8 /
// 2.0e-7): the arithmetic contract of conv_wino4.hip (fp32 products, fp32 accumulation) is kept.  Why: on gfx950 the fp32
// matrix instruction runs at 1/16 of the bf16 rate AND occupies the vector lanes, so conv_wino4's K loop costs
// 2 x 4 608 matrix cycles + ~1 300 VALU cycles per 16 input channels and SIMD; six bf16 MFMAs per fp32 product are
// 108 x 32 = 3 456 cycles.
//
// Same nn.Conv2d(k=3, s=1, p=1) (reference: soft_intro_vae/train_soft_intro_vae.py:56-61), same work split, same raw-halo
// staging, same epilogue as conv_wino4.hip — read that file's header first.  What differs is the K loop, which is SERIAL
// per 16 input channels ("step"; one bf16 MFMA contracts 16 channels):
//   T phase   thread (tile, channel PAIR, frequency-column pair) transforms two channels of its tile (B^T d B, 2 x 54 VALU as
//             in conv_wino4), splits the 2 x 12 transformed values into pieces (4 VALU each) and packs the two channels of a
//             piece into one dword (v_perm): 36 ds_write_b32 into Vp[piece][frequency][kg][m][tile] — consecutive lanes =
//             consecutive tiles = consecutive dwords (conflict-free); the pair index (kg, m) IS the k-slot of the MFMA's
//             B operand: lane (tile, kg) holds k = 8 kg + 2 m, + 1 in dword m.
//   barrier
//   M phase   wave (j, s) = frequency column j x 32-channel subtile s as in conv_wino4 (6 accumulators of 32 x 32): per
//             frequency 3 x 4 ds_read_b32 (B pieces; conflict-free), 3 x 16-byte buffer loads (A pieces: U pre-split and
//             stored MFMA-ready by pack_wino4_b6, 1 KB per (frequency, piece, subtile, step): perfectly coalesced), six
//             MFMAs, smallest terms first.  The A pieces travel through a three-slot register ring two frequencies ahead;
//             the first slot is requested late in the T phase (after its register peak) and lands under the barrier.
//   barrier   (+ this wave's halo requests of the next step have landed: s_waitcnt vmcnt)
// LDS: raw halo of one step 2 x 24 KB (the two 8-channel halves are separate static arrays, filled by LDS-direct loads
// during the M phase) + Vp 108 KB = 156 KB; the epilogue's 48 KB exchange aliases Vp.  One block of 12 waves per CU.
// The fused BatchNorm + LeakyReLU prologue takes its per-channel parameters through scalar loads (no room for a table).
#include "bf16_common.h"
#include "pack_batch.h"
#include <stdlib.h>

struct Wino4B6Args {
  const float* x;
  const void* up;  // pre-split U: [j 6][step Ci_pad/16][co-subtile Co_pad/32][i 6][piece 3][lane 64] x 16 bytes
  float* y;
  float* stats;  // [n_px_tiles][Co][2] or null
  const float* pro_mean;
  const float* pro_invstd;
  const float* pro_gamma;
  const float* pro_beta;
  float pro_slope;
  int pro_seg_images, pro_nseg;
  int B, Ci, Co, H, W;
  int Ci_pad, Co_pad;
  int nbh, nbw;
  int n_co_tiles;
  int accumulate;
  int n_items;
  int xcd_group;
  int two;       // 16 x 16 maps: a work item is a PAIR of images side by side (32 x 16 pixels)
  int ksl, sps;  // split-K: K slices, 16-channel steps per slice
  long long slice_stride;
};

#define B6_RS 40
#define B6_PLANE 768
#define B6_XBUF (8 * B6_PLANE)     // raw halo of 8 channels: 6144 floats (24 KB)
#define B6_FSTR 256                // dwords per (piece, frequency): [kg 2][m 4][tile 32]
#define B6_PSTR (36 * B6_FSTR)     // dwords per piece: 9216
#define B6_VP (3 * B6_PSTR)        // 27648 dwords = 108 KB
#define B6_NT 768
#define B6_TCO 64
#define B6_PXH 16
#define B6_PXW 32
#define B6_OOB16 0x80000000u       // (see conv_wino4.hip::W4_OOB16)
#define B6_ABLK 1024u              // bytes of one (frequency, piece) block of U: 64 lanes x 16 bytes

__device__ __forceinline__ void b6_store_f32x4(__amdgpu_buffer_rsrc_t r, float4 v, unsigned voff, unsigned soff) {
  f32x4 f = {v.x, v.y, v.z, v.w};
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f), r, (int)voff, (int)soff, 0);
}

// Timing ablations (compile with -DB6_ABLATE=<bits>; results are WRONG with any bit set — tools/b6_timing.py):
//   1 no A-operand loads, 2 no T phase, 4 no halo requests, 8 no MFMAs, 16 no B-operand reads
#ifndef B6_ABLATE
#define B6_ABLATE 0
#endif
#ifndef B6_PREFETCH
#define B6_PREFETCH 0
#endif
// Phase times (a -DB6_TIMING build only; tools/b6_timing.py): every wave reads the 100 MHz wall clock at the phase
// boundaries and keeps running totals in scalar registers — no memory traffic inside the loop; wave 0 of a block writes
// {T phases, M phases, epilogues + item set-up, steps, items} at the end.
#ifdef B6_TIMING
__device__ long long b6_dbg[256 * 8];
extern "C" int sivae_debug_b6_read(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(b6_dbg), sizeof(long long) * 256 * 8);
}
#define B6_CLK(V) const long long V = (long long)wall_clock64();
#else
#define B6_CLK(V)
#endif

typedef const float __attribute__((address_space(4))) * b6_cptr;  // constant address space: uniform loads become s_load

template <bool PRO>
__global__ void __launch_bounds__(B6_NT, 1) conv_wino4_b6_kernel(Wino4B6Args a) {
  constexpr int RS = B6_RS, PLANE = B6_PLANE, XBUF = B6_XBUF;
  // separate static arrays: the compiler orders an LDS read behind every in-flight LDS-direct load it cannot prove disjoint
  __shared__ __attribute__((aligned(16))) float raw0[XBUF];
  __shared__ __attribute__((aligned(16))) float raw1[XBUF];
  __shared__ __attribute__((aligned(16))) unsigned vp[B6_VP];
  // target of the halo PREFETCH (LDS-direct loads of the step after next whose only purpose is to pull the lines into L2
  // a whole step early; every wave writes the same 1 KB, nobody reads it)
  __shared__ __attribute__((aligned(16))) float sink[256];
#define RAWB(BUF) ((BUF) ? raw1 : raw0)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int wj = wave % 6, ws = wave / 6;
  const int H = a.H, W = a.W, HW = H * W;

  // ---- transform role: column pair tp, tile tt, channel pair tc = 4 kgT + mT of the step (channels 2 tc, 2 tc + 1)
  const int tp = wave >> 2;  // 0: columns (1,2)   1: columns (3,4)   2: columns (0,5)
  const int ti = (wave & 3) * 64 + lane, tt = ti & 31;
  const int kgT = (wave & 3) >> 1;           // (wave-uniform: which raw buffer this wave transforms)
  const int mT = ((wave & 1) << 1) | hh;     // pair inside the 8-channel half
  const int trb = (2 * mT) * PLANE + 4 * (tt >> 3) * RS + 4 * (tt & 7) + 4;  // patch column 1 of channel 2 mT
  const int jA = tp == 0 ? 1 : (tp == 1 ? 3 : 0), jB = tp == 0 ? 2 : (tp == 1 ? 4 : 5);
  const int tvb = (kgT * 4 + mT) * 32 + tt;
  const float t_al = tp == 0 ? -4.f : -1.f;
  const float t_be = tp == 0 ? 1.f : 2.f;
  const float t_ga = tp == 0 ? -4.f : -2.f;
  const bool pair05 = tp == 2;
  const unsigned long long seam_m0 = 0x1010101010101010ull, seam_m5 = 0x0808080808080808ull;
  const int two_w = a.two ? 32 : a.W, two_mask = a.two ? 15 : -1, two_img = a.two ? a.Ci * a.H * a.W : 0;
  // ---- MFMA role: B operand dwords vp[p * PSTR + (i*6 + wj) * FSTR + hh * 128 + m * 32 + l31]
  const int vrb = wj * B6_FSTR + hh * 128 + l31;

  // ---- halo role (as conv_wino4: wave w fills third w % 3 of the planes w / 3 + 4 n of each 8-channel half)
  const int dsub = wave % 3, dpl0 = wave / 3;
  const int pg = dsub * 64 + lane, prow = pg / 10, pk = pg - prow * 10;
  const bool pvalid = pg < 180;

  const int n_items = a.n_items;
  const int nsteps = a.sps;
  const int n_cosub = a.Co_pad >> 5, nsteps_all = a.Ci_pad >> 4;
  const __amdgpu_buffer_rsrc_t ursrc = make_rsrc(a.up, 36ull * a.Ci_pad * a.Co_pad * 6ull);
  const unsigned va0 = (unsigned)lane * 16u;
  const unsigned ua_step = (unsigned)n_cosub * (18u * B6_ABLK);  // bytes per 16-channel step (of one frequency column)

  int item = blockIdx.x;
  if (a.xcd_group) item = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
  // Item coordinates (B6_SETUP: the item whose halo the REAL requests fetch, whose U stream the M phase reads and — through
  // the e_* copies taken at its start — whose outputs the epilogue stores) and the light set the halo PREFETCH needs when
  // it has reached the next item (B6_SETUP_PF, from the second-last step of an item on: image, K-slice base, lane offset).
  int b, r0, c0, co0, pt;
  int cbase = 0, kslice = 0;
  __amdgpu_buffer_rsrc_t xrsrc;
  unsigned xo, ua_base;
  int pseg = 0;
  int b_p = 0, cbase_p = 0;
  unsigned xo_p = SIVAE_OOB;
#define B6_SETUP(ITEM)                                                   \
  {                                                                      \
    const int co_tile = (ITEM) % a.n_co_tiles;                           \
    const int iq_ = (ITEM) / a.n_co_tiles;                               \
    kslice = iq_ % a.ksl;                                                \
    pt = iq_ / a.ksl;                                                    \
    cbase = kslice * a.sps * 16;                                         \
    const int tbx = pt % a.nbw;                                          \
    const int t2 = pt / a.nbw;                                           \
    const int tby = t2 % a.nbh;                                          \
    b = t2 / a.nbh;                                                      \
    r0 = tby * B6_PXH;                                                   \
    c0 = tbx * B6_PXW;                                                   \
    co0 = co_tile * B6_TCO;                                              \
    b = a.two ? 2 * pt : b;                                              \
    xrsrc = make_rsrc(a.x + (size_t)b * a.Ci * HW, (unsigned long long)(a.two ? 2 : 1) * a.Ci * HW * 4ull); \
    const int r = r0 - 1 + prow, c = c0 - 4 + 4 * pk;                    \
    xo = (pvalid && r >= 0 && r < H && c >= 0 && c < two_w)              \
             ? (unsigned)((c >> 4) * two_img + r * W + (c & two_mask)) * 4u : SIVAE_OOB; \
    ua_base = (unsigned)(((wj * nsteps_all + (cbase >> 4)) * n_cosub + (co0 >> 5) + ws)) * (18u * B6_ABLK); \
    if (PRO) pseg = (b / a.pro_seg_images) * a.Ci + cbase;               \
  }
#define B6_SETUP_PF(ITEM)                                                \
  {                                                                      \
    const int iq_ = (ITEM) / a.n_co_tiles;                               \
    const int ptq_ = iq_ / a.ksl;                                        \
    cbase_p = (iq_ % a.ksl) * a.sps * 16;                                \
    const int t2 = ptq_ / a.nbw;                                         \
    b_p = a.two ? 2 * ptq_ : t2 / a.nbh;                                 \
    const int r = (t2 % a.nbh) * B6_PXH - 1 + prow, c = (ptq_ % a.nbw) * B6_PXW - 4 + 4 * pk; \
    xo_p = (pvalid && r >= 0 && r < H && c >= 0 && c < two_w)            \
               ? (unsigned)((c >> 4) * two_img + r * W + (c & two_mask)) * 4u : SIVAE_OOB; \
  }
  // halo of step ST of item set (XR, XO, CB): four 16-byte LDS-direct loads per wave (planes dpl0, dpl0 + 4 of both halves)
#define B6_DMA_TO(ST, XR, XO, CB, SINK)                                  \
  if (!((B6_ABLATE & 4) && item >= 0)) _Pragma("unroll") for (int n_ = 0; n_ < 4; ++n_) { \
    const int ck = dpl0 + 4 * (n_ & 1);                                  \
    const int ci = (CB) + (ST)*16 + 8 * (n_ >> 1) + ck;                  \
    const int cic = ci < a.Ci ? ci : a.Ci - 1;                           \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(                            \
        XR, (float __attribute__((address_space(3)))*)((SINK) ? sink : RAWB(n_ >> 1) + ck * PLANE + dsub * 256), 16, XO, \
        (unsigned)cic * (unsigned)HW * 4u, 0, 0);                        \
  }
#define B6_DMA(ST) B6_DMA_TO(ST, xrsrc, xo, cbase, false)
  // fused BatchNorm + LeakyReLU prologue on the four groups this thread requested (parameters by scalar loads: the
  // channel is wave-uniform); padded channels (ci >= Ci) become 0 like the zero padding
#define B6_FIXUP(ST, XO, PSEG, CB)                                       \
  {                                                                      \
    const float msk_ = (XO) != SIVAE_OOB ? 1.f : 0.f;                    \
    _Pragma("unroll") for (int n_ = 0; n_ < 4; ++n_) {                   \
      const int ck = dpl0 + 4 * (n_ & 1);                                \
      const int cl = (ST)*16 + 8 * (n_ >> 1) + ck;                       \
      const bool cok = (CB) + cl < a.Ci;                                 \
      const int ix = cok ? (PSEG) + cl : 0;                              \
      const int ic = cok ? (CB) + cl : 0;                                \
      const float mean_ = ((b6_cptr)a.pro_mean)[ix];                     \
      const float sc_ = cok ? ((b6_cptr)a.pro_invstd)[ix] * ((b6_cptr)a.pro_gamma)[ic] : 0.f; \
      const float be_ = cok ? ((b6_cptr)a.pro_beta)[ic] : 0.f;           \
      float4* q_ = reinterpret_cast<float4*>(RAWB(n_ >> 1) + ck * PLANE + dsub * 256 + lane * 4); \
      float4 v_ = *q_;                                                   \
      v_.x = fmaf(v_.x - mean_, sc_, be_);                               \
      v_.y = fmaf(v_.y - mean_, sc_, be_);                               \
      v_.z = fmaf(v_.z - mean_, sc_, be_);                               \
      v_.w = fmaf(v_.w - mean_, sc_, be_);                               \
      v_.x = fmaxf(v_.x, v_.x * a.pro_slope) * msk_;                     \
      v_.y = fmaxf(v_.y, v_.y * a.pro_slope) * msk_;                     \
      v_.z = fmaxf(v_.z, v_.z * a.pro_slope) * msk_;                     \
      v_.w = fmaxf(v_.w, v_.w * a.pro_slope) * msk_;                     \
      *q_ = v_;                                                          \
    }                                                                    \
  }

  f32x16 acc[6];
  u32x4_t A0[3], A1[3], A2[3];  // the three-slot ring of A pieces (one frequency each)
#if B6_ABLATE & 1
#pragma unroll
  for (int q_ = 0; q_ < 3; ++q_) A0[q_] = A1[q_] = A2[q_] = u32x4_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#endif
  // A pieces of frequency I of the 16-channel step at byte offset SO of this wave's U stream
#define B6_LOAD_A(SLOT, SO, I)                                           \
  if (!(B6_ABLATE & 1)) {                                                \
    SLOT[0] = buf_load_u32x4(ursrc, va0, (SO) + (unsigned)(((I)*3 + 0)) * B6_ABLK); \
    SLOT[1] = buf_load_u32x4(ursrc, va0, (SO) + (unsigned)(((I)*3 + 1)) * B6_ABLK); \
    SLOT[2] = buf_load_u32x4(ursrc, va0, (SO) + (unsigned)(((I)*3 + 2)) * B6_ABLK); \
  }
#define B6_FENCE __builtin_amdgcn_sched_barrier(0);
#define B6_LDS_BARRIER asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // B pieces of frequency I
#define B6_READB(I, BV)                                                  \
  if (!(B6_ABLATE & 16)) {                                               \
    const unsigned* p_ = vp + (I)*6 * B6_FSTR + vrb_;                    \
    _Pragma("unroll") for (int pc = 0; pc < 3; ++pc)                     \
      _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_) BV[pc][m_] = p_[pc * B6_PSTR + m_ * 32]; \
  }
#define B6_MF(I, AP, BP)                                                 \
  if (!(B6_ABLATE & 8)) acc[I] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, AP), __builtin_bit_cast(bf16x8_t, BP), \
                                                   acc[I], 0, 0, 0);
  // the six products of one frequency, smallest terms first
#define B6_MMA(I, SLOT, BV)                                              \
  B6_MF(I, SLOT[0], BV[2]) B6_MF(I, SLOT[1], BV[1]) B6_MF(I, SLOT[2], BV[0]) \
  B6_MF(I, SLOT[0], BV[1]) B6_MF(I, SLOT[1], BV[0]) B6_MF(I, SLOT[0], BV[0])

  // ---- transform pieces (conv_wino4's arithmetic): rows R0, R0 + 1 of one channel's patch -> column dot products.
  // Two rows at a time: with 96 accumulator registers live there are ~70 registers for everything else.  (Requesting all
  // twelve rows of a thread's two channels up front — one exposed LDS latency per phase instead of six: the twelve waves
  // leave the step's barrier in lock step — needs 16 registers more than there are: hipcc then spills one accumulator
  // tile around every T phase and the kernel runs at 0.67x; measured, profiles/r5_conv_wino4_b6_experiments.txt.)
#define B6_TROWS(P, R0, P05)                                             \
  {                                                                      \
    const float* p_ = (P) + (R0)*RS;                                     \
    float4 td_[2];                                                       \
    float te0_[2], te5_[2];                                              \
    _Pragma("unroll") for (int r = 0; r < 2; ++r) {                      \
      td_[r] = *reinterpret_cast<const float4*>(p_ + r * RS);            \
      if (P05) {                                                         \
        te0_[r] = p_[r * RS - 1];                                        \
        te5_[r] = p_[r * RS + 4];                                        \
        if (a.two) { /* the seam between the two images is zero padding for both (tile columns 3 | 4) */ \
          asm volatile("v_cndmask_b32_e64 %0, %0, 0, %1" : "+v"(te0_[r]) : "s"(seam_m0)); \
          asm volatile("v_cndmask_b32_e64 %0, %0, 0, %1" : "+v"(te5_[r]) : "s"(seam_m5)); \
        }                                                                \
      }                                                                  \
    }                                                                    \
    _Pragma("unroll") for (int r = 0; r < 2; ++r) {                      \
      if (P05) {                                                         \
        tA_[(R0) + r] = fmaf(4.f, te0_[r], fmaf(-5.f, td_[r].y, td_[r].w)); \
        tB_[(R0) + r] = fmaf(4.f, td_[r].x, fmaf(-5.f, td_[r].z, te5_[r])); \
      } else {                                                           \
        const float a_ = fmaf(t_al, td_[r].y, td_[r].w);                 \
        const float b_ = fmaf(t_ga, td_[r].x, t_be * td_[r].z);          \
        tA_[(R0) + r] = a_ + b_;                                         \
        tB_[(R0) + r] = a_ - b_;                                         \
      }                                                                  \
    }                                                                    \
  }
  // the six rows of one channel -> tA_[6], tB_[6]
#define B6_TSTAGE1(P, P05)                                               \
  B6_TROWS(P, 0, P05)                                                    \
  B6_FENCE                                                               \
  B6_TROWS(P, 2, P05)                                                    \
  B6_FENCE                                                               \
  B6_TROWS(P, 4, P05)                                                    \
  B6_FENCE
  // V[0..5][J] = B^T t of one column -> OUT[0..5]
#define B6_TCOL(T, OUT)                                                  \
  {                                                                      \
    const float A_ = fmaf(-4.f, T[2], T[4]), B_ = fmaf(-4.f, T[1], T[3]); \
    const float C_ = T[4] - T[2], D_ = T[3] - T[1];                      \
    OUT[0] = fmaf(4.f, T[0], fmaf(-5.f, T[2], T[4]));                    \
    OUT[1] = A_ + B_;                                                    \
    OUT[2] = A_ - B_;                                                    \
    OUT[3] = fmaf(2.f, D_, C_);                                          \
    OUT[4] = fmaf(-2.f, D_, C_);                                         \
    OUT[5] = fmaf(4.f, T[1], fmaf(-5.f, T[3], T[5]));                    \
  }
  // exact three-way split of the pair (X: channel 2 tc, Y: channel 2 tc + 1) of frequency (I, J): three packed dwords
#define B6_SPLIT_STORE(X, Y, I, J)                                       \
  {                                                                      \
    const unsigned xb_ = __builtin_bit_cast(unsigned, (X)), yb_ = __builtin_bit_cast(unsigned, (Y)); \
    const float xr_ = (X) - __builtin_bit_cast(float, xb_ & 0xffff0000u); \
    const float yr_ = (Y) - __builtin_bit_cast(float, yb_ & 0xffff0000u); \
    const unsigned xrb_ = __builtin_bit_cast(unsigned, xr_), yrb_ = __builtin_bit_cast(unsigned, yr_); \
    const float x3_ = xr_ - __builtin_bit_cast(float, xrb_ & 0xffff0000u); \
    const float y3_ = yr_ - __builtin_bit_cast(float, yrb_ & 0xffff0000u); \
    unsigned* q_ = vp + ((I)*6 + (J)) * B6_FSTR + tvb_;                  \
    q_[0] = __builtin_amdgcn_perm(yb_, xb_, 0x07060302u);                \
    q_[B6_PSTR] = __builtin_amdgcn_perm(yrb_, xrb_, 0x07060302u);        \
    q_[2 * B6_PSTR] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, y3_), __builtin_bit_cast(unsigned, x3_), 0x07060302u); \
  }
  // T phase of one step: raw halves (this wave's: kgT) -> Vp.  Channel a is transformed first (12 values kept), then channel
  // b column by column, each column's six pairs split and stored at once: the live set stays at ~42 registers next to the
  // 96 accumulators (the scheduling fences keep hipcc from interleaving the two channels, which spilled ~100 registers).
#define B6_TPHASE(P05)                                                   \
  {                                                                      \
    float va_[12];                                                       \
    const float* pa_ = RAWB(kgT) + trb_;                                 \
    {                                                                    \
      float tA_[6], tB_[6];                                              \
      B6_TSTAGE1(pa_, P05)                                               \
      B6_TCOL(tA_, va_)                                                  \
      B6_TCOL(tB_, (va_ + 6))                                            \
    }                                                                    \
    B6_FENCE                                                             \
    {                                                                    \
      float tA_[6], tB_[6];                                              \
      B6_TSTAGE1(pa_ + PLANE, P05)                                       \
      /* the register peak of the phase is behind us: request the A pieces of frequency 0 (they land under the rest  \
         of the phase and the barrier; held across the whole phase they were spilled) */ \
      B6_LOAD_A(A0, so_, 0)                                              \
      B6_FENCE                                                           \
      {                                                                  \
        float vb_[6];                                                    \
        B6_TCOL(tA_, vb_)                                                \
        _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) B6_SPLIT_STORE(va_[i_], vb_[i_], i_, jA) \
      }                                                                  \
      B6_FENCE                                                           \
      {                                                                  \
        float vb_[6];                                                    \
        B6_TCOL(tB_, vb_)                                                \
        _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) B6_SPLIT_STORE(va_[6 + i_], vb_[i_], i_, jB) \
      }                                                                  \
    }                                                                    \
  }
  // One 16-channel step ST of the current item.  On entry: raw0 / raw1 hold the step's halo (published).  The next step's
  // halo (the next item's first step at the end of an item) is requested right after the T phase's barrier.
#define B6_STEP(ST, P05)                                                 \
  {                                                                      \
    u32x4_t bv_[3], bw_[3];                                              \
    B6_CLK(c0_)                                                          \
    const unsigned so_ = ua_cur + (unsigned)(ST)*ua_step;                \
    /* laundered once per step: otherwise hipcc hoists the ~110 LDS addresses of a step (base + constant) out of the K \
       loop as loop invariants, spills them, and reloads one from scratch in front of every ds_read / ds_write */ \
    int vrb_ = vrb, tvb_ = tvb, trb_ = trb;                              \
    asm volatile("" : "+v"(vrb_), "+v"(tvb_), "+v"(trb_));               \
    if (!(B6_ABLATE & 2)) B6_TPHASE(P05) else B6_LOAD_A(A0, so_, 0)      \
    B6_FENCE                                                             \
    B6_LOAD_A(A1, so_, 1)                                                \
    B6_LOAD_A(A2, so_, 2)                                                \
    B6_FENCE                                                             \
    B6_LDS_BARRIER /* Vp complete; raw0 / raw1 free */                   \
    B6_CLK(c1_)                                                          \
    B6_FENCE                                                             \
    B6_READB(0, bv_)                                                     \
    B6_FENCE                                                             \
    const bool last_ = (ST) + 1 == nsteps;                               \
    if ((ST) + 2 == nsteps && has_next) B6_SETUP_PF(next)                \
    if (last_ && has_next) B6_SETUP(next)                                \
    const int dst_ = last_ ? 0 : (ST) + 1;                               \
    B6_DMA(dst_)                                                         \
    xo_f = xo;                                                           \
    pseg_f = pseg;                                                       \
    cb_f = cbase;                                                        \
    st_f = dst_;                                                         \
    B6_FENCE                                                             \
    /* the B pieces are double-buffered: frequency i + 1 is read while the MFMAs of frequency i run (the twelve waves \
       leave the barrier in lock step: with one buffer every wave sat out the LDS latency with an idle matrix pipe) */ \
    B6_READB(1, bw_)                                                     \
    B6_FENCE                                                             \
    B6_MMA(0, A0, bv_)                                                   \
    B6_FENCE                                                             \
    B6_LOAD_A(A0, so_, 3)                                                \
    B6_READB(2, bv_)                                                     \
    B6_FENCE                                                             \
    B6_MMA(1, A1, bw_)                                                   \
    B6_FENCE                                                             \
    B6_LOAD_A(A1, so_, 4)                                                \
    B6_READB(3, bw_)                                                     \
    B6_FENCE                                                             \
    B6_MMA(2, A2, bv_)                                                   \
    B6_FENCE                                                             \
    B6_LOAD_A(A2, so_, 5)                                                \
    B6_READB(4, bv_)                                                     \
    B6_FENCE                                                             \
    B6_MMA(3, A0, bw_)                                                   \
    B6_FENCE                                                             \
    B6_READB(5, bw_)                                                     \
    B6_FENCE                                                             \
    B6_MMA(4, A1, bv_)                                                   \
    B6_FENCE                                                             \
    B6_MMA(5, A2, bw_)                                                   \
    B6_FENCE                                                             \
    /* PREFETCH: the halo of the step after next (the next item's from the second-last step on) is requested into the \
       sink — the real request a step later finds the lines in L2 instead of paying an HBM latency inside the M phase.  \
       These four loads stay outstanding across the barrier (vmcnt(4)): the real requests, older, have landed. */ \
    if (B6_PREFETCH) {                                                   \
      const bool wrap_ = (ST) + 2 >= nsteps;                             \
      const int ps_ = wrap_ ? (ST) + 2 - nsteps : (ST) + 2;              \
      /* (at the last step B6_SETUP has already moved the R set on to the next item: P == R then) */ \
      const int bq_ = (wrap_ && !last_) ? b_p : b;                       \
      const __amdgpu_buffer_rsrc_t xr_ =                                 \
          make_rsrc(a.x + (size_t)bq_ * a.Ci * HW, (unsigned long long)(a.two ? 2 : 1) * a.Ci * HW * 4ull); \
      const unsigned xq_ = (wrap_ && !last_) ? xo_p : xo;                \
      const int cq_ = (wrap_ && !last_) ? cbase_p : cbase;               \
      B6_DMA_TO(ps_, xr_, xq_, cq_, true)                                \
      __builtin_amdgcn_s_waitcnt(0x0F74);                                \
    } else {                                                             \
      __builtin_amdgcn_s_waitcnt(0x0F70);                                \
    }                                                                    \
    B6_LDS_BARRIER                                                       \
    B6_CLK(c2_)                                                          \
    B6_TACC                                                              \
    if (PRO) {                                                           \
      B6_FIXUP(st_f, xo_f, pseg_f, cb_f)                                 \
      B6_LDS_BARRIER                                                     \
    }                                                                    \
  }

  B6_SETUP(item)
  B6_DMA(0)
  unsigned ua_cur = ua_base;
  __builtin_amdgcn_s_waitcnt(0x0F70);
  B6_LDS_BARRIER
  unsigned xo_f = xo;
  int pseg_f = pseg, cb_f = cbase, st_f = 0;
  if (PRO) {
    B6_FIXUP(0, xo_f, pseg_f, cb_f)
    B6_LDS_BARRIER
  }
#ifdef B6_TIMING
  long long t_T = 0, t_M = 0, t_E = 0, n_st = 0, n_it = 0;
  long long c_last = (long long)wall_clock64();
#define B6_TACC { t_E += c0_ - c_last; t_T += c1_ - c0_; t_M += c2_ - c1_; c_last = c2_; ++n_st; }
#else
#define B6_TACC
#endif
  for (;;) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int e_pt = pt, e_b = b, e_r0 = r0, e_c0 = c0, e_co0 = co0, e_ks = kslice;
    const int next = item + (int)gridDim.x;
    const bool has_next = next < n_items;
    if (pair05) {
      for (int st = 0; st < nsteps; ++st) B6_STEP(st, true)
    } else {
      for (int st = 0; st < nsteps; ++st) B6_STEP(st, false)
    }

    // ---- output transform: identical to conv_wino4.hip (acc[i][r]: frequency (i, wj), tile = l31, channel =
    // ws*32 + (r&3) + 8*(r>>2) + 4*hh); the 48 KB exchange aliases Vp (every wave is past its last read of it)
    {
      float* ex = reinterpret_cast<float*>(vp);
      const __amdgpu_buffer_rsrc_t yrsrc =
          make_rsrc(a.y + (size_t)e_ks * a.slice_stride + (size_t)e_b * a.Co * HW,
                    (unsigned long long)(a.two ? 2 : 1) * a.Co * HW * 4ull);
      float ssum[3] = {0.f, 0.f, 0.f}, ssq[3] = {0.f, 0.f, 0.f};
      int lane_;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_));
      const int ty_ = (lane_ >> 3) & 3, hh_ = lane_ >> 5;
      const int tx_ = a.two ? (lane_ & 3) : (lane_ & 7);
      const unsigned img_off = a.two ? (unsigned)((lane_ >> 2) & 1) * (unsigned)(a.Co * HW) * 4u : 0u;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float m0 = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r], m4 = acc[4][r], m5 = acc[5][r];
        const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
        acc[0][r] = m0 + s12 + s34;
        acc[1][r] = d12 + 2.f * d34;
        acc[2][r] = s12 + 4.f * s34;
        acc[3][r] = d12 + 8.f * d34 + m5;
      }
      unsigned off0[3];
#pragma unroll
      for (int qi = 0; qi < 3; ++qi) {
        const int q = wave + 12 * qi;
        const int chn = e_co0 + (q >> 4) * 32 + (q & 3) + 8 * ((q & 15) >> 2) + 4 * hh_;
        off0[qi] = chn < a.Co ? (unsigned)((chn * H + e_r0 + 4 * ty_) * W + e_c0 + 4 * tx_) * 4u + img_off : B6_OOB16;
      }
      float4 held[3];  // store-data lifetime: see conv_wino4.hip
      held[0] = held[1] = held[2] = make_float4(0.f, 0.f, 0.f, 0.f);
#define B6_KEEP(V) asm volatile("" ::"v"((V).x), "v"((V).y), "v"((V).z), "v"((V).w));
#pragma unroll
      for (int ar = 0; ar < 4; ++ar) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ex[((wj * 2 + ws) * 16 + r) * 64 + lane_] = acc[ar][r];
        __syncthreads();
        const unsigned row_off = (unsigned)(ar * W) * 4u;
#pragma unroll
        for (int qi = 0; qi < 3; ++qi) {
          const int q = wave + 12 * qi;
          if (q < 32) {
            const int s = q >> 4, r = q & 15;
            float z[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) z[j] = ex[((j * 2 + s) * 16 + r) * 64 + lane_];
            float4 o;
            o.x = z[0] + (z[1] + z[2]) + (z[3] + z[4]);
            o.y = (z[1] - z[2]) + 2.f * (z[3] - z[4]);
            o.z = (z[1] + z[2]) + 4.f * (z[3] + z[4]);
            o.w = (z[1] - z[2]) + 8.f * (z[3] - z[4]) + z[5];
            if (a.accumulate) {
              const float4 old = buf_load_f32x4(yrsrc, off0[qi], row_off);
              o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
            }
            held[qi] = o;
            b6_store_f32x4(yrsrc, held[qi], off0[qi], row_off);
            if (qi > 0) B6_KEEP(held[qi - 1])
            ssum[qi] += (o.x + o.y) + (o.z + o.w);
            ssq[qi] += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
          }
        }
        __syncthreads();
        B6_KEEP(held[1]) B6_KEEP(held[2])
      }
#undef B6_KEEP
      if (a.stats != nullptr) {
#pragma unroll
        for (int qi = 0; qi < 3; ++qi) {
          const int q = wave + 12 * qi;
          const float s_ = half_wave_sum_hi(ssum[qi]);
          const float q_ = half_wave_sum_hi(ssq[qi]);
          if (q < 32) {
            const int chn = e_co0 + (q >> 4) * 32 + (q & 3) + 8 * ((q & 15) >> 2) + 4 * hh_;
            if ((lane_ & 31) == 31 && chn < a.Co) {
              float* dst = a.stats + ((size_t)e_pt * a.Co + chn) * 2;
              dst[0] = s_;
              dst[1] = q_;
            }
          }
        }
      }
    }
#ifdef B6_TIMING
    ++n_it;
#endif
    if (!has_next) break;
    item = next;
    ua_cur = ua_base;
  }
#ifdef B6_TIMING
  if (tid == 0 && blockIdx.x < 256) {
    long long* d_ = b6_dbg + (int)blockIdx.x * 8;
    d_[0] = t_T; d_[1] = t_M; d_[2] = t_E + ((long long)wall_clock64() - c_last); d_[3] = n_st; d_[4] = n_it;
  }
#endif
#undef B6_SETUP
#undef B6_SETUP_PF
#undef B6_DMA_TO
#undef B6_DMA
#undef B6_FIXUP
#undef B6_LOAD_A
#undef B6_FENCE
#undef B6_LDS_BARRIER
#undef B6_READB
#undef B6_MF
#undef B6_MMA
#undef B6_TROWS
#undef B6_TSTAGE1
#undef B6_TCOL
#undef B6_SPLIT_STORE
#undef B6_TPHASE
#undef B6_STEP
#undef RAWB
}

// ---- weight transform U = G g G^T (the fp32 arithmetic of pack_wino4_body, conv_wino4.hip), split into three bf16 pieces
// by truncation and stored MFMA-ready: [j][step][co-subtile][i][piece][lane = (co & 31) + 32 kg] x 8 bf16 (ci = 16 step +
// 8 kg + 0..7).  One thread = one (co, 8-channel group, j): 8 weights in, 6 x 3 x 16 bytes out.
//   mode 0 (forward): g = w[n][k]          mode 1 (dgrad): g = flip180(w[k][n])
__device__ __forceinline__ void pack_wino4_b6_body(const float* __restrict__ w, unsigned char* __restrict__ up, int Ci,
                                                   int mode, int kdim, int ndim, int kpad, int npad, size_t idx0_,
                                                   const size_t stride_) {
  const float G[6][3] = {{0.25f, 0.f, 0.f},
                         {-1.f / 6.f, -1.f / 6.f, -1.f / 6.f},
                         {-1.f / 6.f, 1.f / 6.f, -1.f / 6.f},
                         {1.f / 24.f, 1.f / 12.f, 1.f / 6.f},
                         {1.f / 24.f, -1.f / 12.f, 1.f / 6.f},
                         {0.f, 0.f, 1.f}};
  const int kg_n = kpad >> 3, nsteps = kpad >> 4, n_cosub = npad >> 5;
  const size_t total = (size_t)6 * kg_n * npad;  // (j, 8-channel group, n)
  for (size_t idx = idx0_; idx < total; idx += stride_) {
    const int n = (int)(idx % npad);
    const int kgq = (int)((idx / npad) % kg_n);
    const int j = (int)(idx / ((size_t)npad * kg_n));
    float u[6][8];  // U[i][j] of the 8 channels
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = kgq * 8 + e;
      float g[3][3];
      const bool ok = k < kdim && n < ndim;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float v = 0.f;
          if (ok) v = (mode == 0) ? w[((size_t)n * Ci + k) * 9 + r * 3 + c] : w[((size_t)k * Ci + n) * 9 + (2 - r) * 3 + (2 - c)];
          g[r][c] = v;
        }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        float gg[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) gg[c] = G[i][0] * g[0][c] + G[i][1] * g[1][c] + G[i][2] * g[2][c];
        u[i][e] = gg[0] * G[j][0] + gg[1] * G[j][1] + gg[2] * G[j][2];
      }
    }
    const int step = kgq >> 1, kg = kgq & 1, sub = n >> 5, lane = (n & 31) + 32 * kg;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      u32x4_t p1, p2, p3;
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        unsigned q1[2], q2[2], q3[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float v = u[i][2 * e2 + h];
          const unsigned b1 = __builtin_bit_cast(unsigned, v) & 0xffff0000u;
          const float r1 = v - __builtin_bit_cast(float, b1);
          const unsigned b2 = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
          const float r2 = r1 - __builtin_bit_cast(float, b2);
          q1[h] = b1;
          q2[h] = b2;
          q3[h] = __builtin_bit_cast(unsigned, r2);
        }
        p1[e2] = (q1[0] >> 16) | (q1[1] & 0xffff0000u);
        p2[e2] = (q2[0] >> 16) | (q2[1] & 0xffff0000u);
        p3[e2] = (q3[0] >> 16) | (q3[1] & 0xffff0000u);
      }
      unsigned char* dst = up + ((((size_t)(j * nsteps + step) * n_cosub + sub) * 6 + i) * 3) * B6_ABLK + (size_t)lane * 16;
      *reinterpret_cast<u32x4_t*>(dst) = p1;
      *reinterpret_cast<u32x4_t*>(dst + B6_ABLK) = p2;
      *reinterpret_cast<u32x4_t*>(dst + 2 * B6_ABLK) = p3;
    }
  }
}

__global__ void __launch_bounds__(256) pack_wino4_b6_kernel(const float* __restrict__ w, unsigned char* __restrict__ up,
                                                            int Ci, int mode, int kdim, int ndim, int kpad, int npad) {
  pack_wino4_b6_body(w, up, Ci, mode, kdim, ndim, kpad, npad, (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)gridDim.x * 256);
}

__global__ void __launch_bounds__(256) pack_wino4_b6_batch_kernel(const SivaePackJob* __restrict__ jobs,
                                                                  const unsigned short* __restrict__ block_job) {
  const SivaePackJob j = jobs[block_job[blockIdx.x]];
  pack_wino4_b6_body(j.w, reinterpret_cast<unsigned char*>(j.dst), j.Ci, j.mode, j.kdim, j.ndim, j.kpad, j.npad,
                     (size_t)(blockIdx.x - j.blk0) * 256 + threadIdx.x, (size_t)j.nblk * 256);
}

static inline int b6_kpad(int k) { return ((k + 31) / 32) * 32; }  // (conv_wino4's padding: an even number of steps)
static inline int b6_npad(int n) { return ((n + B6_TCO - 1) / B6_TCO) * B6_TCO; }

extern "C" size_t sivae_pack_wino4_b6_weight_bytes(int Co, int Ci, int mode) {
  if (Co <= 0 || Ci <= 0 || (mode != 0 && mode != 1)) return 0;
  const int kdim = mode == 0 ? Ci : Co, ndim = mode == 0 ? Co : Ci;
  return (size_t)36 * b6_kpad(kdim) * b6_npad(ndim) * 6;
}

extern "C" int sivae_pack_wino4_b6_weight(const float* w, void* up, int Co, int Ci, int mode, hipStream_t stream) {
  if (!w || !up) return SIVAE_ERR_NULL;
  if (Co <= 0 || Ci <= 0) return SIVAE_ERR_SHAPE;
  if (mode != 0 && mode != 1) return SIVAE_ERR_MODE;
  if (((uintptr_t)up & 15u) != 0) return SIVAE_ERR_SHAPE;
  const int kdim = mode == 0 ? Ci : Co, ndim = mode == 0 ? Co : Ci;
  const int kpad = b6_kpad(kdim), npad = b6_npad(ndim);
  int nb = cdiv((long long)6 * (kpad >> 3) * npad, 256);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(pack_wino4_b6_kernel, dim3(nb), dim3(256), 0, stream, w, reinterpret_cast<unsigned char*>(up), Ci, mode,
                     kdim, ndim, kpad, npad);
  return sivae_launch_status();
}

static inline int b6_supported_map(int H, int W) {
  if (H == 16 && W == 16) return 2;
  return (H >= 16 && W >= 32 && (H % B6_PXH) == 0 && (W % B6_PXW) == 0) ? 1 : 0;
}
static inline long long b6_px_tiles(int B, int H, int W) {
  return (H == 16 && W == 16) ? B / 2 : (long long)B * (H / B6_PXH) * (W / B6_PXW);
}

// the K slices of the fp32 kernel's split-K plan (sivae_conv2d_wino4_splitk) in 16-channel steps: S slices of
// Ci_pad / 16 / S steps each
extern "C" int sivae_conv2d_wino4_splitk(int B, int Ci, int Co, int H, int W);

// y[B][Co][H][W] (+)= conv3x3(x', U): same contract as sivae_conv2d_wino4_fwd_pro / _fwd_splitk (conv_wino4.hip) with the
// pre-split operand of sivae_pack_wino4_b6_weight; pro_mean == NULL: no prologue.  ksl > 1: y is the [ksl][B][Co][H][W]
// partial-sum workspace.
static int wino4_b6_impl(const float* x, const void* up, float* y, const float* pro_mean, const float* pro_invstd,
                         const float* pro_gamma, const float* pro_beta, float pro_slope, float* stats_partial, int B,
                         int Ci, int Co, int H, int W, int accumulate, int seg_images, hipStream_t stream, int ksl) {
  if (!x || !up || !y) return SIVAE_ERR_NULL;
  if (pro_mean && (!pro_invstd || !pro_gamma || !pro_beta)) return SIVAE_ERR_NULL;
  if (pro_mean && !(pro_slope >= 0.f && pro_slope <= 1.f)) return SIVAE_ERR_MODE;
  if (seg_images < 0 || (seg_images > 0 && B % seg_images != 0)) return SIVAE_ERR_SHAPE;
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  const int sup = b6_supported_map(H, W);
  if (!sup) return SIVAE_ERR_SHAPE;
  if (sup == 2 && ((B & 1) || (seg_images & 1))) return SIVAE_ERR_SHAPE;
  if (((uintptr_t)y & 15u) != 0 || ((uintptr_t)up & 15u) != 0) return SIVAE_ERR_SHAPE;
  const long long hw = (long long)H * W;
  if ((long long)Ci * hw * 4 >= 0x7fffffffLL || (long long)Co * hw * 4 >= 0x7fffffffLL) return SIVAE_ERR_RANGE;
  Wino4B6Args a;
  a.x = x;
  a.up = up;
  a.y = y;
  a.stats = stats_partial;
  a.pro_mean = pro_mean;
  a.pro_invstd = pro_invstd;
  a.pro_gamma = pro_gamma;
  a.pro_beta = pro_beta;
  a.pro_slope = pro_slope;
  a.pro_seg_images = seg_images > 0 ? seg_images : B;
  a.pro_nseg = B / a.pro_seg_images;
  a.B = B;
  a.Ci = Ci;
  a.Co = Co;
  a.H = H;
  a.W = W;
  a.Ci_pad = b6_kpad(Ci);
  a.Co_pad = b6_npad(Co);
  if (36ull * a.Ci_pad * a.Co_pad * 6ull >= 0xffffffffull) return SIVAE_ERR_RANGE;
  a.two = sup == 2 ? 1 : 0;
  a.nbh = a.two ? 1 : H / B6_PXH;
  a.nbw = a.two ? 1 : W / B6_PXW;
  a.n_co_tiles = a.Co_pad / B6_TCO;
  a.accumulate = accumulate;
  const int nsteps_all = a.Ci_pad / 16;
  if (ksl < 1 || nsteps_all % ksl != 0) return SIVAE_ERR_SHAPE;
  a.ksl = ksl;
  a.sps = nsteps_all / ksl;
  a.slice_stride = ksl > 1 ? (long long)B * Co * hw : 0;
  const long long nitems = b6_px_tiles(B, H, W) * a.n_co_tiles * ksl;
  if (nitems > 0x7fffffffLL) return SIVAE_ERR_RANGE;
  a.n_items = (int)nitems;
  const int cus = sivae_num_cus();
  const int grid = nitems < cus ? (int)nitems : cus;
  // Block order.  xcd_group: the co-tile siblings of a pixel tile run on ONE XCD and share the halo in its L2 (conv_wino4's
  // order); plain order: block b takes item b, i.e. co-tile (b mod n_co_tiles) on XCD (b mod 8) — the blocks of one XCD walk
  // the SAME co-tile's U slab in step and share it in that L2.  This kernel streams 221 KB of U pieces against 48 KB of halo
  // per 16 channels and block, so sharing U is what matters once U no longer fits an L2 (SIVAE_B6_XCD_GROUP=0/1 forces it).
  static int xg = -2;
  if (xg == -2) {
    const char* e = getenv("SIVAE_B6_XCD_GROUP");
    xg = e ? atoi(e) : -1;
  }
  const bool u_fits_l2 = 36ull * a.Ci_pad * a.Co_pad * 6ull <= (2ull << 20);
  (void)u_fits_l2;  // (measured: the plain order loses 4-20 % on the 128 ... 512-channel layers: the halo sharing wins)
  const bool want_group = xg >= 0 ? xg != 0 : true;
  a.xcd_group = (sivae_xcd_remap() && want_group && a.n_co_tiles > 1 && !(grid & 7)) ? 1 : 0;
  if (pro_mean)
    hipLaunchKernelGGL(conv_wino4_b6_kernel<true>, dim3((unsigned)grid), dim3(B6_NT), 0, stream, a);
  else
    hipLaunchKernelGGL(conv_wino4_b6_kernel<false>, dim3((unsigned)grid), dim3(B6_NT), 0, stream, a);
  return sivae_launch_status();
}

// Forward / data gradient (mode-1 pack) with the optional fused producer BatchNorm + LeakyReLU (pro_mean != NULL;
// seg_images > 0: segmented batch, pro_mean / pro_invstd are [B / seg_images][Ci]).  Maps: sivae_conv2d_wino4_supported.
extern "C" int sivae_conv2d_wino4_b6_fwd(const float* x, const void* up, float* y, const float* pro_mean,
                                         const float* pro_invstd, const float* pro_gamma, const float* pro_beta,
                                         float pro_slope, float* stats_partial, int B, int Ci, int Co, int H, int W,
                                         int accumulate, int seg_images, hipStream_t stream) {
  return wino4_b6_impl(x, up, y, pro_mean, pro_invstd, pro_gamma, pro_beta, pro_slope, stats_partial, B, Ci, Co, H, W,
                       accumulate, seg_images, stream, 1);
}

__global__ void __launch_bounds__(64) wino4_b6_splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ y,
                                                                    float* __restrict__ stats, int S, int HW,
                                                                    size_t slice_stride, int accumulate) {
  const int bc = blockIdx.x;  // b * C + c
  const size_t base = (size_t)bc * HW;
  float s = 0.f, q = 0.f;
  for (int p = threadIdx.x; p < HW; p += 64) {
    float v = accumulate ? y[base + p] : 0.f;
    for (int k = 0; k < S; ++k) v += part[(size_t)k * slice_stride + base + p];
    y[base + p] = v;
    s += v;
    q += v * v;
  }
  if (stats != nullptr) {
    s = wave_sum(s);
    q = wave_sum(q);
    if (threadIdx.x == 0) {
      stats[(size_t)bc * 2 + 0] = s;
      stats[(size_t)bc * 2 + 1] = q;
    }
  }
}

// split-K form: the slice count is sivae_conv2d_wino4_splitk(...) (the plan of the fp32 kernel: same workspace size,
// sivae_conv2d_wino4_splitk_workspace_bytes; same per-image statistics rows when S > 1)
extern "C" int sivae_conv2d_wino4_b6_fwd_splitk(const float* x, const void* up, float* y, const float* pro_mean,
                                                const float* pro_invstd, const float* pro_gamma, const float* pro_beta,
                                                float pro_slope, float* stats_partial, int B, int Ci, int Co, int H, int W,
                                                int accumulate, int seg_images, void* workspace, size_t workspace_bytes,
                                                hipStream_t stream) {
  const int S = sivae_conv2d_wino4_splitk(B, Ci, Co, H, W);
  if (S < 0) return S;
  if (S == 1)
    return wino4_b6_impl(x, up, y, pro_mean, pro_invstd, pro_gamma, pro_beta, pro_slope, stats_partial, B, Ci, Co, H, W,
                         accumulate, seg_images, stream, 1);
  if (!y || !workspace) return SIVAE_ERR_NULL;
  if (workspace_bytes < (size_t)S * B * Co * H * W * sizeof(float)) return SIVAE_ERR_WORKSPACE;
  if (((uintptr_t)workspace & 15u) != 0) return SIVAE_ERR_SHAPE;
  float* part = reinterpret_cast<float*>(workspace);
  const int rc = wino4_b6_impl(x, up, part, pro_mean, pro_invstd, pro_gamma, pro_beta, pro_slope, nullptr, B, Ci, Co, H, W,
                               0, seg_images, stream, S);
  if (rc != SIVAE_OK) return rc;
  hipLaunchKernelGGL(wino4_b6_splitk_reduce_kernel, dim3((unsigned)(B * Co)), dim3(64), 0, stream, part, y, stats_partial, S,
                     H * W, (size_t)B * Co * H * W, accumulate);
  return sivae_launch_status();
}

// ---- batched packing (pack_batch.h)
int sivae_packjob_wino4_b6(SivaePackJob* j, int Co, int Ci, int mode) {
  j->kdim = mode == 0 ? Ci : Co;
  j->ndim = mode == 0 ? Co : Ci;
  j->kpad = b6_kpad(j->kdim);
  j->npad = b6_npad(j->ndim);
  j->total = (unsigned long long)6 * (j->kpad >> 3) * j->npad;
  return SIVAE_OK;
}
void sivae_packbatch_wino4_b6(const SivaePackJob* jobs, const unsigned short* block_job, int nblocks, hipStream_t s) {
  hipLaunchKernelGGL(pack_wino4_b6_batch_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, jobs, block_job);
}
