// EXPERIMENT (round 5; not compiled into the library): the third form of the six-product F(4x4,3x3) K loop — three row-pair
// phases per 16-channel step, fp32 V in LDS split at read time, software-pipelined MFMA / vector slices, every VMEM
// instruction of the loop issued through inline asm with hand-counted vmcnt.  A drop-in replacement of
// soft-intro-vae-pytorch_amd/csrc/conv_wino4_b6.hip (same entry points; no fused BatchNorm prologue: pro_mean must be NULL,
// an even number of steps per K slice): copy it over that file, run csrc/build.sh; tools/experiments/b6_phases_timing.py
// reads its -DB6_TIMING counters.  Correct (87 kernel checks at the fp32 kernel's tolerances) and at 0.87x of the fp32-MFMA
// kernel: per phase a SIMD drains its three waves in 1.8-2.0 us against a vector-issue floor of ~1.0 us — each wave's stream
// is latency-bound (dependent MFMAs issue every ~64 cycles, an LDS round trip per transform slice) and three waves per
// SIMD do not cover it.  Numbers: profiles/r5_conv_wino4_b6_experiments.txt.
// Winograd F(4x4, 3x3) stride-1 "same" convolution whose 36 frequency GEMMs run on the BF16 matrix pipe with FP32-exact
// products (round 5): every fp32 operand is split into three bf16 pieces by truncation (x = x1 + x2 + x3, 8 mantissa bits
// each, exact), and a product u*v is accumulated in fp32 from six of the nine piece products
//     u1 v3 + u2 v2 + u3 v1 + u1 v2 + u2 v1 + u1 v1            (the three dropped terms are <= 2^-24 |u v| together)
// on v_mfma_f32_32x32x16_bf16.  Measured against fp64 (tools/probes/probe_bf16x6_gemm.hip, profiles/r4_probe_bf16x6_accuracy.txt)
// the six-product form errs 2.4e-8 rms / 2.5e-7 max of sum|uv| — the figures of v_mfma_f32_32x32x2_f32 itself (2.7e-8 /
// 2.0e-7): the arithmetic contract of conv_wino4.hip (fp32 products, fp32 accumulation) is kept.  Why: on gfx950 the fp32
// matrix instruction runs at 1/16 of the bf16 rate AND occupies the vector lanes, so conv_wino4's K loop costs
// 2 x 4 608 matrix cycles + ~1 300 VALU cycles per 16 input channels and SIMD; six bf16 MFMAs per fp32 product are
// 108 x 32 = 3 456 cycles, and ordinary VALU instructions issue in the shadow of a bf16 MFMA.
//
// Same nn.Conv2d(k=3, s=1, p=1) (reference: soft_intro_vae/train_soft_intro_vae.py:56-61), same work split (a block = 12
// waves = 64 output channels x 32 tiles x 36 frequencies; wave (j, s) owns frequency COLUMN j of the 32-channel subtile s: 6
// accumulators of 32 x 32), same raw-halo staging and the same epilogue as conv_wino4.hip — read that file's header first.
//
// K loop (third form).  The first form was serial (all waves transform + split into LDS, barrier, all waves multiply):
// parity with the fp32 kernel — twelve waves in lock step expose every LDS latency and the matrix pipe idles during the
// transform (profiles/r5_conv_wino4_b6_experiments.txt).  Now a 16-channel step (one bf16 MFMA contracts 16 channels) is
// THREE PHASES, one per frequency ROW pair (1,2), (3,4), (0,5) — V = B^T d B with the ROW transform first:
//     s[i][c] = sum_r B^T[i][r] d[r][c]   (pairs share their partial sums: (d4-4d2) +- (d3-4d1), (d4-d2) +- 2(d3-d1))
//     V[i][j] = sum_c s[i][c] B[c][j]
// In phase p every wave
//   * multiplies the two frequencies (rows of pair p) of its column out of the fp32 V buffer the previous phase filled:
//     per frequency 8 ds_read_b32 (the 8 channels of the lane's K half), the exact three-way split + pair packing in
//     registers (44 VALU), six MFMAs; the 3 x 16-byte A pieces (U pre-split, MFMA-ready: pack_wino4_b6) sit in two
//     register slots that are refilled right after use with the NEXT phase's rows (a lead of one phase);
//   * and — eight of the twelve waves, the idle four (wave % 3 == p: one per SIMD) rotate — transforms row pair p + 1
//     (of the same step; in phase 2 of the next step) for its (tile, channel): four (six) patch rows of six values,
//     ~48 VALU, twelve conflict-free ds_write_b32 into the other V buffer.
// One LDS-only barrier per phase.  Transform, split and MFMAs of different waves overlap on every SIMD; nothing needs
// more than ~36 registers next to the 96 accumulators and the 24 of the A slots.
// LDS: V ping-pong 2 x 24 KB (a row pair = 12 frequencies x 16 channels x 32 tiles fp32), raw halo of TWO steps 2 x 48 KB
// (LDS-direct loads: step s + 1 is requested at the end of phase 0 of step s and first read in phase 2) = 144 KB; the
// epilogue's 48 KB exchange aliases the raw buffer of the finished step.  No fused BatchNorm prologue in this form.
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
  int ksl, sps;  // split-K: K slices, 16-channel steps per slice (even)
  long long slice_stride;
};

#define B6_RS 40
#define B6_PLANE 768
#define B6_XBUF (8 * B6_PLANE)     // raw halo of 8 channels: 6144 floats (24 KB)
#define B6_VBUF (12 * 16 * 32)     // transformed fp32 values of one column pair: [cc 2][i 6][ch 16][tile 32] = 6144 floats
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
//   1 no A-operand loads, 2 no transform work, 4 no halo requests, 8 no MFMAs, 16 no B-operand reads, 32 no split
#ifndef B6_ABLATE
#define B6_ABLATE 0
#endif
// Phase times (a -DB6_TIMING build only; tools/b6_timing.py): wave 0 accumulates the 100 MHz wall clock per phase in scalar
// registers and writes {phase 0, phase 1, phase 2, epilogue + set-up, steps, items} at the end.
#ifdef B6_TIMING
__device__ long long b6_dbg[256 * 8];
__device__ long long b6_dbg2[256 * 12 * 4];  // per block: summed stage times of wave 0 (split of row A | row A's MFMAs | row B's MFMAs + transform)
extern "C" int sivae_debug_b6_read(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(b6_dbg), sizeof(long long) * 256 * 8);
}
extern "C" int sivae_debug_b6_read2(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(b6_dbg2), sizeof(long long) * 256 * 12 * 4);
}
#define B6_CLK(V) const long long V = (long long)wall_clock64();
#else
#define B6_CLK(V)
#endif

__global__ void __launch_bounds__(B6_NT, 1) conv_wino4_b6_kernel(Wino4B6Args a) {
  constexpr int RS = B6_RS, PLANE = B6_PLANE, XBUF = B6_XBUF;
  // the two raw-halo sets and the V ping-pong are SEPARATE static arrays: hipcc orders an LDS read behind every in-flight
  // LDS-direct load it cannot prove disjoint (the requests of step s + 1 fill one set while the transforms read the other)
  __shared__ __attribute__((aligned(16))) float rawA[2 * XBUF];  // steps 0, 2, 4 ... of an item: channels 0-7 | 8-15
  __shared__ __attribute__((aligned(16))) float rawB[2 * XBUF];  // steps 1, 3, 5 ...
  __shared__ __attribute__((aligned(16))) float vq[2 * B6_VBUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int wj = wave % 6, ws = wave / 6;
  const int H = a.H, W = a.W, HW = H * W;
  const int wm3 = wave % 3;
  const int rk = wave & 7;  // transform rank: waves 8..11 redo the items of waves 0..3 (no wave-dependent branches: see B6_WORK)

  // ---- transform role (thread = tile tt x channel 2 rk + hh of the step): lane-constant parts of its LDS offsets
  //   raw offset = hh * PLANE + 4 (tt >> 3) * RS + 4 (tt & 7) + 4 [patch (0, 1)] + (rk >> 2) * XBUF + 2 (rk & 3) * PLANE
  //   V offset   = hh * 32 + tt + rk * 64
  const unsigned long long seam_m0 = 0x1010101010101010ull, seam_m5 = 0x0808080808080808ull;
  const int two_w = a.two ? 32 : a.W, two_mask = a.two ? 15 : -1, two_img = a.two ? a.Ci * a.H * a.W : 0;
  // ---- MFMA role: B operand = V[(ii*6 + wj) * 512 + (8 hh + e) * 32 + l31], ii = 0, 1 (the pair's rows), e = 0..7
  // ---- halo role (as conv_wino4: wave w fills third w % 3 of the planes w / 3 + 4 n of each 8-channel half)
  const int dsub = wm3, dpl0 = wave / 3;
  const int pg = dsub * 64 + lane, prow = pg / 10, pk = pg - prow * 10;
  const bool pvalid = pg < 180;

  const int n_items = a.n_items;
  const int nsteps = a.sps;
  const int n_cosub = a.Co_pad >> 5, nsteps_all = a.Ci_pad >> 4;
  // Every VMEM instruction of the K loop is issued through inline asm with hand-counted s_waitcnt vmcnt(N): hipcc (1) orders
  // the first LDS read after an LDS-direct load behind vmcnt(0) when it cannot tell the arrays apart (it could not here:
  // every phase began by waiting out the halo requests issued a moment earlier), and (2) cannot be told that a halo request
  // may stay outstanding across two phases while younger A-piece loads are consumed.  vmcnt counts IN ORDER; the order of
  // issue per wave and step is fixed (see B6_STEP), so the counts are constants.  Buffer descriptors for asm: int4.
  typedef int b6_i4 __attribute__((ext_vector_type(4)));
  const unsigned long long up_a_ = (unsigned long long)a.up;
  const b6_i4 ur4 = {(int)(unsigned)up_a_, (int)((up_a_ >> 32) & 0xffffull), (int)(unsigned)(36ull * a.Ci_pad * a.Co_pad * 6ull),
                     0x00020000};
  const unsigned va0 = (unsigned)lane * 16u;
  const unsigned ua_step = (unsigned)n_cosub * (18u * B6_ABLK);  // bytes per 16-channel step (of one frequency column)

  int item = blockIdx.x;
  if (a.xcd_group) item = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
  int b, r0, c0, co0, pt;
  int cbase = 0, kslice = 0;
  b6_i4 xr4;
  unsigned xo, ua_base;
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
    {                                                                    \
      const unsigned long long xa_ = (unsigned long long)(a.x + (size_t)b * a.Ci * HW); \
      xr4 = b6_i4{(int)(unsigned)xa_, (int)((xa_ >> 32) & 0xffffull), (int)((unsigned)(a.two ? 2 : 1) * (unsigned)(a.Ci * HW) * 4u), 0x00020000}; \
    }                                                                    \
    const int r = r0 - 1 + prow, c = c0 - 4 + 4 * pk;                    \
    xo = (pvalid && r >= 0 && r < H && c >= 0 && c < two_w)              \
             ? (unsigned)((c >> 4) * two_img + r * W + (c & two_mask)) * 4u : SIVAE_OOB; \
    ua_base = (unsigned)(((wj * nsteps_all + (cbase >> 4)) * n_cosub + (co0 >> 5) + ws)) * (18u * B6_ABLK); \
  }
  // the halo of step ST -> raw set RSET: four 16-byte LDS-direct loads per wave (planes dpl0, dpl0 + 4 of both halves;
  // out-of-image / padding groups receive 0; channels beyond Ci re-read the last one: their U is zero)
#define B6_DMA(ST, RSET)                                                 \
  if (!((B6_ABLATE & 4) && item >= 0)) _Pragma("unroll") for (int n_ = 0; n_ < 4; ++n_) { \
    const int ck = dpl0 + 4 * (n_ & 1);                                  \
    const int ci = cbase + (ST)*16 + 8 * (n_ >> 1) + ck;                 \
    const int cic = ci < a.Ci ? ci : a.Ci - 1;                           \
    const unsigned lds_ = (unsigned)(unsigned long long)(float __attribute__((address_space(3)))*)(RSET + (n_ >> 1) * XBUF + ck * PLANE + dsub * 256); \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" \
                 :: "s"(lds_), "v"(xo), "s"(xr4), "s"((unsigned)cic * (unsigned)HW * 4u) : "memory"); \
  }

  f32x16 acc[6];
  u32x4_t S0[3], S1[3];  // the A pieces of the two frequencies of the coming multiply (refilled right after use)
#if B6_ABLATE & 1
#pragma unroll
  for (int q_ = 0; q_ < 3; ++q_) S0[q_] = S1[q_] = u32x4_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#endif
#define B6_LOAD_A(SLOT, SO, I)                                           \
  if (!(B6_ABLATE & 1)) {                                                \
    asm volatile("buffer_load_dwordx4 %0, %3, %4, %5 offen\n\tbuffer_load_dwordx4 %1, %3, %4, %6 offen\n\t" \
                 "buffer_load_dwordx4 %2, %3, %4, %7 offen"              \
                 : "=&v"(SLOT[0]), "=&v"(SLOT[1]), "=&v"(SLOT[2])        \
                 : "v"(va0), "s"(ur4), "s"((SO) + (unsigned)(((I)*3 + 0)) * B6_ABLK), \
                   "s"((SO) + (unsigned)(((I)*3 + 1)) * B6_ABLK), "s"((SO) + (unsigned)(((I)*3 + 2)) * B6_ABLK)); \
  }
  // wait until at most N VMEM instructions of this wave are outstanding; tied to the slot whose pieces it makes valid, so
  // that the MFMAs reading them cannot be scheduled above it
#define B6_VMWAIT(N, SLOT) \
  asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(SLOT[0]), "+v"(SLOT[1]), "+v"(SLOT[2])::"memory");
#define B6_FENCE __builtin_amdgcn_sched_barrier(0);
  // Workgroup barrier that orders LDS traffic only (see conv_wino4.hip: __syncthreads() would wait out the halo requests)
#define B6_LDS_BARRIER asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#define B6_MF(I, AP, BP)                                                 \
  if (!(B6_ABLATE & 8)) acc[I] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, AP), \
                                                                     __builtin_bit_cast(bf16x8_t, BP), acc[I], 0, 0, 0);
  // the six products of one frequency, smallest terms first
#define B6_MMA(I, SLOT, BV)                                              \
  B6_MF(I, SLOT[0], BV[2]) B6_MF(I, SLOT[1], BV[1]) B6_MF(I, SLOT[2], BV[0]) \
  B6_MF(I, SLOT[0], BV[1]) B6_MF(I, SLOT[1], BV[0]) B6_MF(I, SLOT[0], BV[0])
  // four fp32 B values of the pair's row II (this wave's column): channels 8 hh + 4 HF .. + 3 of the lane's tile
#define B6_READ4(VP, II, HF, F)                                          \
  if (!(B6_ABLATE & 16)) { _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) F[e_] = (VP)[(II)*6 * 512 + (4 * (HF) + e_) * 32]; } \
  else { _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) F[e_] = 1.f; }
  // ... split exactly three ways and packed in pairs: BV[piece][2 HF + m] = (piece of F[2m], piece of F[2m+1]).  Four values
  // at a time: the split's temporaries stay at 12 registers
#define B6_SPLIT4(F, HF, BV)                                             \
  if (!(B6_ABLATE & 32)) {                                               \
    unsigned b2_[4], b3_[4];                                             \
    _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                   \
      const float r1_ = F[e_] - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, F[e_]) & 0xffff0000u); \
      b2_[e_] = __builtin_bit_cast(unsigned, r1_);                       \
      b3_[e_] = __builtin_bit_cast(unsigned, r1_ - __builtin_bit_cast(float, b2_[e_] & 0xffff0000u)); \
    }                                                                    \
    _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_) {                   \
      BV[0][2 * (HF) + m_] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, F[2 * m_ + 1]), __builtin_bit_cast(unsigned, F[2 * m_]), 0x07060302u); \
      BV[1][2 * (HF) + m_] = __builtin_amdgcn_perm(b2_[2 * m_ + 1], b2_[2 * m_], 0x07060302u); \
      BV[2][2 * (HF) + m_] = __builtin_amdgcn_perm(b3_[2 * m_ + 1], b3_[2 * m_], 0x07060302u); \
    }                                                                    \
  }
  // one patch row of six values (columns 0..5 of the tile's patch; P at patch column 1, 16-byte aligned)
#define B6_ROW6(P, X)                                                    \
  {                                                                      \
    const float4 q4_ = *reinterpret_cast<const float4*>(P);              \
    X[0] = (P)[-1];                                                      \
    X[5] = (P)[4];                                                       \
    X[1] = q4_.x; X[2] = q4_.y; X[3] = q4_.z; X[4] = q4_.w;              \
    if (a.two) { /* the seam between the two images is zero padding for both (tile columns 3 | 4) */ \
      asm volatile("v_cndmask_b32_e64 %0, %0, 0, %1" : "+v"(X[0]) : "s"(seam_m0)); \
      asm volatile("v_cndmask_b32_e64 %0, %0, 0, %1" : "+v"(X[5]) : "s"(seam_m5)); \
    }                                                                    \
  }
  // V[i][0..5] = s[i][.] B  (the column transform of one row i) -> Q[j * 512]
#define B6_TCOL(T, Q)                                                    \
  {                                                                      \
    const float A_ = fmaf(-4.f, T[2], T[4]), B_ = fmaf(-4.f, T[1], T[3]); \
    const float C_ = T[4] - T[2], D_ = T[3] - T[1];                      \
    (Q)[0 * 512] = fmaf(4.f, T[0], fmaf(-5.f, T[2], T[4]));              \
    (Q)[1 * 512] = A_ + B_;                                              \
    (Q)[2 * 512] = A_ - B_;                                              \
    (Q)[3 * 512] = fmaf(2.f, D_, C_);                                    \
    (Q)[4 * 512] = fmaf(-2.f, D_, C_);                                   \
    (Q)[5 * 512] = fmaf(4.f, T[1], fmaf(-5.f, T[3], T[5]));              \
  }
  // ---- the transform of row pair TQ (0: rows (1,2), 1: (3,4), 2: (0,5)) of this thread's (tile, channel), in five slices
  // that are issued BETWEEN the six MFMAs of a frequency (a wave's dependent MFMAs leave ~30 idle issue cycles each)
#define B6_T1(TQ, P)                                                     \
  if ((TQ) != 2) { B6_ROW6((P) + 2 * RS, xa_) B6_ROW6((P) + 4 * RS, xb_) } \
  else { B6_ROW6((P) + 0 * RS, xa_) B6_ROW6((P) + 2 * RS, xb_) }
#define B6_T2(TQ, P)                                                     \
  _Pragma("unroll") for (int c_ = 0; c_ < 6; ++c_)                       \
    sA_[c_] = (TQ) == 2 ? fmaf(4.f, xa_[c_], -5.f * xb_[c_]) : fmaf((TQ) == 0 ? -4.f : -1.f, xa_[c_], xb_[c_]); \
  if ((TQ) != 2) { B6_ROW6((P) + 1 * RS, xa_) B6_ROW6((P) + 3 * RS, xb_) } \
  else { B6_ROW6((P) + 4 * RS, xa_) B6_ROW6((P) + 1 * RS, xb_) }
  // (the (0,5) pair reads six rows: through TWO row buffers, in three rounds — a third buffer pushed the phase over its
  // register budget and hipcc spilled an accumulator tile around it)
#define B6_T3(TQ, P)                                                     \
  _Pragma("unroll") for (int c_ = 0; c_ < 6; ++c_) {                     \
    if ((TQ) == 2) {                                                     \
      sA_[c_] += xa_[c_];                                                \
      sB_[c_] = 4.f * xb_[c_];                                           \
    } else {                                                             \
      const float b_ = (TQ) == 0 ? fmaf(-4.f, xa_[c_], xb_[c_]) : fmaf(-2.f, xa_[c_], 2.f * xb_[c_]); \
      sB_[c_] = sA_[c_] - b_;                                            \
      sA_[c_] = sA_[c_] + b_;                                            \
    }                                                                    \
  }                                                                      \
  if ((TQ) == 2) { B6_ROW6((P) + 3 * RS, xa_) B6_ROW6((P) + 5 * RS, xb_) }
#define B6_T4(TQ)                                                        \
  if ((TQ) == 2) _Pragma("unroll") for (int c_ = 0; c_ < 6; ++c_) sB_[c_] = fmaf(-5.f, xa_[c_], sB_[c_]) + xb_[c_];
  // ---- one phase of a wave: rows IA, IB of its column out of V buffer VB, software-pipelined so that every MFMA is followed
  // by a slice of vector work: the split of row IB's operand under the MFMAs of row IA, the transform of row pair TQ (raw
  // set RSET -> V buffer VW; this wave's rank RK among the phase's transform waves, or none: TDO false) under those of
  // row IB.  Only the split of row IA has no MFMAs of its own wave to hide under.  The two A slots are refilled right
  // after use with rows NA, NB (the next phase's) of the U stream at byte offset NSO.
#define B6_WORK(VB, IA, IB, NSO, NA, NB, W1, W2, TQ, RSET, VW, RK)        \
  {                                                                      \
    /* lane-dependent LDS offsets are RECOMPUTED from the lane index every phase: kept in registers across the K loop   \
       hipcc spills them and reloads them from scratch at the top of a phase — a vmcnt wait that (vmcnt counts in order) \
       also waits for the halo requests issued a moment earlier */       \
    int ln_;                                                             \
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln_)); \
    const float* vp_ = vq + (VB)*B6_VBUF + wj * 512 + (ln_ >> 5) * 256 + (ln_ & 31); \
    u32x4_t bvA_[3], bvB_[3];                                            \
    B6_CLK(w0_)                                                          \
    {                                                                    \
      float f_[4];                                                       \
      B6_READ4(vp_, 0, 0, f_)                                            \
      B6_SPLIT4(f_, 0, bvA_)                                             \
      B6_READ4(vp_, 0, 1, f_)                                            \
      B6_SPLIT4(f_, 1, bvA_)                                             \
    }                                                                    \
    B6_FENCE                                                             \
    B6_CLK(w1_)                                                          \
    {                                                                    \
      float f_[4];                                                       \
      B6_VMWAIT(W1, S0)                                                  \
      B6_MF(IA, S0[0], bvA_[2]) B6_READ4(vp_, 1, 0, f_) B6_FENCE         \
      B6_MF(IA, S0[1], bvA_[1]) B6_SPLIT4(f_, 0, bvB_) B6_FENCE          \
      B6_MF(IA, S0[2], bvA_[0]) B6_READ4(vp_, 1, 1, f_) B6_FENCE         \
      B6_MF(IA, S0[0], bvA_[1]) B6_SPLIT4(f_, 1, bvB_) B6_FENCE          \
      B6_MF(IA, S0[1], bvA_[0]) B6_FENCE                                 \
      B6_MF(IA, S0[0], bvA_[0]) B6_FENCE                                 \
    }                                                                    \
    B6_CLK(w2_)                                                          \
    /* NO wave-dependent branch around this: a load inside an `if` makes hipcc lose count of the outstanding loads at the \
       join and turn the following waits into vmcnt(0) — every MFMA group then waited for the refill issued just before it \
       (2 us per phase instead of 0.9).  All twelve waves transform; the four "spare" ones of a phase redo the first four  \
       ranks' items (same values to the same addresses). */              \
    {                                                                    \
      const int tt_ = ln_ & 31, hh_ = ln_ >> 5;                          \
      const float* p_ = RSET + (hh_ * PLANE + 4 * (tt_ >> 3) * RS + 4 * (tt_ & 7) + 4 + ((RK) >> 2) * XBUF + 2 * ((RK)&3) * PLANE); \
      float* q_ = vq + (VW)*B6_VBUF + hh_ * 32 + tt_ + (RK)*64;          \
      float xa_[6], xb_[6], sA_[6], sB_[6];                              \
      B6_VMWAIT(W2, S1)                                                  \
      B6_MF(IB, S1[0], bvB_[2]) B6_T1(TQ, p_) B6_FENCE                   \
      B6_MF(IB, S1[1], bvB_[1]) B6_T2(TQ, p_) B6_FENCE                   \
      B6_MF(IB, S1[2], bvB_[0]) B6_T3(TQ, p_) B6_FENCE                   \
      /* (slot S0 is refilled here, once the transform's row registers are free) */ \
      B6_MF(IB, S1[0], bvB_[1]) B6_T4(TQ) B6_LOAD_A(S0, NSO, NA) B6_TCOL(sA_, q_) B6_FENCE \
      B6_MF(IB, S1[1], bvB_[0]) B6_TCOL(sB_, (q_ + 6 * 512)) B6_FENCE    \
      B6_MF(IB, S1[0], bvB_[0]) B6_FENCE                                 \
    }                                                                    \
    B6_LOAD_A(S1, NSO, NB)                                               \
    B6_FENCE                                                             \
    B6_WACC                                                              \
  }
  // the transform alone (the very first row pair of a block: no multiply to hide under yet)
#define B6_TWORK(TQ, RSET, VW, RK)                                       \
  if (!(B6_ABLATE & 2)) {                                                \
    int ln_;                                                             \
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln_)); \
    const int tt_ = ln_ & 31, hh_ = ln_ >> 5;                            \
    const float* p_ = RSET + (hh_ * PLANE + 4 * (tt_ >> 3) * RS + 4 * (tt_ & 7) + 4 + ((RK) >> 2) * XBUF + 2 * ((RK)&3) * PLANE); \
    float* q_ = vq + (VW)*B6_VBUF + hh_ * 32 + tt_ + (RK)*64;            \
    float xa_[6], xb_[6], sA_[6], sB_[6];                                \
    B6_T1(TQ, p_) B6_T2(TQ, p_) B6_T3(TQ, p_) B6_T4(TQ) B6_TCOL(sA_, q_) B6_TCOL(sB_, (q_ + 6 * 512)) \
  }
#ifdef B6_TIMING
  long long t_ph[3] = {0, 0, 0}, t_E = 0, n_st = 0, n_it = 0, t_w[3] = {0, 0, 0};
#define B6_WACC { B6_CLK(w3_) t_w[0] += w1_ - w0_; t_w[1] += w2_ - w1_; t_w[2] += w3_ - w2_; }
  long long c_last = (long long)wall_clock64();
  const long long sh0_ = (long long)clock64(), wl0_ = c_last;
#define B6_TACC { B6_CLK(c3_) t_E += c0_ - c_last; t_ph[0] += c1_ - c0_; t_ph[1] += c2_ - c1_; t_ph[2] += c3_ - c2_; c_last = c3_; ++n_st; }
#else
#define B6_TACC
#define B6_WACC
#endif
  // One 16-channel step ST (RC: the raw set holding its halo, RN: the other set; V buffer read in phase 0 = VB0).
  //   phase 0: multiply rows (1,2) out of V[VB0]; transform rows (3,4) of step ST (RC -> V[VB0^1]); A slots <- rows (3,4);
  //            then request the halo of step ST + 1 (the next item's first step at the end of an item) -> RN
  //   phase 1: multiply rows (3,4) out of V[VB0^1]; transform rows (0,5) of step ST (RC -> V[VB0]); A slots <- rows (0,5);
  //            wait for this wave's halo requests (one phase old; the six refill loads are younger) before the barrier
  //   phase 2: multiply rows (0,5) out of V[VB0]; transform rows (1,2) of step ST + 1 (RN -> V[VB0^1]); A slots <- rows (1,2)
  //            of the next step's U stream
#define B6_STEP(ST, RC, RN, VB0)                                         \
  {                                                                      \
    const unsigned so_ = ua_cur + (unsigned)(ST)*ua_step;                \
    const bool last_ = (ST) + 1 == nsteps;                               \
    B6_CLK(c0_)                                                          \
    if (last_ && has_next) B6_SETUP(next)                                \
    B6_WORK(VB0, 1, 2, so_, 3, 4, 3, 0, 1, RC, (VB0) ^ 1, rk)       \
    B6_FENCE                                                             \
    {                                                                    \
      const int dst_ = last_ ? 0 : (ST) + 1;                             \
      B6_DMA(dst_, RN)                                                   \
    }                                                                    \
    B6_FENCE                                                             \
    B6_LDS_BARRIER                                                       \
    B6_CLK(c1_)                                                          \
    B6_WORK((VB0) ^ 1, 3, 4, so_, 0, 5, 7, 4, 2, RC, VB0, rk)       \
    B6_FENCE                                                             \
    /* this wave's four halo requests (older than the six refill loads of this phase) have landed */ \
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                     \
    B6_LDS_BARRIER                                                       \
    B6_CLK(c2_)                                                          \
    {                                                                    \
      const unsigned nso_ = last_ ? ua_base : so_ + ua_step;             \
      B6_WORK(VB0, 0, 5, nso_, 1, 2, 3, 0, 0, RN, (VB0) ^ 1, rk)    \
    }                                                                    \
    B6_FENCE                                                             \
    B6_LDS_BARRIER                                                       \
    B6_TACC                                                              \
  }

  B6_SETUP(item)
  B6_DMA(0, rawA)
  unsigned ua_cur = ua_base;
  B6_LOAD_A(S0, ua_cur, 1)
  B6_LOAD_A(S1, ua_cur, 2)
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  B6_LDS_BARRIER
  // rows (1,2) of the first step (what phase 2 of a previous step would have left in V buffer 0)
  B6_TWORK(0, rawA, 0, rk)
  B6_FENCE
  B6_LDS_BARRIER
  for (;;) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int e_pt = pt, e_b = b, e_r0 = r0, e_c0 = c0, e_co0 = co0, e_ks = kslice;
    const int next = item + (int)gridDim.x;
    const bool has_next = next < n_items;
    for (int st = 0; st < nsteps; st += 2) {
      B6_STEP(st, rawA, rawB, 0)
      B6_STEP(st + 1, rawB, rawA, 1)
    }

    // ---- output transform: identical to conv_wino4.hip (acc[i][r]: frequency (i, wj), tile = l31, channel =
    // ws*32 + (r&3) + 8*(r>>2) + 4*hh).  The 48 KB exchange aliases rawB: the last step of an item (odd) read it in its
    // phases 0 and 1; the next request into it (step 1 of the next item) is issued in that item's first phase.
    {
      float* ex = rawB;
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
    d_[0] = t_ph[0]; d_[1] = t_ph[1]; d_[2] = t_ph[2]; d_[3] = t_E + ((long long)wall_clock64() - c_last); d_[4] = n_st; d_[5] = n_it;
    d_[6] = (long long)clock64() - sh0_; d_[7] = (long long)wall_clock64() - wl0_;  // shader cycles / 100 MHz ticks of the block
  }
  if (lane == 0 && blockIdx.x < 256) {
    long long* e_ = b6_dbg2 + ((int)blockIdx.x * 12 + wave) * 4;
    e_[0] = t_w[0]; e_[1] = t_w[1]; e_[2] = t_w[2]; e_[3] = n_st;
  }
#endif
#undef B6_SETUP
#undef B6_DMA
#undef B6_LOAD_A
#undef B6_VMWAIT
#undef B6_FENCE
#undef B6_LDS_BARRIER
#undef B6_MF
#undef B6_MMA
#undef B6_READ4
#undef B6_SPLIT4
#undef B6_WORK
#undef B6_T1
#undef B6_T2
#undef B6_T3
#undef B6_T4
#undef B6_ROW6
#undef B6_TCOL
#undef B6_TWORK
#undef B6_STEP
#undef B6_TACC
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
  if (pro_mean) return SIVAE_ERR_MODE;  // (no fused BatchNorm prologue in this kernel: conv_wino4.hip has it)
  (void)pro_invstd; (void)pro_gamma; (void)pro_beta; (void)pro_slope;
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
  // (the K loop is unrolled by step pairs — the two raw-halo sets alternate —: an even number of steps per slice)
  if (ksl < 1 || nsteps_all % ksl != 0 || ((nsteps_all / ksl) & 1)) return SIVAE_ERR_SHAPE;
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
  hipLaunchKernelGGL(conv_wino4_b6_kernel, dim3((unsigned)grid), dim3(B6_NT), 0, stream, a);
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
