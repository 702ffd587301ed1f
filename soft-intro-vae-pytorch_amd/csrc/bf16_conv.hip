// bf16 implicit-GEMM stride-1 "same" convolution for gfx950 on v_mfma_f32_32x32x16_bf16 (fp32 accumulate): forward
// and — fed the flipped/transposed weight pack — data gradient of the nn.Conv2d(k in {1,3,5}) layers of
// soft_intro_vae/train_soft_intro_vae.py:51-61,89,159 in the build-defined bf16 mode of config 3 (BASELINE.json
// configs[2]).  The reference has no mixed precision; the mode's contract is: bf16 activation / gradient storage in the
// blocked layout of bf16_common.h, bf16 operands, fp32 accumulation, fp32 BatchNorm statistics, fp32 master weights.
//
//   D[co][pixel] = sum_{tap, ci} W[co][tap][ci] * X[pixel + tap][ci]
//
// MFMA roles: A = weights (row = output channel, 8 consecutive input channels per lane), B = activations (col = pixel,
// the same 8 input channels) — one ds_read_b128 per operand, every tap a 16-byte-aligned shift inside the halo tile.
// LDS per block:  ws[TAPS][CKS][2][TCO]  16-byte vectors: the weight slab of the current 16*CKS-channel chunk, already
//                                        in operand order in HBM (sivae_bf16_pack_conv_weight) -> a straight copy
//                 xs[2*CKS][plane]       16-byte vectors: zero-padded halo tile of the chunk's 8-channel blocks
// Accumulator register r of a lane is output channel (r&3) + 8*(r>>2) + 4*(lane>>5) of pixel lane&31, so the four
// registers of a group are 4 consecutive channels of one pixel = one 8-byte store into the blocked output; lanes l and
// l+32 complete the pixel's 16-byte vector.
//
// Fusions (runtime flags unless noted): producer BatchNorm + LeakyReLU applied while the halo tile is staged (PRO),
// nearest-2x upsample addressing of x, bias, y += result, per-channel {sum, sumsq} partials of the ROUNDED output for
// the consumer BatchNorm, fp32 NCHW output (OUTF32: Decoder.predict feeds the fp32 loss kernels).
#include "bf16_common.h"
#include "pack_batch.h"
#include <stdlib.h>
#include <type_traits>

struct Bf16ConvArgs {
  const void* x;   // bf16 blocked [B][Cib][Hs][Ws][8]
  const void* wp;  // packed bf16 [n_co_tiles][nchunks][TAPS][CKS][2][TCO][8]
  void* y;         // bf16 blocked [B][Cob][H][W][8], or float [B][Co][H][W] (OUTF32)
  const float* bias;
  const float* pro_mean;
  const float* pro_invstd;
  const float* pro_gamma;
  const float* pro_beta;
  float pro_slope;
  float* stats;  // [n_px_tiles][Co][2] or null
  int B, Ci, Co, H, W;
  int Cib, Cob;  // 8-channel blocks of x / y in storage
  int tb_log2, th_log2, tw_log2;
  int ntb, nth, ntw;
  int n_co_tiles, nchunks;
  int accumulate, upsample;
  // split-K (small grids: the 512-channel 8x8 / 4x4 layers are 64-256 blocks each walking 32 chunks): block
  // (slice s, base block) accumulates chunks [s*chunks_per_split, ...) and writes fp32 partials (NCHW, the OUTF32
  // epilogue) to y + s*split_stride floats; bf16_splitk_reduce_kernel sums the slices
  int nblk_base, chunks_per_split;
  long long split_stride;
  // fastdiv magics (common.h) of the halo tile's plane / row length / row count: the staging map of a thread is five
  // (vector -> channel block, image, row, column) decompositions — as run-time integer divisions they were 400 of the
  // kernel's 820 set-up instructions (set by launch_cfg)
  unsigned magic_plane, magic_lw, magic_lh;
  // 1: y is [B][Cob][H/2][W/2][8] and receives the 2x2 BLOCK SUMS of the conv's outputs (3x3 kernels, H and W even) —
  // the data gradient of a conv whose input is read through nearest-2x upsample addressing (adjoint of nn.Upsample,
  // train_soft_intro_vae.py:155) without the full-resolution tensor ever being written
  int poolsum = 0;
};


// compile-time loop: f(std::integral_constant<int, I>{}) for I in [I0, N)
template <int I0, int N, typename F>
__device__ __forceinline__ void ad_steps(F&& f) {
  if constexpr (I0 < N) {
    f(std::integral_constant<int, I0>{});
    ad_steps<I0 + 1, N>(f);
  }
}

// KS x KW taps (KW = KS except for the kw-packed 5x1 form of the RGB-side layers, ks code 51)
// AD ("A direct", the 3x3 kernels): the weight operands do not pass through LDS.  The pack is in operand order, so a
// lane's A vector of (chunk, tap, k-step, m) is ONE 16-byte load at a lane-constant offset + a scalar offset, and every
// block of a co tile reads the same slab (L2-resident).  Per wave a ring of RD = 3 k-steps keeps those loads in flight
// behind the MFMAs (refilled right after the slot's MFMAs; hipcc must be fenced or it sinks the refill to its use); the
// MFMAs of a k-step run pixel-tile-major, and each B fragment is re-read for the next k-step as soon as its WM MFMAs are
// issued (one register set, (WN - 1) * WM MFMAs of slack).  LDS carries the halo tile only, in TWO buffers: chunk ch + 1
// is written in front of the last k-step's MFMAs of chunk ch — one barrier per chunk.  Against the staged form: 36 KB of
// ds_write_b128 (13 LDS cycles each) per chunk and block and a third of the operand reads gone; 3x3 layers 6-13 % faster.
// VMEM returns in order: every load (the next chunk's halo tile too) has to land within RD k-steps of the ring's next
// wait, and the halo loads are therefore issued unconditionally (through an empty window behind the last chunk) — a load
// under a branch turns every later wait into vmcnt(0).
// Timing-only ablations (wrong results): -DAD_ABL_NOA / NOB / NOX / NOSTAGE / NOBAR, tools/build_ad_ablations.sh.  With
// ALL of them (a bare MFMA loop) the 512 -> 512 @ 16x16 layer runs at 1.58 PF/s (the sustained clock under dense bf16 MFMA
// work, not issue, bounds it); the full kernel reaches 1.34 (profiles/r6_bf16_conv_ad_ablations.txt).
template <int KS, int WM, int WN, int WVM, int WVN, int CKS, int MAXV, bool PRO, bool OUTF32, int MINW, int KW = KS,
          bool AD = false>
__global__ void __launch_bounds__(WVM* WVN * 64, MINW) bf16_conv_kernel(Bf16ConvArgs a) {
  constexpr int P = KS / 2, PW = KW / 2;
  constexpr int NT = WVM * WVN * 64;
  constexpr int TCO = WVM * WM * 32;
  constexpr int TAPS = KS * KW;
  constexpr int NCB = 2 * CKS;
  constexpr int NW = TAPS * CKS * 2 * TCO;  // 16-byte weight vectors per chunk
  constexpr int NWQ = (NW + NT - 1) / NT;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4_t* ws = reinterpret_cast<u32x4_t*>(smem_raw);
  u32x4_t* xs = AD ? ws : ws + NW;  // (AD: no weight slab in LDS)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int wvm = wave / WVN, wvn = wave % WVN;

  const int TW = 1 << a.tw_log2, TH = 1 << a.th_log2, TB = 1 << a.tb_log2;
  const int LW = TW + 2 * PW, LH = TH + 2 * P;
  const int plane = TB * LH * LW;
  const int nvec = NCB * plane;
  const int H = a.H, W = a.W, HW = H * W;
  const int Hs = a.upsample ? (H >> 1) : H, Ws = a.upsample ? (W >> 1) : W;
  const int HWs = Hs * Ws;
  float* ps = reinterpret_cast<float*>(xs + (AD ? 2 : 1) * nvec);  // [Cib][16]: 8 scales, 8 shifts per channel block (PRO); AD: two halo buffers

  const int slice = (int)blockIdx.x / a.nblk_base;
  const int bid = (int)blockIdx.x - slice * a.nblk_base;
  const int kc0 = slice * a.chunks_per_split;
  const int kc1 = kc0 + a.chunks_per_split < a.nchunks ? kc0 + a.chunks_per_split : a.nchunks;
  const int co_tile = bid % a.n_co_tiles;
  const int pt = bid / a.n_co_tiles;
  const int tw_i = pt % a.ntw;
  const int t2 = pt / a.ntw;
  const int th_i = t2 % a.nth;
  const int tb_i = t2 / a.nth;
  const int b0 = tb_i << a.tb_log2, r0 = th_i << a.th_log2, c0 = tw_i << a.tw_log2;
  const int co0 = co_tile * TCO;

  int nb_here = a.B - b0;
  if (nb_here > TB) nb_here = TB;
  const __amdgpu_buffer_rsrc_t xrsrc =
      make_rsrc(reinterpret_cast<const unsigned char*>(a.x) + (size_t)b0 * a.Cib * HWs * 16,
                (unsigned long long)nb_here * a.Cib * HWs * 16ull);
  const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(a.wp, (unsigned long long)NW * 16ull);
  const unsigned char* wbase = reinterpret_cast<const unsigned char*>(a.wp);

  if (PRO) {
    for (int c = tid; c < a.Cib * 8; c += NT) {
      float sc = 0.f, sh = 0.f;
      if (c < a.Ci) {
        sc = a.pro_invstd[c] * a.pro_gamma[c];
        sh = a.pro_beta[c] - a.pro_mean[c] * sc;
      }
      ps[(c >> 3) * 16 + (c & 7)] = sc;
      ps[(c >> 3) * 16 + 8 + (c & 7)] = sh;
    }
  }

  // ---- staging map of the halo tile: this thread owns vectors tid + p*NT of [NCB][plane]
  unsigned xo[MAXV];
  int xcb[MAXV];
#pragma unroll
  for (int p = 0; p < MAXV; ++p) {
    const int v = tid + p * NT;
    unsigned off = SIVAE_OOB;
    int cbl = 0;
    if (v < nvec) {
      cbl = (int)fastdiv((unsigned)v, a.magic_plane);
      const int pos = v - cbl * plane;
      const int t = (int)fastdiv((unsigned)pos, a.magic_lw);
      const int cc = pos - t * LW;
      const int tb = (int)fastdiv((unsigned)t, a.magic_lh);
      const int rr = t - tb * LH;
      const int r = r0 + rr - P, c = c0 + cc - PW;
      if (tb < nb_here && r >= 0 && r < H && c >= 0 && c < W) {
        const int rs = a.upsample ? (r >> 1) : r, cs = a.upsample ? (c >> 1) : c;
        off = ((((unsigned)tb * a.Cib + cbl) * Hs + rs) * Ws + cs) * 16u;
      }
    }
    xo[p] = off;
    xcb[p] = cbl;
  }
  unsigned wo[NWQ];
#pragma unroll
  for (int q = 0; q < NWQ; ++q) {
    const int idx = tid + q * NT;
    wo[q] = (NW % NT == 0 || idx < NW) ? (unsigned)idx * 16u : SIVAE_OOB;
  }

  const int a_base = hh * TCO + wvm * WM * 32 + l31;
  int b_base[WN];
#pragma unroll
  for (int n = 0; n < WN; ++n) {
    const int m_pix = (wvn * WN + n) * 32 + l31;
    const int cc = m_pix & (TW - 1);
    const int rr = (m_pix >> a.tw_log2) & (TH - 1);
    const int tb = m_pix >> (a.tw_log2 + a.th_log2);
    b_base[n] = hh * plane + (tb * LH + rr) * LW + cc;
  }

  f32x16 acc[WM][WN];
  if constexpr (!AD) {  // (AD: the first k-step's MFMAs take C = 0 and define the accumulators)
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int n = 0; n < WN; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  }

  u32x4_t xr[MAXV];
  if constexpr (AD) {
    constexpr int NS = TAPS * CKS;  // k-steps (MFMA groups) per chunk, in the order (ks, kh, kw)
    constexpr int RD = 3;           // ring depth in k-steps
    static_assert(NS % RD == 0, "the ring slots are compile-time indices of the unrolled chunk body");
    const __amdgpu_buffer_rsrc_t wr_t = make_rsrc(wbase + (size_t)co_tile * a.nchunks * (size_t)NW * 16,
                                                  (unsigned long long)a.nchunks * NW * 16ull);
    const __amdgpu_buffer_rsrc_t xnull = make_rsrc(a.x, 0ull);  // (every offset out of range: zeros, no traffic)
    const unsigned a_voff = (unsigned)a_base * 16u;
    u32x4_t ar[RD][WM];
    u32x4_t bq[WN];
#define SIVAE_AD_LOAD_A(SLOT, CH, S)                                                                      \
  {                                                                                                       \
    const unsigned so_ = ((unsigned)(CH) * (unsigned)NW + (unsigned)(S) * (unsigned)(2 * TCO)) * 16u;     \
    _Pragma("unroll") for (int m = 0; m < WM; ++m) ar[SLOT][m] = buf_load_u32x4(wr_t, a_voff + m * 512, so_); \
  }
#define SIVAE_AD_LOAD_X(RS, CH)                                                                           \
  {                                                                                                       \
    const unsigned xsoff = (unsigned)(CH) * (unsigned)(NCB * 16) * (unsigned)HWs;                         \
    _Pragma("unroll") for (int p = 0; p < MAXV; ++p) xr[p] = buf_load_u32x4(RS, xo[p], xsoff);            \
  }
#define SIVAE_AD_READ_B(N, S)                                                                             \
  {                                                                                                       \
    constexpr int ks_ = (S) / TAPS, tap_ = (S) % TAPS, kh_ = tap_ / KW, kw_ = tap_ % KW;                  \
    bq[N] = xs[xb + b_base[N] + ks_ * 2 * plane + kh_ * LW + kw_];                                        \
  }
    // halo tile of chunk CH (in xr) -> LDS buffer DST, the producer's BatchNorm + LeakyReLU applied on the way (PRO)
#define SIVAE_AD_STAGE_X(DST, CH)                                                                         \
  {                                                                                                       \
    _Pragma("unroll") for (int p = 0; p < MAXV; ++p) {                                                    \
      const int v = tid + p * NT;                                                                         \
      u32x4_t q = xr[p];                                                                                  \
      if (PRO) {                                                                                          \
        float f[8];                                                                                       \
        unpack8(q, f);                                                                                    \
        const float* pp = ps + ((CH) * NCB + xcb[p]) * 16;                                                \
        const bool in = xo[p] != SIVAE_OOB;                                                               \
        _Pragma("unroll") for (int e = 0; e < 8; ++e)                                                     \
            f[e] = in ? lrelu01(f[e] * pp[e] + pp[8 + e], a.pro_slope) : 0.f;                             \
        q = pack8(f);                                                                                     \
      }                                                                                                   \
      if (v < nvec) xs[(DST) + v] = q;                                                                    \
    }                                                                                                     \
  }
    // Two halo buffers: chunk ch + 1 is written (in front of the last k-step's MFMAs, by then its loads have had the whole
    // chunk to land) while chunk ch is still being read — ONE barrier per chunk, and the ds_write_b128s sit in the shadow of
    // the queued MFMAs instead of between two barriers.
    SIVAE_AD_LOAD_X(xrsrc, kc0)
#pragma unroll
    for (int s = 0; s < RD; ++s) SIVAE_AD_LOAD_A(s, kc0, s)
    if (PRO) __syncthreads();
    SIVAE_AD_STAGE_X(0, kc0)
    __syncthreads();
    // One chunk.  FIRST: the slice's first chunk (every slice has one), peeled: its first k-step's MFMAs take the inline
    // constant 0 as C — hipcc otherwise zeroes the 128 accumulators TWICE in front of the loop (254 v_movs, once for the
    // initial value and once for the loop-carried copy: ~1 000 cycles of a 4-chunk block's ~9 000-cycle K loop).
    auto chunk_body = [&](const int ch, auto FIRST_) {
      constexpr bool FIRST = decltype(FIRST_)::value;
      const int xb = ((ch - kc0) & 1) ? nvec : 0, xbn = nvec - xb;
      // the next chunk's halo tile: always issued (a conditional load would make every later wait a full drain); behind
      // the last chunk it goes through the empty window
#ifndef AD_ABL_NOX
      SIVAE_AD_LOAD_X((ch + 1 < kc1 ? xrsrc : xnull), ch + 1)
#endif
      const int chn = ch + 1 < a.nchunks ? ch + 1 : ch;  // (the ring runs RD k-steps ahead, past the slice's end too)
#pragma unroll
      for (int n = 0; n < WN; ++n) SIVAE_AD_READ_B(n, 0)
      ad_steps<0, NS>([&](auto S_) {
        constexpr int S = decltype(S_)::value;
#ifndef AD_ABL_NOSTAGE
        if constexpr (S == NS - 1) {
          if (ch + 1 < kc1) SIVAE_AD_STAGE_X(xbn, ch + 1)
        }
#endif
        // pixel-tile-major: a B fragment is dead after its WM MFMAs and is re-read for the next k-step at once — one
        // register set, and (WN - 1) * WM MFMAs between the read and its first use
#pragma unroll
        for (int n = 0; n < WN; ++n) {
#pragma unroll
          for (int m = 0; m < WM; ++m) {
            if constexpr (FIRST && S == 0) {
              const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ar[S % RD][m]),
                                                                  __builtin_bit_cast(bf16x8_t, bq[n]), zero, 0, 0, 0);
            } else {
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ar[S % RD][m]),
                                                                  __builtin_bit_cast(bf16x8_t, bq[n]), acc[m][n], 0, 0, 0);
            }
          }
#ifndef AD_ABL_NOB
          if constexpr (S + 1 < NS) SIVAE_AD_READ_B(n, S + 1)
#endif
        }
#ifndef AD_ABL_NOA
        if constexpr (S + RD < NS)
          SIVAE_AD_LOAD_A(S % RD, ch, S + RD)
        else
          SIVAE_AD_LOAD_A(S % RD, chn, S + RD - NS)
#endif
        // the order above is the order wanted: WM MFMAs, one LDS read, ..., then the WM refill loads (without the fences hipcc
        // sinks the refill to just in front of its use — the slot is dead in between, so that is "free" for register
        // pressure — and the L2 round trip is exposed)
#pragma unroll
        for (int n = 0; n < WN; ++n) {
          __builtin_amdgcn_sched_group_barrier(0x008, WM, 0);
          if constexpr (S + 1 < NS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x020, WM, 0);
        __builtin_amdgcn_sched_barrier(0);
      });
#ifndef AD_ABL_NOBAR
      __syncthreads();
#endif
    };
    chunk_body(kc0, std::true_type{});
    for (int ch = kc0 + 1; ch < kc1; ++ch) chunk_body(ch, std::false_type{});
#undef SIVAE_AD_STAGE_X
#undef SIVAE_AD_LOAD_A
#undef SIVAE_AD_LOAD_X
#undef SIVAE_AD_READ_B
  } else {
  u32x4_t wr[NWQ];

#define SIVAE_LOAD_CHUNK(CH)                                                                                   \
  {                                                                                                            \
    const __amdgpu_buffer_rsrc_t wr_c =                                                                        \
        make_rsrc(wbase + ((size_t)co_tile * a.nchunks + (CH)) * (size_t)NW * 16, (unsigned long long)NW * 16ull); \
    _Pragma("unroll") for (int q = 0; q < NWQ; ++q) wr[q] = buf_load_u32x4(wr_c, wo[q], 0u);                   \
    const unsigned xsoff = (unsigned)(CH) * (unsigned)(NCB * 16) * (unsigned)HWs;                              \
    _Pragma("unroll") for (int p = 0; p < MAXV; ++p) xr[p] = buf_load_u32x4(xrsrc, xo[p], xsoff);              \
  }
  (void)wrsrc;

  SIVAE_LOAD_CHUNK(kc0)
  if (PRO) __syncthreads();
  for (int ch = kc0; ch < kc1; ++ch) {
#pragma unroll
    for (int q = 0; q < NWQ; ++q) {
      const int idx = tid + q * NT;
      if (NW % NT == 0 || idx < NW) ws[idx] = wr[q];
    }
#pragma unroll
    for (int p = 0; p < MAXV; ++p) {
      const int v = tid + p * NT;
      u32x4_t q = xr[p];
      if (PRO) {
        float f[8];
        unpack8(q, f);
        const float* pp = ps + (ch * NCB + xcb[p]) * 16;
        const bool in = xo[p] != SIVAE_OOB;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = in ? lrelu01(f[e] * pp[e] + pp[8 + e], a.pro_slope) : 0.f;
        q = pack8(f);
      }
      if (v < nvec) xs[v] = q;
    }
    __syncthreads();
    if (ch + 1 < kc1) SIVAE_LOAD_CHUNK(ch + 1)

#pragma unroll
    for (int ks = 0; ks < CKS; ++ks) {
#pragma unroll
      for (int kh = 0; kh < KS; ++kh) {
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
          const int tap = kh * KW + kw;
          bf16x8_t av[WM], bv[WN];
#pragma unroll
          for (int m = 0; m < WM; ++m)
            av[m] = __builtin_bit_cast(bf16x8_t, ws[a_base + (tap * CKS + ks) * 2 * TCO + m * 32]);
          const int soff = ks * 2 * plane + kh * LW + kw;
#pragma unroll
          for (int n = 0; n < WN; ++n) bv[n] = __builtin_bit_cast(bf16x8_t, xs[b_base[n] + soff]);
#pragma unroll
          for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[m], bv[n], acc[m][n], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
#undef SIVAE_LOAD_CHUNK
  }  // (!AD)

  // ---- epilogue
  int px_ok[WN];
  unsigned y_off[WN];
#pragma unroll
  for (int n = 0; n < WN; ++n) {
    const int m_pix = (wvn * WN + n) * 32 + l31;
    const int cc = m_pix & (TW - 1);
    const int rr = (m_pix >> a.tw_log2) & (TH - 1);
    const int tb = m_pix >> (a.tw_log2 + a.th_log2);
    const int r = r0 + rr, c = c0 + cc;
    px_ok[n] = (tb < nb_here && r < H && c < W) ? 1 : 0;
    if (OUTF32 && KS == 3)  // split-K partials: fp32 in the blocked layout, [b][Cob][H][W][8] floats
      y_off[n] = px_ok[n] ? (unsigned)((tb * a.Cob * H + r) * W + c) * 32u + (unsigned)hh * 16u : SIVAE_OOB16;  // (16-byte stores)
    else if (OUTF32)
      y_off[n] = px_ok[n] ? (unsigned)((tb * a.Co * H + r) * W + c) * 4u : SIVAE_OOB;
    else
      y_off[n] = px_ok[n] ? (unsigned)((tb * a.Cob * H + r) * W + c) * 16u + (unsigned)hh * 8u : SIVAE_OOB;
  }
  float* red = reinterpret_cast<float*>(smem_raw);  // [WVN][TCO][2]
  const bool want_stats = a.stats != nullptr;

  if constexpr (OUTF32 && KS == 3) {
    // Split-K partials (the only use of the 3x3 fp32-output instantiations): the accumulator's own grouping — four
    // consecutive channels of a pixel per lane and register group — is one 16-byte store into the fp32 twin of the blocked
    // layout, part[slice][b][Cob][H][W][8] (the NCHW form was 64 range-checked dword stores per lane: 508 vector + 108 branch
    // instructions behind a K loop of 144 MFMAs).  Padded channels have zero weights: they store zeros, nothing to check.
    const __amdgpu_buffer_rsrc_t prsrc =
        make_rsrc(reinterpret_cast<float*>(a.y) + (size_t)slice * a.split_stride + (size_t)b0 * a.Cob * 8 * HW,
                  (unsigned long long)nb_here * a.Cob * HW * 32ull);
#pragma unroll
    for (int m = 0; m < WM; ++m) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cb = (co0 >> 3) + (wvm * WM + m) * 4 + g;  // (block-uniform)
        const unsigned cboff = (unsigned)cb * (unsigned)HW * 32u;
#pragma unroll
        for (int n = 0; n < WN; ++n) {
          const f32x4 fv = {acc[m][n][4 * g], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, fv), prsrc, (int)(cb < a.Cob ? y_off[n] : SIVAE_OOB16), (int)cboff, 0);
        }
      }
    }
    return;
  } else if constexpr (OUTF32) {
    const __amdgpu_buffer_rsrc_t yrsrc =
        make_rsrc(reinterpret_cast<float*>(a.y) + (size_t)slice * a.split_stride + (size_t)b0 * a.Co * HW,
                  (unsigned long long)nb_here * a.Co * HW * 4ull);
#pragma unroll
    for (int m = 0; m < WM; ++m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int chn = co0 + (wvm * WM + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const bool ch_ok = chn < a.Co;
        const float bias = (a.bias != nullptr && ch_ok) ? a.bias[chn] : 0.f;
        const unsigned choff = (unsigned)chn * (unsigned)HW * 4u;
#pragma unroll
        for (int n = 0; n < WN; ++n) {
          const unsigned off = (ch_ok && px_ok[n]) ? y_off[n] + choff : SIVAE_OOB;
          float v = acc[m][n][r] + bias;
          if (a.accumulate) v += buf_load_f32(yrsrc, off, 0u);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrsrc, (int)off, 0, 0);
        }
      }
    }
    return;
  }

  if constexpr (KS == 3 && KW == 3 && !PRO) {
    if (a.poolsum) {
      // 2x2 block sums of the fp32 accumulators, rounded once.  A 32-pixel MFMA tile is one image row when the tile is 32
      // pixels wide (the vertical neighbour is then tile n ^ 1 of the same lane), else 32 / TW rows of TW pixels (the
      // vertical neighbour is lane ^ TW); the horizontal neighbour is lane ^ 1.  Tiles start on even rows / columns and H, W
      // are even: a block never leaves its tile or the image.  The lane of the block's top-left pixel stores.
      const int Hh = H >> 1, Wh = W >> 1, HWh = Hh * Wh;
      const __amdgpu_buffer_rsrc_t prs = make_rsrc(reinterpret_cast<unsigned char*>(a.y) + (size_t)b0 * a.Cob * HWh * 16,
                                                   (unsigned long long)nb_here * a.Cob * HWh * 16ull);
      const bool row_tiles = a.tw_log2 >= 5;
      bool own[WN];
      unsigned p_off[WN];
#pragma unroll
      for (int n = 0; n < WN; ++n) {
        const int m_pix = (wvn * WN + n) * 32 + l31;
        const int cc = m_pix & (TW - 1);
        const int rr = (m_pix >> a.tw_log2) & (TH - 1);
        const int tb = m_pix >> (a.tw_log2 + a.th_log2);
        const int r = r0 + rr, c = c0 + cc;
        own[n] = px_ok[n] && !(r & 1) && !(c & 1);
        p_off[n] = (unsigned)((tb * a.Cob * Hh + (r >> 1)) * Wh + (c >> 1)) * 16u + (unsigned)hh * 8u;
      }
      auto pool_epilogue = [&](auto ACC_) {
        constexpr bool ACC = decltype(ACC_)::value;
#pragma unroll
        for (int m = 0; m < WM; ++m) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int cb = (co0 >> 3) + (wvm * WM + m) * 4 + g;
            const bool cb_ok = cb < a.Cob;
            const unsigned cboff = (unsigned)cb * (unsigned)HWh * 16u;
#pragma unroll
            for (int n = 0; n < WN; ++n) {
              if (row_tiles && (n & 1)) continue;
              float v[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float t = acc[m][n][4 * g + e];
                if (row_tiles)
                  t += acc[m][n ^ 1][4 * g + e];
                else
                  t += __shfl_xor(t, TW, 64);
                t += dpp_mov_f32<0xB1, 0xf>(t);  // lane ^ 1
                v[e] = t;
              }
              const unsigned off = (cb_ok && own[n]) ? p_off[n] + cboff : SIVAE_OOB;
              if constexpr (ACC) {
                const u32x2_t old = __builtin_amdgcn_raw_buffer_load_b64(prs, (int)off, 0, 0);
                v[0] += bf16_lo(old[0]);
                v[1] += bf16_hi(old[0]);
                v[2] += bf16_lo(old[1]);
                v[3] += bf16_hi(old[1]);
              }
              u32x2_t o;
              o[0] = pack_bf16(v[0], v[1]);
              o[1] = pack_bf16(v[2], v[3]);
              __builtin_amdgcn_raw_buffer_store_b64(o, prs, (int)off, 0, 0);
            }
          }
        }
      };
      if (a.accumulate)
        pool_epilogue(std::true_type{});
      else
        pool_epilogue(std::false_type{});
      return;
    }
  }
  const __amdgpu_buffer_rsrc_t yrsrc = make_rsrc(reinterpret_cast<unsigned char*>(a.y) + (size_t)b0 * a.Cob * HW * 16,
                                                 (unsigned long long)nb_here * a.Cob * HW * 16ull);
  // The run-time switches of the epilogue are block-uniform: one copy of the loop per (accumulate, statistics, whole tile)
  // combination instead of 149 branches inside it.  Statistics: the lane's {sum, sumsq} of all its WM * 16 channel slots stay
  // in registers (slot k = (m * 4 + g) * 4 + e) and ONE transposing reduction over the 32 pixel lanes (common.h) leaves
  // lane k with the totals of slot k — 2 * (WM * 32 - WM) adds instead of WM * 32 five-step butterflies (the round-2 form:
  // 640 of the epilogue's 1 740 vector instructions).
  const bool full_tile = nb_here == TB && r0 + TH <= H && c0 + TW <= W;
  auto epilogue = [&](auto ACC_, auto STATS_, auto FULL_) {
    constexpr bool ACC = decltype(ACC_)::value, STATS = decltype(STATS_)::value, FULL = decltype(FULL_)::value;
    float sq[WM * 32];  // [0, WM*16): sums, [WM*16, WM*32): sums of squares (dead code without statistics)
    if constexpr (STATS) {
#pragma unroll
      for (int k = 0; k < WM * 32; ++k) sq[k] = 0.f;
    }
#pragma unroll
    for (int m = 0; m < WM; ++m) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col0 = (wvm * WM + m) * 32 + 8 * g + 4 * hh;  // first of this lane's 4 channels, within the tile
        const int cb = (co0 >> 3) + (wvm * WM + m) * 4 + g;
        const bool cb_ok = cb < a.Cob;
        float bias[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias != nullptr) {
#pragma unroll
          for (int e = 0; e < 4; ++e) bias[e] = (co0 + col0 + e < a.Co) ? a.bias[co0 + col0 + e] : 0.f;
        }
        const unsigned cboff = (unsigned)cb * (unsigned)HW * 16u;
#pragma unroll
        for (int n = 0; n < WN; ++n) {
          const unsigned off = (cb_ok && (FULL || px_ok[n])) ? y_off[n] + cboff : SIVAE_OOB;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[m][n][4 * g + e] + bias[e];
          if constexpr (ACC) {
            const u32x2_t old = __builtin_amdgcn_raw_buffer_load_b64(yrsrc, (int)off, 0, 0);
            v[0] += bf16_lo(old[0]);
            v[1] += bf16_hi(old[0]);
            v[2] += bf16_lo(old[1]);
            v[3] += bf16_hi(old[1]);
          }
          u32x2_t o;
          o[0] = pack_bf16(v[0], v[1]);
          o[1] = pack_bf16(v[2], v[3]);
          __builtin_amdgcn_raw_buffer_store_b64(o, yrsrc, (int)off, 0, 0);
          if constexpr (STATS) {
            float w[4] = {bf16_lo(o[0]), bf16_hi(o[0]), bf16_lo(o[1]), bf16_hi(o[1])};
            if constexpr (!FULL) {
              const float okf = px_ok[n] ? 1.f : 0.f;
#pragma unroll
              for (int e = 0; e < 4; ++e) w[e] *= okf;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              sq[(m * 4 + g) * 4 + e] += w[e];
              sq[WM * 16 + (m * 4 + g) * 4 + e] = fmaf(w[e], w[e], sq[WM * 16 + (m * 4 + g) * 4 + e]);
            }
          }
        }
      }
    }
    if constexpr (STATS) {
      lanes32_transpose_sum<WM * 32>(sq, lane);
      // WM == 2: lane k holds {sum, sumsq} of slot k in sq[0], sq[1];  WM == 1: lanes 0-15 the sum of slot k, lanes 16-31 the
      // sum of squares of slot k - 16, in sq[0]
      const int k = WM == 2 ? l31 : (l31 & 15);
      const int col = (wvm * WM + (k >> 4)) * 32 + 8 * ((k >> 2) & 3) + 4 * hh + (k & 3);
      if (WM == 2) {
        red[(wvn * TCO + col) * 2 + 0] = sq[0];
        red[(wvn * TCO + col) * 2 + 1] = sq[WM == 2 ? 1 : 0];
      } else {
        red[(wvn * TCO + col) * 2 + (l31 >> 4)] = sq[0];
      }
    }
  };
  {
    using T = std::true_type;
    using F = std::false_type;
    if (a.accumulate) {
      if (want_stats) {
        if (full_tile) epilogue(T{}, T{}, T{}); else epilogue(T{}, T{}, F{});
      } else {
        if (full_tile) epilogue(T{}, F{}, T{}); else epilogue(T{}, F{}, F{});
      }
    } else {
      if (want_stats) {
        if (full_tile) epilogue(F{}, T{}, T{}); else epilogue(F{}, T{}, F{});
      } else {
        if (full_tile) epilogue(F{}, F{}, T{}); else epilogue(F{}, F{}, F{});
      }
    }
  }
  if (want_stats) {
    __syncthreads();
    for (int c = tid; c < TCO; c += NT) {
      if (co0 + c < a.Co) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < WVN; ++w) {
          s += red[(w * TCO + c) * 2 + 0];
          q += red[(w * TCO + c) * 2 + 1];
        }
        float* dst = a.stats + ((size_t)pt * a.Co + co0 + c) * 2;
        dst[0] = s;
        dst[1] = q;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight pack: fp32 master weights [Co][Ci][KS][KS] -> bf16 operand slabs
//   wp[co_tile][chunk][tap][ks][hh][TCO][8]   (input channel = (chunk*CKS + ks)*16 + hh*8 + e)
// mode 0: forward operand; mode 1: data-gradient operand (output channels = the conv's input channels, taps flipped)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bf16_pack_element(const float* __restrict__ w, unsigned short* __restrict__ wp, int Co,
                                                  int Ci, int KS, int KW, int mode, int TCO, int CKS, int nchunks, size_t i) {
  // outputs / inputs of the GEMM this pack feeds
  const int N_out = mode == 0 ? Co : Ci, N_in = mode == 0 ? Ci : Co;
  const int TAPS = KS * KW;
  size_t t = i;
  const int e = (int)(t % 8);
  t /= 8;
  const int o = (int)(t % TCO);
  t /= TCO;
  const int hh = (int)(t % 2);
  t /= 2;
  const int ks = (int)(t % CKS);
  t /= CKS;
  const int tap = (int)(t % TAPS);
  t /= TAPS;
  const int chunk = (int)(t % nchunks);
  const int co_tile = (int)(t / nchunks);
  const int oc = co_tile * TCO + o;
  const int ic = (chunk * CKS + ks) * 16 + hh * 8 + e;
  float v = 0.f;
  if (oc < N_out && ic < N_in) {
    const int kh = tap / KW, kw = tap % KW;
    if (mode == 0)
      v = w[(((size_t)oc * Ci + ic) * KS + kh) * KW + kw];
    else
      v = w[(((size_t)ic * Ci + oc) * KS + (KS - 1 - kh)) * KW + (KW - 1 - kw)];
  }
  const unsigned u = pack_bf16(v, 0.f);
  wp[i] = (unsigned short)(u & 0xffffu);
}

__global__ void bf16_pack_conv_weight_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int Co,
                                             int Ci, int KS, int KW, int mode, int TCO, int CKS, int nchunks,
                                             size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  bf16_pack_element(w, wp, Co, Ci, KS, KW, mode, TCO, CKS, nchunks, i);
}

// batched form (pack_batch.h, form SIVAE_PACK_BF16): block -> job through the uint16 map, a job's blocks stride over its
// elements — ONE launch rebuilds the operand slabs of every conv weight of a network after its optimizer step (the bf16
// mode's ~65 per-weight launches of 7 us per step were 1 % of the celeb128 iteration, 2 % of the 16-image shard's)
__global__ void __launch_bounds__(256) bf16_pack_conv_weight_batch_kernel(const SivaePackJob* __restrict__ jobs,
                                                                          const unsigned short* __restrict__ block_job) {
  const SivaePackJob j = jobs[block_job[blockIdx.x]];
  unsigned short* wp = reinterpret_cast<unsigned short*>(j.dst);
  for (size_t i = (size_t)(blockIdx.x - j.blk0) * 256 + threadIdx.x; i < (size_t)j.total; i += (size_t)j.nblk * 256)
    bf16_pack_element(j.w, wp, j.Co, j.Ci, j.kdim, j.ndim, j.mode, j.kpad, j.npad, j.aux, i);
}

namespace {

struct Bf16Cfg {
  int TCO, TPX, CKS;
};
// ks codes: 1, 3, 5 = square kernels; 51 = 5 rows x 1 column (the kw-packed form of the RGB-side 5x5 layers: the 3
// image channels and the 5 kernel columns share the 16-channel k-step, bf16_bn.hip::bf16_im2col_kw5_kernel)
bool ks_ok(int ks) { return ks == 1 || ks == 3 || ks == 5 || ks == 51; }
int ks_h(int ks) { return ks == 51 ? 5 : ks; }
int ks_w(int ks) { return ks == 51 ? 1 : ks; }
// tile configuration by (ks, output channels, input channels) — shared by the pack and the launch
Bf16Cfg bf16_cfg(int ks, int n_out, int n_in) {
  Bf16Cfg c;
  c.TCO = n_out <= 32 ? 32 : (n_out <= 64 ? 64 : 128);
  c.TPX = n_out <= 64 ? 256 : 128;
  c.CKS = (ks == 1 && (bf16_cblocks(n_in) % 8) == 0) ? 4 : 1;
  return c;
}

template <int KS, int WM, int WN, int WVM, int WVN, int CKS, int MAXV, bool PRO, bool OUTF32, int MINW, int KW = KS,
          bool AD = false>
int launch_cfg(Bf16ConvArgs& a, hipStream_t stream) {
  constexpr int TCO = WVM * WM * 32;
  constexpr int TPX = WVN * WN * 32;
  constexpr int NT = WVM * WVN * 64;
  constexpr int P = KS / 2, PW = KW / 2;
  constexpr int NW = KS * KW * CKS * 2 * TCO;
  TileGeom g = make_tile_geom(a.B, a.H, a.W, TPX);
  a.tb_log2 = g.tb_log2;
  a.th_log2 = g.th_log2;
  a.tw_log2 = g.tw_log2;
  a.ntb = g.ntb;
  a.nth = g.nth;
  a.ntw = g.ntw;
  a.n_co_tiles = cdiv(a.Co, TCO);
  a.nchunks = a.Cib / (2 * CKS);
  const int plane = (1 << g.tb_log2) * ((1 << g.th_log2) + 2 * P) * ((1 << g.tw_log2) + 2 * PW);
  if (2 * CKS * plane > MAXV * NT) return SIVAE_ERR_SHAPE;
  if (plane >= 1024) return SIVAE_ERR_SHAPE;  // (fastdiv: divisors < 2^10, dividends < 2^22 — MAXV * NT <= 2048 vectors)
  a.magic_plane = make_magic((unsigned)plane);
  a.magic_lw = make_magic((unsigned)((1 << g.tw_log2) + 2 * PW));
  a.magic_lh = make_magic((unsigned)((1 << g.th_log2) + 2 * P));
  size_t lds = (size_t)(AD ? 2 * (2 * CKS * plane) : NW + 2 * CKS * plane) * 16 + (PRO ? (size_t)a.Cib * 64 : 0);
  const size_t red = (size_t)WVN * TCO * 2 * sizeof(float);
  if (lds < red) lds = red;
  const long long nbase = (long long)a.n_co_tiles * g.ntb * g.nth * g.ntw;
  if (a.chunks_per_split <= 0) a.chunks_per_split = a.nchunks;
  const long long nblk = nbase * cdiv(a.nchunks, a.chunks_per_split);
  if (nblk > 0x7fffffffLL) return SIVAE_ERR_RANGE;
  a.nblk_base = (int)nbase;
  auto kern = bf16_conv_kernel<KS, WM, WN, WVM, WVN, CKS, MAXV, PRO, OUTF32, MINW, KW, AD>;
  static size_t lds_hwm = 0;
  const int rc_lds = sivae_ensure_lds(reinterpret_cast<const void*>(kern), lds, &lds_hwm);
  if (rc_lds != SIVAE_OK) return rc_lds;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(NT), lds, stream, a);
  return sivae_launch_status();
}

// SIVAE_BF16_CONV_TILE=0 forces the small pixel tiles (A/B switch for tools/bench_conv16.py)
int big_tiles_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("SIVAE_BF16_CONV_TILE");
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v;
}

// SIVAE_BF16_CONV_AD=0: the 3x3 kernels stage their weight slabs through LDS again (A/B switch)
int a_direct_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("SIVAE_BF16_CONV_AD");
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v;
}

// Pixel tile of the 3x3 kernels.  Per 16-channel chunk a block stages its whole weight slab (288 bytes per output
// channel) through LDS, so with the small tile (128 co x 128 px: 1152 MFMA cycles per SIMD per chunk) the LDS pipe is as
// busy as the matrix pipe (~535 cycles of ds_write_b128 + ~576 of operand reads).  Doubling the pixels per wave
// (64 co x 128 px, 128 accumulators) halves the weight staging per MFMA -> LDS ~65 % of the MFMA time.  The big tile is
// used where it still gives every CU two blocks; small maps / short grids keep the small one.
int px_tile_3x3(int TCO, int B, int H, int W, int n_out) {
  const int small = TCO == 128 ? 128 : 256;
  if (!big_tiles_enabled() || TCO == 32) return small;
  const int big = 2 * small;
  TileGeom g = make_tile_geom(B, H, W, big);
  const long long nblk = (long long)g.ntb * g.nth * g.ntw * cdiv(n_out, TCO);
  const int plane = (1 << g.tb_log2) * ((1 << g.th_log2) + 2) * ((1 << g.tw_log2) + 2);
  if (nblk < 512 || 2 * plane > 5 * 256) return small;
  return big;
}

template <int KS, int CKS, int MAXV, bool PRO, bool OUTF32, int KW = KS>
int launch_by_co(Bf16ConvArgs& a, int TCO, hipStream_t stream) {
  if (TCO == 32) return launch_cfg<KS, 1, 2, 1, 4, CKS, MAXV, PRO, OUTF32, 2, KW>(a, stream);
  if constexpr (OUTF32 && KS == 3) {  // split-K partials of the 3x3 kernels (small pixel tiles)
    if (a_direct_enabled()) {
      if (TCO == 64) return launch_cfg<KS, 2, 2, 1, 4, CKS, MAXV, PRO, true, 2, KS, true>(a, stream);
      return launch_cfg<KS, 2, 2, 2, 2, CKS, MAXV, PRO, true, 2, KS, true>(a, stream);
    }
    if (TCO == 64) return launch_cfg<KS, 2, 2, 1, 4, CKS, MAXV, PRO, true, 2>(a, stream);
    return launch_cfg<KS, 2, 2, 2, 2, CKS, MAXV, PRO, true, 2>(a, stream);
  }
  if (OUTF32) return SIVAE_ERR_SHAPE;  // fp32 NCHW output: the RGB-side `predict` conv and split-K partials
  if constexpr (!OUTF32) {
    if constexpr (KS == 3) {
      const int tpx = px_tile_3x3(TCO, a.B, a.H, a.W, a.Co);
      if (a_direct_enabled()) {  // weights straight from L2 (see the kernel's AD note)
        if (TCO == 64 && tpx == 512) return launch_cfg<KS, 2, 4, 1, 4, CKS, MAXV, PRO, false, 2, KS, true>(a, stream);
        if (TCO == 128 && tpx == 256) return launch_cfg<KS, 2, 4, 2, 2, CKS, MAXV, PRO, false, 2, KS, true>(a, stream);
        if (TCO == 64) return launch_cfg<KS, 2, 2, 1, 4, CKS, MAXV, PRO, false, 2, KS, true>(a, stream);
        if (TCO == 128) return launch_cfg<KS, 2, 2, 2, 2, CKS, MAXV, PRO, false, 2, KS, true>(a, stream);
      }
      if (TCO == 64 && tpx == 512) return launch_cfg<KS, 2, 4, 1, 4, CKS, MAXV, PRO, false, 2>(a, stream);
      if (TCO == 128 && tpx == 256) return launch_cfg<KS, 2, 4, 2, 2, CKS, MAXV, PRO, false, 2>(a, stream);
    }
    if (TCO == 64) return launch_cfg<KS, 2, 2, 1, 4, CKS, MAXV, PRO, false, 2, KW>(a, stream);
    return launch_cfg<KS, 2, 2, 2, 2, CKS, MAXV, PRO, false, 2, KW>(a, stream);
  }
  return SIVAE_ERR_SHAPE;
}

}  // namespace

extern "C" int sivae_bf16_cblocks(int C) { return C <= 0 ? SIVAE_ERR_SHAPE : bf16_cblocks(C); }

extern "C" size_t sivae_bf16_pack_conv_weight_bytes(int Co, int Ci, int ks, int mode) {
  if (Co <= 0 || Ci <= 0 || !ks_ok(ks) || (mode != 0 && mode != 1)) return 0;
  const int n_out = mode == 0 ? Co : Ci, n_in = mode == 0 ? Ci : Co;
  const Bf16Cfg c = bf16_cfg(ks, n_out, n_in);
  const int nchunks = bf16_cblocks(n_in) / (2 * c.CKS);
  return (size_t)cdiv(n_out, c.TCO) * nchunks * ks_h(ks) * ks_w(ks) * c.CKS * 2 * c.TCO * 16;
}

extern "C" int sivae_bf16_pack_conv_weight(const float* w, void* wp, int Co, int Ci, int ks, int mode,
                                           hipStream_t stream) {
  if (!w || !wp) return SIVAE_ERR_NULL;
  const size_t bytes = sivae_bf16_pack_conv_weight_bytes(Co, Ci, ks, mode);
  if (bytes == 0) return !ks_ok(ks) ? SIVAE_ERR_KSIZE : SIVAE_ERR_SHAPE;
  const int n_out = mode == 0 ? Co : Ci, n_in = mode == 0 ? Ci : Co;
  const Bf16Cfg c = bf16_cfg(ks, n_out, n_in);
  const int nchunks = bf16_cblocks(n_in) / (2 * c.CKS);
  const size_t total = bytes / 2;
  hipLaunchKernelGGL(bf16_pack_conv_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w,
                     reinterpret_cast<unsigned short*>(wp), Co, Ci, ks_h(ks), ks_w(ks), mode, c.TCO, c.CKS, nchunks,
                     total);
  return sivae_launch_status();
}

// job shape of the batched pack (pack_batch.h): kdim / ndim = kernel rows / columns, kpad = TCO, npad = CKS, aux = chunks
int sivae_packjob_bf16(SivaePackJob* j, int Co, int Ci, int ks, int mode) {
  if (!ks_ok(ks)) return SIVAE_ERR_KSIZE;
  const size_t bytes = sivae_bf16_pack_conv_weight_bytes(Co, Ci, ks, mode);
  if (bytes == 0) return SIVAE_ERR_SHAPE;
  const int n_out = mode == 0 ? Co : Ci, n_in = mode == 0 ? Ci : Co;
  const Bf16Cfg c = bf16_cfg(ks, n_out, n_in);
  j->taps = ks_h(ks) * ks_w(ks);
  j->kdim = ks_h(ks);
  j->ndim = ks_w(ks);
  j->kpad = c.TCO;
  j->npad = c.CKS;
  j->aux = bf16_cblocks(n_in) / (2 * c.CKS);
  j->total = bytes / 2;
  return SIVAE_OK;
}
void sivae_packbatch_bf16(const SivaePackJob* jobs, const unsigned short* block_job, int nblocks, hipStream_t s) {
  hipLaunchKernelGGL(bf16_pack_conv_weight_batch_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, jobs, block_job);
}

extern "C" int sivae_bf16_conv2d_num_px_tiles(int B, int Co, int H, int W, int ks) {
  if (B <= 0 || Co <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  if (!ks_ok(ks)) return SIVAE_ERR_KSIZE;
  const Bf16Cfg c = bf16_cfg(ks, Co, 16);
  // (only the 3x3 kernels have the big pixel tiles; the launch below makes the same choice)
  TileGeom g = make_tile_geom(B, H, W, ks == 3 ? px_tile_3x3(c.TCO, B, H, W, Co) : c.TPX);
  return g.ntb * g.nth * g.ntw;
}

extern "C" int sivae_bf16_conv2d_fwd(const void* x, const void* wp, void* y, const float* bias, const float* pro_mean,
                                     const float* pro_invstd, const float* pro_gamma, const float* pro_beta,
                                     float pro_slope, float* stats_partial, int B, int Ci, int Co, int H, int W, int ks,
                                     int upsample, int accumulate, int out_f32_nchw, hipStream_t stream) {
  if (!x || !wp || !y) return SIVAE_ERR_NULL;
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  if (!ks_ok(ks)) return SIVAE_ERR_KSIZE;
  if (upsample && ((H & 1) || (W & 1))) return SIVAE_ERR_SHAPE;
  if (pro_mean && (!pro_invstd || !pro_gamma || !pro_beta)) return SIVAE_ERR_NULL;
  if (pro_mean && ks != 3) return SIVAE_ERR_MODE;  // the producer-BatchNorm prologue exists on the 3x3 kernel (conv2)
  if (out_f32_nchw && (stats_partial || Co > 32)) return SIVAE_ERR_MODE;
  const long long hw = (long long)H * W;
  const int Cib = bf16_cblocks(Ci), Cob = bf16_cblocks(Co);
  if ((long long)Cib * hw * 16 >= 0x7fffffffLL || (long long)Cob * hw * 16 >= 0x3fffffffLL) return SIVAE_ERR_RANGE;
  Bf16ConvArgs a;
  a.x = x;
  a.wp = wp;
  a.y = y;
  a.bias = bias;
  a.pro_mean = pro_mean;
  a.pro_invstd = pro_invstd;
  a.pro_gamma = pro_gamma;
  a.pro_beta = pro_beta;
  a.pro_slope = pro_slope;
  a.stats = stats_partial;
  a.B = B;
  a.Ci = Ci;
  a.Co = Co;
  a.H = H;
  a.W = W;
  a.Cib = Cib;
  a.Cob = Cob;
  a.accumulate = accumulate;
  a.upsample = upsample;
  a.chunks_per_split = 0;
  a.split_stride = 0;
  const Bf16Cfg c = bf16_cfg(ks, Co, Ci);
  if (ks == 3) {
    if (pro_mean) return launch_by_co<3, 1, 5, true, false>(a, c.TCO, stream);
    return launch_by_co<3, 1, 5, false, false>(a, c.TCO, stream);
  }
  if (ks == 1) {
    if (c.CKS == 4) return launch_by_co<1, 4, 8, false, false>(a, c.TCO, stream);
    return launch_by_co<1, 1, 2, false, false>(a, c.TCO, stream);
  }
  if (ks == 51) {
    if (out_f32_nchw) return launch_by_co<5, 1, 4, false, true, 1>(a, c.TCO, stream);
    return launch_by_co<5, 1, 4, false, false, 1>(a, c.TCO, stream);
  }
  if (out_f32_nchw) return launch_by_co<5, 1, 4, false, true>(a, c.TCO, stream);
  return launch_by_co<5, 1, 4, false, false>(a, c.TCO, stream);
}

// y_half[b][co][h/2][w/2] (+)= the 2x2 block sums of conv3x3(x)[b][co][h][w] (H, W even; y_half blocked bf16 at half
// resolution): the data gradient of a 3x3 conv whose input was read through nearest-2x upsample addressing, with the
// adjoint of the nn.Upsample (train_soft_intro_vae.py:155) folded into the epilogue — the four fp32 accumulators are
// summed before the one rounding, and the full-resolution gradient is never written.  wp: the data-gradient pack
// (mode 1; Ci / Co are this launch's inputs / outputs).
extern "C" int sivae_bf16_conv2d_fwd_pool(const void* x, const void* wp, void* y_half, int B, int Ci, int Co, int H, int W,
                                          int accumulate, hipStream_t stream) {
  if (!x || !wp || !y_half) return SIVAE_ERR_NULL;
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return SIVAE_ERR_SHAPE;
  const long long hw = (long long)H * W;
  const int Cib = bf16_cblocks(Ci), Cob = bf16_cblocks(Co);
  if ((long long)Cib * hw * 16 >= 0x7fffffffLL || (long long)Cob * hw * 16 >= 0x3fffffffLL) return SIVAE_ERR_RANGE;
  Bf16ConvArgs a;
  a.x = x;
  a.wp = wp;
  a.y = y_half;
  a.bias = nullptr;
  a.pro_mean = a.pro_invstd = a.pro_gamma = a.pro_beta = nullptr;
  a.pro_slope = 1.f;
  a.stats = nullptr;
  a.B = B;
  a.Ci = Ci;
  a.Co = Co;
  a.H = H;
  a.W = W;
  a.Cib = Cib;
  a.Cob = Cob;
  a.accumulate = accumulate;
  a.upsample = 0;
  a.chunks_per_split = 0;
  a.split_stride = 0;
  a.poolsum = 1;
  const Bf16Cfg c = bf16_cfg(3, Co, Ci);
  return launch_by_co<3, 1, 5, false, false>(a, c.TCO, stream);
}

// ---- split-K form (3x3, small grids) ------------------------------------------------------------------------------
// y[b][cb][p][8] = bf16( sum_s part[s][b][cb][p][8] (+ y_old) );  stats[b][c] = {sum, sumsq} of the rounded values.
// The partials are fp32 in the blocked layout (the conv's 16-byte stores): a thread owns ONE pixel of an (image, channel
// block) plane — two 16-byte loads per slice, independent across the slices —, sums the slices in order, rounds, and the
// plane's lanes fold the 16 statistics with a transposing reduction (common.h).
// ROWS: planes of 16 pixels (4 x 4 maps), one per 16-lane row of a wave.  Otherwise a wave walks one plane, 64 pixels a turn.
template <bool ROWS>
__global__ void __launch_bounds__(256) bf16_splitk_reduce_kernel(const float* __restrict__ part, void* __restrict__ y,
                                                                 float* __restrict__ stats, int S, int Co, int Cob, int HW,
                                                                 int nplanes, size_t slice_stride, int accumulate) {
  const int lane = (int)threadIdx.x & 63;
  const int wave_g = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
  const int plane = ROWS ? wave_g * 4 + (lane >> 4) : wave_g;  // plane = b * Cob + cb
  const bool live = plane < nplanes;
  const int b = live ? plane / Cob : 0, cb = live ? plane - b * Cob : 0;
  float sq[16];  // [0, 8): sums, [8, 16): sums of squares of this lane's pixels
#pragma unroll
  for (int k = 0; k < 16; ++k) sq[k] = 0.f;
  const float4* pp = reinterpret_cast<const float4*>(part) + (size_t)plane * HW * 2;
  u32x4_t* yv = reinterpret_cast<u32x4_t*>(y) + (size_t)plane * HW;
  for (int p = ROWS ? (lane & 15) : lane; p < HW; p += 64) {  // (ROWS: HW == 16, one turn)
    float f[8];
    if (live && accumulate) {
      unpack8(yv[p], f);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = 0.f;
    }
    if (live) {
#pragma unroll 4
      for (int k = 0; k < S; ++k) {
        const float4* q = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(pp + 2 * p) + (size_t)k * slice_stride);
        const float4 lo = q[0], hi = q[1];
        f[0] += lo.x; f[1] += lo.y; f[2] += lo.z; f[3] += lo.w;
        f[4] += hi.x; f[5] += hi.y; f[6] += hi.z; f[7] += hi.w;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (cb * 8 + e >= Co) f[e] = 0.f;
      const u32x4_t o = pack8(f);
      yv[p] = o;
      float r[8];
      unpack8(o, r);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sq[e] += r[e];
        sq[8 + e] = fmaf(r[e], r[e], sq[8 + e]);
      }
    }
  }
  if (stats == nullptr) return;
  // (every lane of the wave is here: the reductions read their neighbours' registers)
  int k;
  float tot;
  if (ROWS) {
    row16_transpose_sum16(sq, lane);  // lane l: the row's total of value l & 15
    k = lane & 15;
    tot = sq[0];
  } else {
    float s8[8], q8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s8[e] = sq[e];
      q8[e] = sq[8 + e];
    }
    wave_transpose_sum8(s8, lane);  // lane l: the wave's total of value l & 7
    wave_transpose_sum8(q8, lane);
    k = lane & 15;
    tot = (lane & 8) ? q8[0] : s8[0];
    if (lane >= 16) return;
  }
  const int c = cb * 8 + (k & 7);
  if (live && c < Co) stats[((size_t)b * Co + c) * 2 + (k >> 3)] = tot;
}

static void launch_splitk_reduce(const float* part, void* y, float* stats, int S, int B, int Co, int Cob, int HW,
                                 size_t slice_stride, int accumulate, hipStream_t stream) {
  const int nplanes = B * Cob;
  if (HW == 16)
    hipLaunchKernelGGL(bf16_splitk_reduce_kernel<true>, dim3((unsigned)((nplanes + 15) / 16)), dim3(256), 0, stream, part, y,
                       stats, S, Co, Cob, HW, nplanes, slice_stride, accumulate);
  else
    hipLaunchKernelGGL(bf16_splitk_reduce_kernel<false>, dim3((unsigned)((nplanes + 3) / 4)), dim3(256), 0, stream, part, y,
                       stats, S, Co, Cob, HW, nplanes, slice_stride, accumulate);
}

// number of K slices sivae_bf16_conv2d_fwd_splitk uses (1: it is the plain kernel)
extern "C" int sivae_bf16_conv2d_splitk(int B, int Ci, int Co, int H, int W, int ks) {
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("SIVAE_BF16_SPLITK");
    enabled = (e && e[0] == '0') ? 0 : 1;
  }
  if (!enabled || ks != 3 || Co <= 32) return 1;
  const Bf16Cfg c = bf16_cfg(3, Co, Ci);
  const int small = c.TCO == 128 ? 128 : 256;
  if (px_tile_3x3(c.TCO, B, H, W, Co) != small) return 1;  // the grid already fills the chip
  TileGeom g = make_tile_geom(B, H, W, small);
  const long long blocks = (long long)g.ntb * g.nth * g.ntw * cdiv(Co, c.TCO);
  const int nchunks = bf16_cblocks(Ci) / 2;
  if (blocks > 256 || nchunks < 8) return 1;
  int S = (int)(512 / blocks);
  if (S > nchunks / 4) S = nchunks / 4;
  if (S > 8) S = 8;
  return S < 2 ? 1 : S;
}

extern "C" size_t sivae_bf16_conv2d_splitk_workspace_bytes(int B, int Ci, int Co, int H, int W, int ks) {
  const int S = sivae_bf16_conv2d_splitk(B, Ci, Co, H, W, ks);
  return S <= 1 ? 0 : (size_t)S * B * bf16_cblocks(Co) * 8 * H * W * sizeof(float);  // (fp32, blocked: padded channels)
}

extern "C" int sivae_bf16_conv2d_splitk_stats_rows(int B, int Ci, int Co, int H, int W, int ks) {
  const int S = sivae_bf16_conv2d_splitk(B, Ci, Co, H, W, ks);
  if (S < 0) return S;
  return S > 1 ? B : sivae_bf16_conv2d_num_px_tiles(B, Co, H, W, ks);
}

extern "C" int sivae_bf16_conv2d_fwd_splitk(const void* x, const void* wp, void* y, const float* pro_mean,
                                            const float* pro_invstd, const float* pro_gamma, const float* pro_beta,
                                            float pro_slope, float* stats_partial, int B, int Ci, int Co, int H, int W,
                                            int ks, int upsample, int accumulate, void* workspace,
                                            size_t workspace_bytes, hipStream_t stream) {
  const int S = sivae_bf16_conv2d_splitk(B, Ci, Co, H, W, ks);
  if (S < 0) return S;
  if (S == 1)
    return sivae_bf16_conv2d_fwd(x, wp, y, nullptr, pro_mean, pro_invstd, pro_gamma, pro_beta, pro_slope, stats_partial,
                                 B, Ci, Co, H, W, ks, upsample, accumulate, 0, stream);
  if (!x || !wp || !y || !workspace) return SIVAE_ERR_NULL;
  if (workspace_bytes < sivae_bf16_conv2d_splitk_workspace_bytes(B, Ci, Co, H, W, ks)) return SIVAE_ERR_WORKSPACE;
  if (upsample && ((H & 1) || (W & 1))) return SIVAE_ERR_SHAPE;
  if (pro_mean && (!pro_invstd || !pro_gamma || !pro_beta)) return SIVAE_ERR_NULL;
  Bf16ConvArgs a;
  a.x = x;
  a.wp = wp;
  a.y = workspace;
  a.bias = nullptr;
  a.pro_mean = pro_mean;
  a.pro_invstd = pro_invstd;
  a.pro_gamma = pro_gamma;
  a.pro_beta = pro_beta;
  a.pro_slope = pro_slope;
  a.stats = nullptr;
  a.B = B;
  a.Ci = Ci;
  a.Co = Co;
  a.H = H;
  a.W = W;
  a.Cib = bf16_cblocks(Ci);
  a.Cob = bf16_cblocks(Co);
  a.accumulate = 0;
  a.upsample = upsample;
  const int nchunks = a.Cib / 2;
  a.chunks_per_split = cdiv(nchunks, S);
  a.split_stride = (long long)B * a.Cob * 8 * H * W;
  const Bf16Cfg c = bf16_cfg(3, Co, Ci);
  const int rc = pro_mean ? launch_by_co<3, 1, 5, true, true>(a, c.TCO, stream)
                          : launch_by_co<3, 1, 5, false, true>(a, c.TCO, stream);
  if (rc != SIVAE_OK) return rc;
  launch_splitk_reduce(reinterpret_cast<const float*>(workspace), y, stats_partial, cdiv(nchunks, a.chunks_per_split), B, Co,
                       a.Cob, H * W, (size_t)a.split_stride, accumulate, stream);
  return sivae_launch_status();
}
