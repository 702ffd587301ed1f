// Implicit-GEMM stride-1 "same" convolution for gfx950 on the exact-fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32).  One kernel serves the forward pass and — fed the flipped/transposed
// weight pack — the data gradient.
//
//   D[co][pixel] = sum_{tap, ci} Wp[tap][ci][co] * X[ci][pixel + tap]
//
// MFMA roles: A = weights (row i = output channel), B = activations (col j = pixel), so every
// accumulator register of a lane is one channel of 32 consecutive pixels -> coalesced NCHW stores
// and per-channel BatchNorm partial sums by a half-wave shuffle reduction.
//
// LDS per block:  ws[TAPS][CK][TCO]   weight slab of the current input-channel chunk (co contiguous)
//                 xs[CK][TB][TH+2P][TW+2P]  zero-padded input halo tile, staged ONCE per chunk and
//                 re-read by all KS*KS taps (the im2col matrix is never materialised).
// Both MFMA operands are ds_read_b32 of 32 consecutive floats per half-wave -> conflict free.
// HBM/L2 -> LDS goes through registers with raw buffer loads (per-lane 32-bit byte offset + uniform
// SGPR channel offset; out-of-image lanes carry an out-of-range offset and read 0, so the halo /
// padding costs no branches).  The next chunk's loads are issued before the MFMA loop of the current
// chunk and written to LDS after it (latency hidden under ~9k cycles of MFMA); two blocks per CU
// cover each other's barriers.
//
// Optional fusions (all runtime flags):
//   prologue : x' = LeakyReLU((x - mean[ci]) * invstd[ci]*gamma[ci] + beta[ci])  (producer BatchNorm) on load
//   upsample : x is stored at (H/2, W/2); read x[h>>1][w>>1] (nearest 2x) — the 4x tensor never exists
//   epilogue : + bias[co];  y += result (accumulate);  per-channel sum / sumsq partials for the
//              consumer BatchNorm (deterministic: one partial per pixel tile, reduced later in fp64)
//
// Reference op being replaced: nn.Conv2d(k in {1,3,5}, stride 1, padding k//2) as used at
// soft_intro_vae/train_soft_intro_vae.py:51-61,89,159 (and the F.linear calls at :109,:146 via KS=1).
#include "common.h"
#include <stdlib.h>

struct ConvFwdArgs {
  const float* x;
  const float* wp;  // packed [TAPS][Ci_pad][Co_pad]
  float* y;
  const float* bias;      // [Co] or null
  const float* pro_mean;  // [Ci] or null: fused producer BatchNorm-apply + LeakyReLU
  const float* pro_invstd;
  const float* pro_gamma;
  const float* pro_beta;
  float pro_slope;
  float* stats;  // [n_px_tiles][Co][2] or null
  int B, Ci, Co, H, W;
  int Ci_pad, Co_pad;
  int tb_log2, th_log2, tw_log2;
  int ntb, nth, ntw;
  int n_co_tiles;
  int accumulate;
  int upsample;
};

template <int KS, int WM, int WN, int WVM, int WVN, int CK, int MAXPOS, bool PRO, int MINW = 2>
__global__ void __launch_bounds__(WVM* WVN * 64, MINW) conv_fwd_kernel(ConvFwdArgs a) {
  constexpr int P = KS / 2;
  constexpr int NT = WVM * WVN * 64;
  constexpr int TCO = WVM * WM * 32;
  constexpr int TAPS = KS * KS;
  constexpr int NW4 = TAPS * CK * TCO / 4;  // float4 weight loads per chunk (whole block)
  constexpr int NWQ = (NW4 + NT - 1) / NT;
  constexpr int TCO4 = TCO / 4;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ws = smem;
  float* xs = smem + TAPS * CK * TCO;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int wvm = wave / WVN, wvn = wave % WVN;

  const int TW = 1 << a.tw_log2, TH = 1 << a.th_log2, TB = 1 << a.tb_log2;
  const int LW = TW + 2 * P, LH = TH + 2 * P;
  const int plane = TB * LH * LW;
  const int H = a.H, W = a.W, HW = H * W;
  const int Hs = a.upsample ? (H >> 1) : H, Ws = a.upsample ? (W >> 1) : W;
  const int HWs = Hs * Ws;

  // ---- block -> (co tile, pixel tile).  blockIdx % 8 is the XCD: with 1/2/4/8 co tiles every XCD
  // keeps re-using one weight slab in its private L2.
  const int bid = blockIdx.x;
  const int co_tile = bid % a.n_co_tiles;
  const int pt = bid / a.n_co_tiles;
  const int tw_i = pt % a.ntw;
  const int t2 = pt / a.ntw;
  const int th_i = t2 % a.nth;
  const int tb_i = t2 / a.nth;
  const int b0 = tb_i << a.tb_log2, r0 = th_i << a.th_log2, c0 = tw_i << a.tw_log2;
  const int co0 = co_tile * TCO;

  // ---- descriptors: x slab of this tile's TB images, and the packed weights
  int nb_here = a.B - b0;
  if (nb_here > TB) nb_here = TB;
  const unsigned long long img_bytes = (unsigned long long)a.Ci * HWs * 4ull;
  const __amdgpu_buffer_rsrc_t xrsrc = make_rsrc(a.x + (size_t)b0 * a.Ci * HWs, img_bytes * nb_here);
  const __amdgpu_buffer_rsrc_t wrsrc =
      make_rsrc(a.wp, (unsigned long long)TAPS * a.Ci_pad * a.Co_pad * 4ull);

  // ---- staging map for the input halo tile: this thread owns positions tid + p*NT of the plane
  // for all CK channels of a chunk (the decomposition is chunk-invariant, done once).
  unsigned xo[MAXPOS];  // byte offset of (tb, ci=0, r, c) inside the slab, SIVAE_OOB if padding
#pragma unroll
  for (int p = 0; p < MAXPOS; ++p) {
    const int pos = tid + p * NT;
    unsigned off = SIVAE_OOB;
    if (pos < plane) {
      const int cc = pos % LW;
      const int t = pos / LW;
      const int rr = t % LH;
      const int tb = t / LH;
      const int r = r0 + rr - P, c = c0 + cc - P;
      if (tb < nb_here && r >= 0 && r < H && c >= 0 && c < W) {
        const int rs = a.upsample ? (r >> 1) : r, cs = a.upsample ? (c >> 1) : c;
        off = (((unsigned)tb * a.Ci * Hs + rs) * Ws + cs) * 4u;
      }
    }
    xo[p] = off;
  }
  unsigned wo[NWQ];  // byte offset of this thread's float4 inside chunk 0 of the weight pack
#pragma unroll
  for (int q = 0; q < NWQ; ++q) {
    const int idx = tid + q * NT;
    const int row = idx / TCO4, c4 = idx % TCO4;
    const int tap = row / CK, ck = row % CK;
    wo[q] = (NW4 % NT == 0 || idx < NW4)
                ? (unsigned)(((tap * a.Ci_pad + ck) * a.Co_pad + co0 + c4 * 4) * 4)
                : SIVAE_OOB;
  }

  // ---- per-lane MFMA operand bases
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
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int n = 0; n < WN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  float4 wr[NWQ];
  float xr[MAXPOS][CK];

  // raw loads only (weights + activations) — no consumer of the loaded values in here, so all of
  // them stay in flight across the MFMA loop; the BatchNorm/LeakyReLU prologue is applied when the
  // registers are written to LDS one iteration later.
#define SIVAE_LOAD_CHUNK(CI0)                                                                       \
  {                                                                                                 \
    const unsigned wsoff = (unsigned)(CI0) * (unsigned)a.Co_pad * 4u;                               \
    _Pragma("unroll") for (int q = 0; q < NWQ; ++q) wr[q] = buf_load_f32x4(wrsrc, wo[q], wsoff);    \
    _Pragma("unroll") for (int ck = 0; ck < CK; ++ck) {                                             \
      const int ci = (CI0) + ck;                                                                    \
      const int cic = ci < a.Ci ? ci : a.Ci - 1;                                                    \
      const unsigned xsoff = (unsigned)cic * (unsigned)HWs * 4u;                                    \
      _Pragma("unroll") for (int p = 0; p < MAXPOS; ++p) xr[p][ck] = buf_load_f32(xrsrc, xo[p], xsoff); \
    }                                                                                               \
  }

  const int nchunks = a.Ci_pad / CK;
  SIVAE_LOAD_CHUNK(0)
  for (int ch = 0; ch < nchunks; ++ch) {
    // registers -> LDS
#pragma unroll
    for (int q = 0; q < NWQ; ++q) {
      const int idx = tid + q * NT;
      if (NW4 % NT == 0 || idx < NW4) {
        const int row = idx / TCO4, c4 = idx % TCO4;
        *reinterpret_cast<float4*>(&ws[row * TCO + c4 * 4]) = wr[q];
      }
    }
#pragma unroll
    for (int ck = 0; ck < CK; ++ck) {
      const int ci = ch * CK + ck;
      const bool ci_ok = ci < a.Ci;
      float pm = 0.f, pg = 1.f, pb = 0.f;
      if (PRO) {
        const int cic = ci_ok ? ci : a.Ci - 1;
        pm = a.pro_mean[cic];
        pg = a.pro_invstd[cic] * a.pro_gamma[cic];
        pb = a.pro_beta[cic];
      }
#pragma unroll
      for (int p = 0; p < MAXPOS; ++p) {
        const int pos = tid + p * NT;
        float v = xr[p][ck];
        if (PRO) v = (xo[p] != SIVAE_OOB) ? lrelu((v - pm) * pg + pb, a.pro_slope) : 0.f;
        v = ci_ok ? v : 0.f;
        if (pos < plane) xs[ck * plane + pos] = v;
      }
    }
    __syncthreads();
    if (ch + 1 < nchunks) SIVAE_LOAD_CHUNK((ch + 1) * CK)

    // MFMA loop over the (tap, channel-pair) k-steps of this chunk, fully unrolled; hipcc pipelines the
    // ds_reads against the MFMAs with counted lgkmcnt waits.  (Hand-pipelining the operand reads one k-step
    // ahead with sched_barrier pinning measured 0-3 % SLOWER at two waves per SIMD: the other wave already
    // covers the LDS latency.  A static-priority stagger of co-resident waves also measured null.)
#pragma unroll
    for (int kh = 0; kh < KS; ++kh) {
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) {
        const int tap = kh * KS + kw;
#pragma unroll
        for (int kk = 0; kk < CK / 2; ++kk) {
          float av[WM], bv[WN];
#pragma unroll
          for (int m = 0; m < WM; ++m) av[m] = ws[a_base + (tap * CK + 2 * kk) * TCO + m * 32];
          const int soff = 2 * kk * plane + kh * LW + kw;
#pragma unroll
          for (int n = 0; n < WN; ++n) bv[n] = xs[b_base[n] + soff];
#pragma unroll
          for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m][n], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
#undef SIVAE_LOAD_CHUNK

  // ---- epilogue: bias / accumulate / store, optional BatchNorm partial statistics.  Stores go through a buffer
  // descriptor of this tile's images with 32-bit offsets; a pixel or channel outside the tensor gets an out-of-range
  // offset and its store is dropped — no exec-mask branches, no 64-bit address arithmetic per element.
  const __amdgpu_buffer_rsrc_t yrsrc =
      make_rsrc(a.y + (size_t)b0 * a.Co * HW, (unsigned long long)nb_here * a.Co * HW * 4ull);
  unsigned y_off[WN];
#pragma unroll
  for (int n = 0; n < WN; ++n) {
    const int m_pix = (wvn * WN + n) * 32 + l31;
    const int cc = m_pix & (TW - 1);
    const int rr = (m_pix >> a.tw_log2) & (TH - 1);
    const int tb = m_pix >> (a.tw_log2 + a.th_log2);
    const int r = r0 + rr, c = c0 + cc;
    y_off[n] = (tb < nb_here && r < H && c < W) ? (unsigned)((tb * a.Co * H + r) * W + c) * 4u : SIVAE_OOB;
  }
  float* red = smem;  // reuse LDS: [WVN][TCO][2]
  const bool want_stats = a.stats != nullptr;
#pragma unroll
  for (int m = 0; m < WM; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int col = (wvm * WM + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;  // channel in tile
      const int chn = co0 + col;
      const bool ch_ok = chn < a.Co;
      const float bias = (a.bias != nullptr && ch_ok) ? a.bias[chn] : 0.f;
      const unsigned choff = (unsigned)chn * (unsigned)HW * 4u;
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int n = 0; n < WN; ++n) {
        float v = acc[m][n][r] + bias;
        const bool ok = ch_ok && y_off[n] != SIVAE_OOB;
        const unsigned off = ok ? y_off[n] + choff : SIVAE_OOB;
        if (a.accumulate) v += buf_load_f32(yrsrc, off, 0u);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrsrc, (int)off, 0, 0);
        s += ok ? v : 0.f;
        q += ok ? v * v : 0.f;
      }
      if (want_stats) {
        s = half_wave_sum_hi(s);
        q = half_wave_sum_hi(q);
        if (l31 == 31) {
          red[(wvn * TCO + col) * 2 + 0] = s;
          red[(wvn * TCO + col) * 2 + 1] = q;
        }
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
// host side
// ------------------------------------------------------------------------------------------------
namespace {

template <int KS, int WM, int WN, int WVM, int WVN, int CK, int MAXPOS, int MINW = 2>
int launch_cfg(ConvFwdArgs& a, hipStream_t stream) {
  constexpr int TCO = WVM * WM * 32;
  constexpr int TPX = WVN * WN * 32;
  constexpr int NT = WVM * WVN * 64;
  constexpr int P = KS / 2;
  TileGeom g = make_tile_geom(a.B, a.H, a.W, TPX);
  a.tb_log2 = g.tb_log2;
  a.th_log2 = g.th_log2;
  a.tw_log2 = g.tw_log2;
  a.ntb = g.ntb;
  a.nth = g.nth;
  a.ntw = g.ntw;
  a.n_co_tiles = cdiv(a.Co, TCO);
  const int plane = (1 << g.tb_log2) * ((1 << g.th_log2) + 2 * P) * ((1 << g.tw_log2) + 2 * P);
  if (plane > MAXPOS * NT) return SIVAE_ERR_SHAPE;
  size_t lds = (size_t)(KS * KS * CK * TCO + CK * plane) * sizeof(float);
  const size_t red = (size_t)WVN * TCO * 2 * sizeof(float);
  if (lds < red) lds = red;
  const long long nblk = (long long)a.n_co_tiles * g.ntb * g.nth * g.ntw;
  if (nblk > 0x7fffffffLL) return SIVAE_ERR_RANGE;
  auto kern = a.pro_mean ? conv_fwd_kernel<KS, WM, WN, WVM, WVN, CK, MAXPOS, true, MINW>
                         : conv_fwd_kernel<KS, WM, WN, WVM, WVN, CK, MAXPOS, false, MINW>;
  {
    static size_t lds_hwm[2] = {0, 0};  // per template instantiation, per prologue variant
    const int rc_lds = sivae_ensure_lds(reinterpret_cast<const void*>(kern), lds, &lds_hwm[a.pro_mean ? 1 : 0]);
    if (rc_lds != SIVAE_OK) return rc_lds;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(NT), lds, stream, a);
  return sivae_launch_status();
}

}  // namespace

// Packed-weight padding rules shared with pack.hip: Ci_pad = roundup(Ci, CK(ks)), Co_pad = roundup(Co, 128)
extern "C" int sivae_conv_ck(int ks) { return ks == 1 ? 32 : (ks == 3 ? 8 : (ks == 5 ? 4 : -1)); }
extern "C" int sivae_conv_ci_pad(int ks, int ci) {
  const int ck = sivae_conv_ck(ks);
  return ck < 0 ? SIVAE_ERR_KSIZE : ((ci + ck - 1) / ck) * ck;
}
extern "C" int sivae_conv_co_pad(int co) { return ((co + 127) / 128) * 128; }

// Number of pixel tiles (= rows of the `stats` partial buffer) the forward kernel will use.
// experiment switch (tools/bench_conv.py): SIVAE_FWD3_VARIANT selects the tile configuration of the
// 3x3 kernels; the default (unset / 0) is the production choice.
static int fwd3_variant() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("SIVAE_FWD3_VARIANT");
    v = e ? atoi(e) : 0;
  }
  return v;
}
static int fwd_tpx(int ks, int Co) {
  if (ks == 3 && Co > 64) {
    const int v = fwd3_variant();
    if (v == 2 || v == 3) return 256;
  }
  if (ks == 3 && Co <= 64 && Co > 32 && fwd3_variant() == 4) return 512;
  return (Co <= 64) ? 256 : 128;
}

extern "C" int sivae_conv2d_fwd_num_px_tiles(int B, int Co, int H, int W) {
  if (B <= 0 || Co <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  const int tpx = fwd_tpx(3, Co);
  TileGeom g = make_tile_geom(B, H, W, tpx);
  return g.ntb * g.nth * g.ntw;
}

extern "C" int sivae_conv2d_fwd(const float* x, const float* wp, float* y, const float* bias,
                                const float* pro_mean, const float* pro_invstd, const float* pro_gamma,
                                const float* pro_beta, float pro_slope, float* stats_partial, int B, int Ci,
                                int Co, int H, int W, int ks, int upsample, int accumulate,
                                hipStream_t stream) {
  if (!x || !wp || !y) return SIVAE_ERR_NULL;
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  if (ks != 1 && ks != 3 && ks != 5) return SIVAE_ERR_KSIZE;
  if (upsample && ((H & 1) || (W & 1))) return SIVAE_ERR_SHAPE;
  if (pro_mean && (!pro_invstd || !pro_gamma || !pro_beta)) return SIVAE_ERR_NULL;
  const long long hw = (long long)H * W;
  if ((long long)B * Co * hw >= 0xffffffffLL) return SIVAE_ERR_RANGE;
  // one image of x (Ci*H*W floats) must be addressable with 32-bit byte offsets (tiles spanning
  // several images only exist for images smaller than a tile)
  if ((long long)Ci * hw * 4 >= 0x7fffffffLL) return SIVAE_ERR_RANGE;
  if ((long long)Co * hw * 4 >= 0x3fffffffLL) return SIVAE_ERR_RANGE;  // (32-bit store offsets inside a tile's images)
  ConvFwdArgs a;
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
  a.Ci_pad = sivae_conv_ci_pad(ks, Ci);
  a.Co_pad = sivae_conv_co_pad(Co);
  a.accumulate = accumulate;
  a.upsample = upsample;
  // tile config by output-channel count:  Co<=32 -> 32x256, Co<=64 -> 64x256, else 128x128
  if (ks == 3) {
    if (Co <= 32) return launch_cfg<3, 1, 2, 1, 4, 8, 3>(a, stream);
    if (Co <= 64) {
      if (fwd3_variant() == 4) return launch_cfg<3, 2, 4, 1, 4, 8, 5, 1>(a, stream);  // 64co x 512px
      if (fwd3_variant() == 8) return launch_cfg<3, 2, 2, 1, 4, 4, 3, 3>(a, stream);  // CK=4, three blocks per CU
      if (fwd3_variant() == 9 || fwd3_variant() == 12 || fwd3_variant() == 14)
        return launch_cfg<3, 2, 2, 1, 4, 4, 3, 4>(a, stream);  // CK=4, four blocks per CU
      if (fwd3_variant() == 7) return launch_cfg<3, 2, 2, 1, 4, 8, 3>(a, stream);     // CK=8, two blocks per CU (first version)
      return launch_cfg<3, 2, 2, 1, 4, 2, 3, 4>(a, stream);  // production: CK=2, four blocks per CU
    }
    switch (fwd3_variant()) {
      case 1: return launch_cfg<3, 2, 2, 2, 2, 16, 2, 1>(a, stream);  // CK=16, one block per CU
      case 2: return launch_cfg<3, 2, 4, 2, 2, 8, 3, 1>(a, stream);   // 128co x 256px, 4 waves (64x128 per wave)
      case 3: return launch_cfg<3, 2, 2, 2, 4, 8, 2>(a, stream);      // 128co x 256px, 8 waves
      case 6:
      case 12: return launch_cfg<3, 2, 2, 2, 2, 4, 2, 4>(a, stream);  // CK=4, four blocks per CU
      case 5: return launch_cfg<3, 2, 2, 2, 2, 4, 2, 3>(a, stream);   // CK=4, three blocks per CU
      case 7: return launch_cfg<3, 2, 2, 2, 2, 8, 2>(a, stream);      // CK=8, two blocks per CU (round-1 first version)
      default: return launch_cfg<3, 2, 2, 2, 2, 2, 2, 4>(a, stream);  // production: CK=2, four blocks per CU
                                                                       // (occupancy ladder at bs128: CK8/2 blocks 107.4,
                                                                       //  CK4/3 blocks 110.0, CK2/4 blocks 122.6 img/s)
    }
  } else if (ks == 1) {
    // 16-channel chunks at four blocks per CU: the 1x1 convs are short-K, HBM-leaning kernels (K = 64..512) that
    // need occupancy more than chunk depth (64->128 @128x128 bs128: 0.99 ms with CK=32 / two blocks, 0.63 ms here)
    if (Co <= 32) return launch_cfg<1, 1, 2, 1, 4, 32, 1>(a, stream);
    if (Co <= 64) return launch_cfg<1, 2, 2, 1, 4, 16, 1, 4>(a, stream);
    return launch_cfg<1, 2, 2, 2, 2, 16, 1, 4>(a, stream);
  } else {
    if (Co <= 32) return launch_cfg<5, 1, 2, 1, 4, 4, 4>(a, stream);
    // three blocks per CU: with Ci = 3 (stem forward, `predict` data gradient) the whole K loop is ONE chunk, so
    // nothing inside a block overlaps its load latency — co-resident blocks have to (2.03 -> 1.64 ms at bs128)
    if (Co <= 64) return launch_cfg<5, 2, 2, 1, 4, 4, 4, 3>(a, stream);
    return launch_cfg<5, 2, 2, 2, 2, 4, 3>(a, stream);
  }
}
