// bf16 weight gradient of the stride-1 "same" convolutions (aten::convolution_backward, weight half, of the nn.Conv2d
// layers at soft_intro_vae/train_soft_intro_vae.py:51-61,89,159) for the bf16 mode: operands in the blocked bf16 layout
// of bf16_common.h, v_mfma_f32_32x32x16_bf16 with fp32 accumulation, fp32 gradient out ([Co][Ci][KS][KS], the master
// weights' layout).
//
//   dW[co][ci][tap] = sum_{b, pixel} dY[pixel][co] * X[pixel + tap][ci]           (K = pixels)
//
// The contraction index is the PIXEL, but the blocked layout keeps 8 CHANNELS contiguous, so both MFMA operands (8
// consecutive K values per lane) need a transpose.  It is done by the LDS transpose read of gfx950,
// ds_read_b64_tr_b16: the staged tiles are [32-channel group][position][32 channels] (64 bytes per position); each
// 16-lane group reads a 4-position x 16-channel block and every lane receives 4 positions of ONE channel; two such reads
// give the 8 K values of a lane.  Four consecutive positions x 64 bytes = one 256-byte bank row -> conflict free, and a
// filter tap is an immediate byte offset ((kh*LW + kw) * 64) on the same base address.
// A = dY (rows = output channels), B = X shifted by the tap (cols = input channels): accumulator (tap, r) of a lane is
// dW[co = (r&3) + 8*(r>>2) + 4*(lane>>5)][ci = lane&31][tap].
//
// Work split: block = (co tile, ci tile, pixel slice); a block walks the pixel tiles of its slice (64 pixels per
// stage, next stage's global loads in flight during the MFMA phase), writes its partial dW to the workspace
// [slice][tap][Co][Ci], and a fixed-order reduce kernel sums the slices into [Co][Ci][tap] (deterministic, no atomics).
// Fusions: producer BatchNorm + LeakyReLU re-applied to x on load (conv2's weight gradient reads conv1's raw output),
// nearest-2x upsample addressing of x.
#include "bf16_common.h"
#include <stdlib.h>

struct Bf16WgArgs {
  const void* x;   // bf16 blocked [B][Cib][Hs][Ws][8]
  const void* dy;  // bf16 blocked [B][Cob][H][W][8]
  float* part;     // [nslices][TAPS][Co][Ci]
  const float* pro_mean;
  const float* pro_invstd;
  const float* pro_gamma;
  const float* pro_beta;
  float pro_slope;
  int B, Ci, Co, H, W;
  int Cib, Cob;
  int tb_log2, th_log2, tw_log2;
  int ntb, nth, ntw;
  int n_co_tiles, n_ci_tiles, nslices, ntiles;
  int upsample;
};

// WCO x WCI 32x32 MFMA tiles per wave, WVCO x WVCI waves over (co, ci), WVT waves splitting the taps
// KS x KW taps (KW = KS except for ks code 51 = 5 rows x 1 column, see bf16_conv.hip)
template <int KS, int WCO, int WCI, int WVCO, int WVCI, int WVT, int MAXX, bool PRO, int MINW, int KW = KS>
__global__ void __launch_bounds__(WVCO* WVCI* WVT * 64, MINW) bf16_wgrad_kernel(Bf16WgArgs a) {
  constexpr int P = KS / 2, PW = KW / 2;
  constexpr int NT = WVCO * WVCI * WVT * 64;
  constexpr int TAPS = KS * KW;
  constexpr int TPW = (TAPS + WVT - 1) / WVT;  // taps per wave
  constexpr int TCO = WVCO * WCO * 32, TCI = WVCI * WCI * 32;
  constexpr int NSUB_CO = TCO / 32, NSUB_CI = TCI / 32;
  constexpr int TPX = 64;                      // pixels per stage
  constexpr int NKS = TPX / 16;
  constexpr int NDY = NSUB_CO * TPX * 4;       // 16-byte vectors of the dY tile
  constexpr int MAXD = (NDY + NT - 1) / NT;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4_t* dys = reinterpret_cast<u32x4_t*>(smem_raw);  // [NSUB_CO][TPX][4]
  u32x4_t* xs = dys + NDY;                              // [NSUB_CI][plane][4]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, grp = (lane >> 4) & 1, i16 = lane & 15;
  const int wt = wave % WVT;
  const int wci = (wave / WVT) % WVCI;
  const int wco = wave / (WVT * WVCI);

  const int TW = 1 << a.tw_log2, TH = 1 << a.th_log2, TB = 1 << a.tb_log2;
  const int LW = TW + 2 * PW, LH = TH + 2 * P;
  const int plane = TB * LH * LW;
  const int nxv = NSUB_CI * plane * 4;
  const int H = a.H, W = a.W, HW = H * W;
  const int Hs = a.upsample ? (H >> 1) : H, Ws = a.upsample ? (W >> 1) : W;
  const int HWs = Hs * Ws;
  float* ps = reinterpret_cast<float*>(xs + nxv);  // [TCI/8][16] prologue parameters of this block's ci tile

  int bid = blockIdx.x;
  const int slice = bid % a.nslices;
  bid /= a.nslices;
  const int ci_tile = bid % a.n_ci_tiles;
  const int co_tile = bid / a.n_ci_tiles;
  const int co0 = co_tile * TCO, ci0 = ci_tile * TCI;
  const int cob0 = co0 >> 3, cib0 = ci0 >> 3;

  const int t_begin = (int)(((long long)a.ntiles * slice) / a.nslices);
  const int t_end = (int)(((long long)a.ntiles * (slice + 1)) / a.nslices);

  if (PRO) {
    for (int c = tid; c < TCI; c += NT) {
      float sc = 0.f, sh = 0.f;
      if (ci0 + c < a.Ci) {
        sc = a.pro_invstd[ci0 + c] * a.pro_gamma[ci0 + c];
        sh = a.pro_beta[ci0 + c] - a.pro_mean[ci0 + c] * sc;
      }
      ps[(c >> 3) * 16 + (c & 7)] = sc;
      ps[(c >> 3) * 16 + 8 + (c & 7)] = sh;
    }
  }

  // ---- operand addresses (bytes): lane (i16, grp, hh) reads position k = 8*hh + 4*j + (i16 >> 2), channels
  // 16*grp + 4*(i16 & 3) .. +3 of a 32-channel group
  int dy_addr[NKS][2], x_addr[NKS][2];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int slot = ks * 16 + 8 * hh + 4 * j + (i16 >> 2);
      const int cc = slot & (TW - 1);
      const int rr = (slot >> a.tw_log2) & (TH - 1);
      const int tb = slot >> (a.tw_log2 + a.th_log2);
      const int chb = (16 * grp + 4 * (i16 & 3)) * 2;
      dy_addr[ks][j] = slot * 64 + chb;
      x_addr[ks][j] = ((tb * LH + rr) * LW + cc) * 64 + chb;
    }

  f32x16 acc[TPW][WCO][WCI];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int m = 0; m < WCO; ++m)
#pragma unroll
      for (int n = 0; n < WCI; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][m][n][r] = 0.f;

  const __amdgpu_buffer_rsrc_t xrsrc =
      make_rsrc(a.x, (unsigned long long)a.B * a.Cib * HWs * 16ull);
  const __amdgpu_buffer_rsrc_t dyrsrc =
      make_rsrc(a.dy, (unsigned long long)a.B * a.Cob * HW * 16ull);

  u32x4_t dr[MAXD], xr[MAXX];
  bool x_in[MAXX];

  // ---- staging maps, tile-invariant part (done once): in-tile coordinates and the offset relative to the tile
  // origin.  ((b*Cb + cb)*H + r)*W + c is linear in (b0, r0, c0), so a tile costs each vector one add and three
  // range checks instead of a chain of integer divisions.
  unsigned d_rel[MAXD];
  int d_crd[MAXD];  // tb | rr << 8 | cc << 16, or -1: vector outside the tile image / channel range
#pragma unroll
  for (int p = 0; p < MAXD; ++p) {
    const int v = tid + p * NT;
    d_crd[p] = -1;
    d_rel[p] = 0;
    if (NDY % NT == 0 || v < NDY) {
      const int cbl = v & 3;
      const int slot = (v >> 2) % TPX;
      const int sub = (v >> 2) / TPX;
      const int cc = slot & (TW - 1);
      const int rr = (slot >> a.tw_log2) & (TH - 1);
      const int tb = slot >> (a.tw_log2 + a.th_log2);
      const int cb = cob0 + sub * 4 + cbl;
      if (cb < a.Cob) {
        d_crd[p] = tb | (rr << 8) | (cc << 16);
        d_rel[p] = ((((unsigned)tb * a.Cob + cb) * H + rr) * W + cc) * 16u;
      }
    }
  }
  int x_rel[MAXX];  // signed: the halo reaches above / left of the tile origin
  int x_crd[MAXX];
#pragma unroll
  for (int p = 0; p < MAXX; ++p) {
    const int v = tid + p * NT;
    x_crd[p] = -1;
    x_rel[p] = 0;
    if (v < nxv) {
      const int cbl = v & 3;
      const int pos = (v >> 2) % plane;
      const int sub = (v >> 2) / plane;
      const int cc = pos % LW;
      const int t3 = pos / LW;
      const int rr = t3 % LH;
      const int tb = t3 / LH;
      const int cb = cib0 + sub * 4 + cbl;
      if (cb < a.Cib) {
        x_crd[p] = tb | (rr << 8) | (cc << 16);
        // tile origins are even, so with upsample addressing (r0 + rr - P) >> 1 == (r0 >> 1) + ((rr - P) >> 1)
        const int rs = a.upsample ? ((rr - P) >> 1) : (rr - P), cs = a.upsample ? ((cc - PW) >> 1) : (cc - PW);
        x_rel[p] = (((tb * a.Cib + cb) * Hs + rs) * Ws + cs) * 16;
      }
    }
  }

#define SIVAE_WG_LOAD(T)                                                                                        \
  {                                                                                                             \
    const int tw_i = (T) % a.ntw;                                                                               \
    const int t2 = (T) / a.ntw;                                                                                 \
    const int th_i = t2 % a.nth;                                                                                \
    const int tb_i = t2 / a.nth;                                                                                \
    const int b0 = tb_i << a.tb_log2, r0 = th_i << a.th_log2, c0 = tw_i << a.tw_log2;                           \
    const unsigned d_org = (((unsigned)b0 * a.Cob * H + r0) * W + c0) * 16u;                                    \
    const int r0s = a.upsample ? (r0 >> 1) : r0, c0s = a.upsample ? (c0 >> 1) : c0;                             \
    const unsigned x_org = (((unsigned)b0 * a.Cib * Hs + r0s) * Ws + c0s) * 16u;                                \
    _Pragma("unroll") for (int p = 0; p < MAXD; ++p) {                                                          \
      const int crd = d_crd[p];                                                                                 \
      const bool ok = crd >= 0 && b0 + (crd & 255) < a.B && r0 + ((crd >> 8) & 255) < H && c0 + (crd >> 16) < W; \
      dr[p] = buf_load_u32x4(dyrsrc, ok ? d_org + d_rel[p] : SIVAE_OOB, 0u);                                    \
    }                                                                                                           \
    _Pragma("unroll") for (int p = 0; p < MAXX; ++p) {                                                          \
      const int crd = x_crd[p];                                                                                 \
      const int r = r0 + ((crd >> 8) & 255) - P, c = c0 + (crd >> 16) - PW;                                     \
      const bool ok = crd >= 0 && b0 + (crd & 255) < a.B && r >= 0 && r < H && c >= 0 && c < W;                 \
      x_in[p] = ok;                                                                                             \
      xr[p] = buf_load_u32x4(xrsrc, ok ? (unsigned)((int)x_org + x_rel[p]) : SIVAE_OOB, 0u);                    \
    }                                                                                                           \
  }

  if (t_begin < t_end) SIVAE_WG_LOAD(t_begin)
  if (PRO) __syncthreads();
  for (int t = t_begin; t < t_end; ++t) {
#pragma unroll
    for (int p = 0; p < MAXD; ++p) {
      const int v = tid + p * NT;
      if (NDY % NT == 0 || v < NDY) dys[v] = dr[p];
    }
#pragma unroll
    for (int p = 0; p < MAXX; ++p) {
      const int v = tid + p * NT;
      u32x4_t q = xr[p];
      if (PRO) {
        float f[8];
        unpack8(q, f);
        const int cbt = ((v >> 2) / plane) * 4 + (v & 3);
        const float* pp = ps + cbt * 16;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = x_in[p] ? lrelu01(f[e] * pp[e] + pp[8 + e], a.pro_slope) : 0.f;
        q = pack8(f);
      }
      if (v < nxv) xs[v] = q;
    }
    __syncthreads();
    if (t + 1 < t_end) SIVAE_WG_LOAD(t + 1)

    const unsigned char* dyb = reinterpret_cast<const unsigned char*>(dys);
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(xs);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      bf16x8_t av[WCO];
#pragma unroll
      for (int m = 0; m < WCO; ++m) {
        const int sub = wco * WCO + m;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4_t __attribute__((address_space(3)))*)(dyb + sub * TPX * 64 + dy_addr[ks][0]));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4_t __attribute__((address_space(3)))*)(dyb + sub * TPX * 64 + dy_addr[ks][1]));
        typedef short s16x8_t __attribute__((ext_vector_type(8)));
        s16x8_t v8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v8[e] = lo[e];
          v8[4 + e] = hi[e];
        }
        av[m] = __builtin_bit_cast(bf16x8_t, v8);
      }
#pragma unroll
      for (int tt = 0; tt < TPW; ++tt) {
        const int tap = wt + tt * WVT;  // (wave-uniform)
        if (tap < TAPS) {
          const int kh = tap / KW, kw = tap % KW;
          const int toff = (kh * LW + kw) * 64;
#pragma unroll
          for (int n = 0; n < WCI; ++n) {
            const int sub = wci * WCI + n;
            const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4_t __attribute__((address_space(3)))*)(xb + sub * plane * 64 + x_addr[ks][0] + toff));
            const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4_t __attribute__((address_space(3)))*)(xb + sub * plane * 64 + x_addr[ks][1] + toff));
            typedef short s16x8_t __attribute__((ext_vector_type(8)));
            s16x8_t v8;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v8[e] = lo[e];
              v8[4 + e] = hi[e];
            }
            const bf16x8_t bv = __builtin_bit_cast(bf16x8_t, v8);
#pragma unroll
            for (int m = 0; m < WCO; ++m)
              acc[tt][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[m], bv, acc[tt][m][n], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  }
#undef SIVAE_WG_LOAD

  // ---- partial dW of this slice: part[slice][tap][co][ci]
  float* pbase = a.part + (size_t)slice * TAPS * a.Co * a.Ci;
  const int l31 = lane & 31;
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
    const int tap = wt + tt * WVT;
    if (tap < TAPS) {
#pragma unroll
      for (int m = 0; m < WCO; ++m)
#pragma unroll
        for (int n = 0; n < WCI; ++n) {
          const int ci = ci0 + (wci * WCI + n) * 32 + l31;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = co0 + (wco * WCO + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (co < a.Co && ci < a.Ci) pbase[((size_t)tap * a.Co + co) * a.Ci + ci] = acc[tt][m][n][r];
          }
        }
    }
  }
}

// dw[(co*Ci + ci)*TAPS + tap] = sum_slices part[slice][tap][co][ci]  (fixed order -> deterministic).
// A block owns (co, 64 consecutive ci): NG groups of 64 lanes walk the slices (group g takes slices g, g + NG, ...) with
// one accumulator per tap, loads coalesced along ci; the groups are folded through LDS in group order and the block
// writes its 64 * TAPS results as ONE contiguous run of the [Co][Ci][TAPS] gradient (the tap index is the fastest one
// there, so the transpose happens in LDS, not in strided global stores).
template <int TAPS, int NG>
__global__ void __launch_bounds__(64 * NG) bf16_wgrad_reduce_kernel(const float* __restrict__ part,
                                                                    float* __restrict__ dw, int nslices, int Co,
                                                                    int Ci) {
  __shared__ float red[NG][TAPS][64];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int n_ci = (Ci + 63) / 64;
  const int co = blockIdx.x / n_ci, ci0 = (blockIdx.x % n_ci) * 64;
  const int ci = ci0 + lane;
  float acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t) acc[t] = 0.f;
  if (ci < Ci) {
    const size_t coci = (size_t)Co * Ci;
    const float* p = part + (size_t)co * Ci + ci;
    for (int s = g; s < nslices; s += NG) {
      const float* ps = p + (size_t)s * TAPS * coci;
#pragma unroll
      for (int t = 0; t < TAPS; ++t) acc[t] += ps[(size_t)t * coci];
    }
  }
#pragma unroll
  for (int t = 0; t < TAPS; ++t) red[g][t][lane] = acc[t];
  __syncthreads();
  int n_here = Ci - ci0;
  if (n_here > 64) n_here = 64;
  float* dst = dw + ((size_t)co * Ci + ci0) * TAPS;
  for (int j = threadIdx.x; j < n_here * TAPS; j += 64 * NG) {
    const int c = j / TAPS, t = j - c * TAPS;
    float v = 0.f;
#pragma unroll
    for (int gg = 0; gg < NG; ++gg) v += red[gg][t][c];
    dst[j] = v;
  }
}

template <int TAPS>
static int launch_wg_reduce(const float* part, float* dw, int nslices, int Co, int Ci, hipStream_t stream) {
  const unsigned nb = (unsigned)Co * (unsigned)((Ci + 63) / 64);
  if (nslices >= 32 && TAPS * 16 * 64 * 4 <= 64 * 1024)
    hipLaunchKernelGGL((bf16_wgrad_reduce_kernel<TAPS, (TAPS <= 9 ? 16 : 8)>), dim3(nb), dim3(64 * (TAPS <= 9 ? 16 : 8)), 0,
                       stream, part, dw, nslices, Co, Ci);
  else if (nslices >= 32)
    hipLaunchKernelGGL((bf16_wgrad_reduce_kernel<TAPS, 8>), dim3(nb), dim3(512), 0, stream, part, dw, nslices, Co, Ci);
  else
    hipLaunchKernelGGL((bf16_wgrad_reduce_kernel<TAPS, 4>), dim3(nb), dim3(256), 0, stream, part, dw, nslices, Co, Ci);
  return sivae_launch_status();
}

namespace {

struct WgCfg {
  int TCO, TCI;
};
bool ks_ok(int ks) { return ks == 1 || ks == 3 || ks == 5 || ks == 51; }
int ks_taps(int ks) { return ks == 51 ? 5 : ks * ks; }
WgCfg wg_cfg(int ks, int Co) {
  WgCfg c;
  if (ks == 1) {
    c.TCO = 128;
    c.TCI = 128;
  } else if (ks == 3) {
    c.TCO = 64;
    c.TCI = 64;
  } else if (ks == 51) {  // kw-packed RGB-side layers: one side has <= 16 channels
    c.TCO = Co <= 32 ? 32 : 64;
    c.TCI = Co <= 32 ? 64 : 32;
  } else {
    c.TCO = 32;
    c.TCI = 32;
  }
  return c;
}

int wg_slices(int B, int Ci, int Co, int H, int W, int ks, int* ntiles_out, TileGeom* g_out) {
  const WgCfg c = wg_cfg(ks, Co);
  TileGeom g = make_tile_geom(B, H, W, 64);
  if (ks == 51) {
    // 5 rows x 1 column: the halo is vertical only, so a tall narrow tile (8 wide x 8 high: 12 staged rows for 8)
    // instead of the default 32 x 2 (6 staged rows for 2)
    int tw = next_pow2(W);
    if (tw > 8) tw = 8;
    int th = next_pow2(H);
    if (th > 64 / tw) th = 64 / tw;
    const int tb = 64 / (tw * th);
    g.tw_log2 = ilog2_exact(tw);
    g.th_log2 = ilog2_exact(th);
    g.tb_log2 = ilog2_exact(tb);
    g.ntw = cdiv(W, tw);
    g.nth = cdiv(H, th);
    g.ntb = cdiv(B, tb);
  }
  const int ntiles = g.ntb * g.nth * g.ntw;
  const long long out_tiles = (long long)cdiv(Co, c.TCO) * cdiv(Ci, c.TCI);
  long long ns = (512 + out_tiles - 1) / out_tiles;  // one resident wave of blocks (2 per CU at 2 waves per SIMD)
  if (ns > ntiles) ns = ntiles;
  const long long bytes_per = (long long)Co * Ci * ks_taps(ks) * 4;
  while (ns > 1 && ns * bytes_per > (96ll << 20)) --ns;
  if (ns < 1) ns = 1;
  if (ntiles_out) *ntiles_out = ntiles;
  if (g_out) *g_out = g;
  return (int)ns;
}

template <int KS, int WCO, int WCI, int WVCO, int WVCI, int WVT, int MAXX, bool PRO, int MINW, int KW = KS>
int launch_wg(Bf16WgArgs& a, const TileGeom& g, hipStream_t stream) {
  constexpr int NT = WVCO * WVCI * WVT * 64;
  constexpr int P = KS / 2, PW = KW / 2;
  constexpr int TCO = WVCO * WCO * 32, TCI = WVCI * WCI * 32;
  const int plane = (1 << g.tb_log2) * ((1 << g.th_log2) + 2 * P) * ((1 << g.tw_log2) + 2 * PW);
  const int nxv = (TCI / 32) * plane * 4;
  if (nxv > MAXX * NT) return SIVAE_ERR_SHAPE;
  const size_t lds = (size_t)((TCO / 32) * 64 * 4 + nxv) * 16 + (PRO ? (size_t)(TCI / 8) * 64 : 0);
  const long long nblk = (long long)a.n_co_tiles * a.n_ci_tiles * a.nslices;
  if (nblk > 0x7fffffffLL) return SIVAE_ERR_RANGE;
  auto kern = bf16_wgrad_kernel<KS, WCO, WCI, WVCO, WVCI, WVT, MAXX, PRO, MINW, KW>;
  static size_t lds_hwm = 0;
  const int rc_lds = sivae_ensure_lds(reinterpret_cast<const void*>(kern), lds, &lds_hwm);
  if (rc_lds != SIVAE_OK) return rc_lds;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(NT), lds, stream, a);
  return sivae_launch_status();
}

}  // namespace

extern "C" size_t sivae_bf16_conv2d_wgrad_workspace_bytes(int B, int Ci, int Co, int H, int W, int ks) {
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0 || !ks_ok(ks)) return 0;
  const int ns = wg_slices(B, Ci, Co, H, W, ks, nullptr, nullptr);
  return (size_t)ns * Co * Ci * ks_taps(ks) * sizeof(float);
}

extern "C" int sivae_bf16_conv2d_wgrad(const void* x, const void* dy, float* dw, const float* pro_mean,
                                       const float* pro_invstd, const float* pro_gamma, const float* pro_beta,
                                       float pro_slope, int B, int Ci, int Co, int H, int W, int ks, int upsample,
                                       void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!x || !dy || !dw || !workspace) return SIVAE_ERR_NULL;
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  if (!ks_ok(ks)) return SIVAE_ERR_KSIZE;
  if (upsample && ((H & 1) || (W & 1))) return SIVAE_ERR_SHAPE;
  if (pro_mean && (!pro_invstd || !pro_gamma || !pro_beta)) return SIVAE_ERR_NULL;
  if (pro_mean && ks != 3) return SIVAE_ERR_MODE;
  if (workspace_bytes < sivae_bf16_conv2d_wgrad_workspace_bytes(B, Ci, Co, H, W, ks)) return SIVAE_ERR_WORKSPACE;
  const int Cib = bf16_cblocks(Ci), Cob = bf16_cblocks(Co);
  if ((long long)B * Cib * H * W * 16 >= 0xffffffffLL || (long long)B * Cob * H * W * 16 >= 0xffffffffLL)
    return SIVAE_ERR_RANGE;
  Bf16WgArgs a;
  a.x = x;
  a.dy = dy;
  a.part = reinterpret_cast<float*>(workspace);
  a.pro_mean = pro_mean;
  a.pro_invstd = pro_invstd;
  a.pro_gamma = pro_gamma;
  a.pro_beta = pro_beta;
  a.pro_slope = pro_slope;
  a.B = B;
  a.Ci = Ci;
  a.Co = Co;
  a.H = H;
  a.W = W;
  a.Cib = Cib;
  a.Cob = Cob;
  a.upsample = upsample;
  TileGeom g;
  a.nslices = wg_slices(B, Ci, Co, H, W, ks, &a.ntiles, &g);
  a.tb_log2 = g.tb_log2;
  a.th_log2 = g.th_log2;
  a.tw_log2 = g.tw_log2;
  a.ntb = g.ntb;
  a.nth = g.nth;
  a.ntw = g.ntw;
  const WgCfg c = wg_cfg(ks, Co);
  a.n_co_tiles = cdiv(Co, c.TCO);
  a.n_ci_tiles = cdiv(Ci, c.TCI);
  int rc;
  if (ks == 3) {
    rc = pro_mean ? launch_wg<3, 1, 1, 2, 2, 1, 5, true, 2>(a, g, stream)
                  : launch_wg<3, 1, 1, 2, 2, 1, 5, false, 2>(a, g, stream);
  } else if (ks == 1) {
    rc = launch_wg<1, 2, 2, 2, 2, 1, 4, false, 2>(a, g, stream);
  } else if (ks == 51) {
    rc = Co <= 32 ? launch_wg<5, 1, 1, 1, 2, 2, 6, false, 2, 1>(a, g, stream)
                  : launch_wg<5, 1, 1, 2, 1, 2, 4, false, 2, 1>(a, g, stream);
  } else {
    rc = launch_wg<5, 1, 1, 1, 1, 4, 4, false, 2>(a, g, stream);
  }
  if (rc != SIVAE_OK) return rc;
  if (ks == 51) return launch_wg_reduce<5>(a.part, dw, a.nslices, Co, Ci, stream);
  if (ks == 3) return launch_wg_reduce<9>(a.part, dw, a.nslices, Co, Ci, stream);
  if (ks == 1) return launch_wg_reduce<1>(a.part, dw, a.nslices, Co, Ci, stream);
  return launch_wg_reduce<25>(a.part, dw, a.nslices, Co, Ci, stream);
}
