// Weight gradient of the stride-1 "same" convolution on the exact-fp32 matrix pipe.
//
//   dW[co][ci][kh][kw] = sum_{b,r,c} dY[b][co][r][c] * X[b][ci][r+kh-P][c+kw-P]
//
// GEMM view per tap: M = co (A = dY, i = channel, k = pixel), N = ci (B = shifted X), K = pixels.
// A wave owns a 32(co) x 32(ci) x NTAP block of accumulators (NTAP = 9 for 3x3: 144 registers) so
// one staged dY/X pixel tile feeds all taps: per k-step (2 pixels) 1 + NTAP ds_read_b32 feed NTAP
// MFMAs.  LDS rows are padded to an odd stride, so the channel-strided operand reads are
// conflict-free.  Staging: every wave owns a set of channels, lanes own tile positions, so a load is
// raw-buffer {per-lane 32-bit byte offset, uniform SGPR channel offset}; halo / out-of-image lanes
// carry an out-of-range offset and read 0 (no branches).  The next pixel tile's loads are in
// flight during the MFMA loop of the current one.
// The reduction over pixels is split across blockIdx.y ("slices"); every slice writes its partial
// dW and a second kernel adds the slices in a fixed order — no atomics, results are run-to-run
// reproducible.
//
// Optional prologue: X' = LeakyReLU((X-mean[ci])*invstd[ci]*gamma[ci]+beta[ci]) recomputed on load
// (the forward pass never stored the BatchNorm output), and nearest-2x upsample addressing of X.
//
// Reference op being replaced: the weight half of aten::convolution_backward for the Conv2d /
// Linear layers of soft_intro_vae/train_soft_intro_vae.py:51-61,89,109,146,159.
#include "common.h"
#include <stdlib.h>

struct ConvWgradArgs {
  const float* x;
  const float* dy;
  float* part;  // [n_slices][Co][Ci][KS*KS]
  const float* pro_mean;
  const float* pro_invstd;
  const float* pro_gamma;
  const float* pro_beta;
  float pro_slope;
  int B, Ci, Co, H, W;
  int tb_log2, th_log2, tw_log2;
  int ntb, nth, ntw;
  int n_co_tiles, n_ci_tiles;
  int n_slices, tiles_per_slice, n_tiles;
  unsigned magic_lw, magic_lh;
  int upsample;
};

// KHB = number of kernel rows handled by one block (blockIdx.z selects the row group).
// TPX is fixed to 64 pixels: one wave-instruction stages one channel row of the dY tile.
template <int KS, int KHB, int WM, int WN, int WVM, int WVN, int NPOS, bool PRO>
__global__ void __launch_bounds__(WVM* WVN * 64, 2) conv_wgrad_kernel(ConvWgradArgs a) {
  constexpr int TPX = 64;
  constexpr int P = KS / 2;
  constexpr int NWAVE = WVM * WVN;
  constexpr int TCO = WVM * WM * 32;
  constexpr int TCI = WVN * WN * 32;
  constexpr int NTAP = KHB * KS;
  constexpr int DLD = TPX + 1;        // odd row stride of the dY tile
  constexpr int CPW_D = TCO / NWAVE;  // dY channels staged per wave
  constexpr int CPW_X = TCI / NWAVE;  // X channels staged per wave
  static_assert(TCO % NWAVE == 0 && TCI % NWAVE == 0, "channels must split evenly over waves");

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int wvm = wave / WVN, wvn = wave % WVN;

  const int TW = 1 << a.tw_log2, TH = 1 << a.th_log2, TB = 1 << a.tb_log2;
  const int LW = TW + 2 * P, LH = TH + KHB - 1;
  const int plane = TB * LH * LW;
  const int XLD = plane | 1;  // odd row stride of the X tile
  float* dys = smem;             // [TCO][DLD]
  float* xs = smem + TCO * DLD;  // [TCI][XLD]

  const int H = a.H, W = a.W, HW = H * W;
  const int Hs = a.upsample ? (H >> 1) : H, Ws = a.upsample ? (W >> 1) : W;
  const int HWs = Hs * Ws;

  const int co_tile = blockIdx.x % a.n_co_tiles;
  const int ci_tile = blockIdx.x / a.n_co_tiles;
  const int co0 = co_tile * TCO, ci0 = ci_tile * TCI;
  const int kh0 = blockIdx.z * KHB;
  const int slice = blockIdx.y;
  const int tile_begin = slice * a.tiles_per_slice;
  int tile_end = tile_begin + a.tiles_per_slice;
  if (tile_end > a.n_tiles) tile_end = a.n_tiles;

  // tile-invariant decomposition of this lane's staging positions
  //   dY : pixel p = lane -> (tb, rr, cc)
  //   X  : pos = lane + 64*j -> (tb, rr, cc) in the halo tile
  const int d_cc = lane & (TW - 1);
  const int d_rr = (lane >> a.tw_log2) & (TH - 1);
  const int d_tb = lane >> (a.tw_log2 + a.th_log2);
  int x_cc[NPOS], x_rr[NPOS], x_tb[NPOS];
#pragma unroll
  for (int j = 0; j < NPOS; ++j) {
    const unsigned pos = lane + 64 * j;
    const unsigned t = fastdiv(pos, a.magic_lw);
    x_cc[j] = pos - t * LW;
    const unsigned tb = fastdiv(t, a.magic_lh);
    x_rr[j] = t - tb * LH;
    x_tb[j] = (pos < (unsigned)plane) ? (int)tb : (1 << 20);  // forces OOB
  }

  f32x16 acc[WM][WN][NTAP];
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int n = 0; n < WN; ++n)
#pragma unroll
      for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][t][r] = 0.f;

  float dr[CPW_D];
  float xr[NPOS][CPW_X];
  unsigned x_off[NPOS];  // byte offsets of the tile being loaded (kept for the prologue's validity test)

#define SIVAE_LOAD_TILE(TILE)                                                                          \
  {                                                                                                    \
    const int tw_i = (TILE) % a.ntw;                                                                   \
    const int t2 = (TILE) / a.ntw;                                                                     \
    const int th_i = t2 % a.nth;                                                                       \
    const int tb_i = t2 / a.nth;                                                                       \
    const int b0 = tb_i << a.tb_log2, r0 = th_i << a.th_log2, c0 = tw_i << a.tw_log2;                  \
    int nb_here = a.B - b0;                                                                            \
    if (nb_here > TB) nb_here = TB;                                                                    \
    const __amdgpu_buffer_rsrc_t drs =                                                                 \
        make_rsrc(a.dy + (size_t)b0 * a.Co * HW, (unsigned long long)a.Co * HW * 4ull * nb_here);      \
    const __amdgpu_buffer_rsrc_t xrs =                                                                 \
        make_rsrc(a.x + (size_t)b0 * a.Ci * HWs, (unsigned long long)a.Ci * HWs * 4ull * nb_here);     \
    {                                                                                                  \
      const int r = r0 + d_rr, c = c0 + d_cc;                                                          \
      const unsigned doff = (d_tb < nb_here && r < H && c < W)                                         \
                                ? (((unsigned)d_tb * a.Co * H + r) * W + c) * 4u                       \
                                : SIVAE_OOB;                                                           \
      _Pragma("unroll") for (int k = 0; k < CPW_D; ++k) {                                              \
        int co = co0 + wave * CPW_D + k;                                                               \
        co = co < a.Co ? co : a.Co - 1;                                                                \
        dr[k] = buf_load_f32(drs, doff, (unsigned)co * (unsigned)HW * 4u);                             \
      }                                                                                                \
    }                                                                                                  \
    _Pragma("unroll") for (int j = 0; j < NPOS; ++j) {                                                 \
      const int r = r0 + x_rr[j] + kh0 - P, c = c0 + x_cc[j] - P;                                      \
      unsigned off = SIVAE_OOB;                                                                        \
      if (x_tb[j] < nb_here && r >= 0 && r < H && c >= 0 && c < W) {                                   \
        const int rs = a.upsample ? (r >> 1) : r, cs = a.upsample ? (c >> 1) : c;                      \
        off = (((unsigned)x_tb[j] * a.Ci * Hs + rs) * Ws + cs) * 4u;                                   \
      }                                                                                                \
      x_off[j] = off;                                                                                  \
      _Pragma("unroll") for (int k = 0; k < CPW_X; ++k) {                                              \
        int ci = ci0 + wave * CPW_X + k;                                                               \
        ci = ci < a.Ci ? ci : a.Ci - 1;                                                                \
        xr[j][k] = buf_load_f32(xrs, off, (unsigned)ci * (unsigned)HWs * 4u);                          \
      }                                                                                                \
    }                                                                                                  \
  }

  int a_row[WM], b_row[WN];
#pragma unroll
  for (int m = 0; m < WM; ++m) a_row[m] = ((wvm * WM + m) * 32 + l31) * DLD + hh;
#pragma unroll
  for (int n = 0; n < WN; ++n) b_row[n] = ((wvn * WN + n) * 32 + l31) * XLD;

  if (tile_begin < tile_end) SIVAE_LOAD_TILE(tile_begin)
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    // registers -> LDS (the BatchNorm/LeakyReLU prologue is applied here, not at load time)
#pragma unroll
    for (int k = 0; k < CPW_D; ++k) dys[(wave * CPW_D + k) * DLD + lane] = dr[k];
#pragma unroll
    for (int k = 0; k < CPW_X; ++k) {
      float pm = 0.f, pg = 1.f, pb = 0.f;
      if (PRO) {
        int ci = ci0 + wave * CPW_X + k;
        ci = ci < a.Ci ? ci : a.Ci - 1;
        pm = a.pro_mean[ci];
        pg = a.pro_invstd[ci] * a.pro_gamma[ci];
        pb = a.pro_beta[ci];
      }
#pragma unroll
      for (int j = 0; j < NPOS; ++j) {
        const int pos = lane + 64 * j;
        float v = xr[j][k];
        if (PRO) v = (x_off[j] != SIVAE_OOB) ? lrelu((v - pm) * pg + pb, a.pro_slope) : 0.f;
        if (pos < plane) xs[(wave * CPW_X + k) * XLD + pos] = v;
      }
    }
    __syncthreads();
    if (tile + 1 < tile_end) SIVAE_LOAD_TILE(tile + 1)

#pragma unroll 2
    for (int s = 0; s < TPX / 2; ++s) {
      const int p = 2 * s + hh;
      const int cc = p & (TW - 1);
      const int rr = (p >> a.tw_log2) & (TH - 1);
      const int tb = p >> (a.tw_log2 + a.th_log2);
      const int xo = (tb * LH + rr) * LW + cc;
      float av[WM];
#pragma unroll
      for (int m = 0; m < WM; ++m) av[m] = dys[a_row[m] + 2 * s];
#pragma unroll
      for (int k = 0; k < KHB; ++k) {
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
          float bv[WN];
#pragma unroll
          for (int n = 0; n < WN; ++n) bv[n] = xs[b_row[n] + xo + k * LW + kw];
#pragma unroll
          for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n)
              acc[m][n][k * KS + kw] =
                  __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m][n][k * KS + kw], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
#undef SIVAE_LOAD_TILE

  // ---- write this slice's partial:  part[slice][co][ci][tap]
  float* outp = a.part + (size_t)slice * a.Co * a.Ci * (KS * KS);
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int n = 0; n < WN; ++n) {
      const int ci = ci0 + (wvn * WN + n) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + (wvm * WM + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (co < a.Co && ci < a.Ci) {
#pragma unroll
          for (int t = 0; t < NTAP; ++t)
            outp[((size_t)co * a.Ci + ci) * (KS * KS) + kh0 * KS + t] = acc[m][n][t][r];
        }
      }
    }
}

namespace {

struct WgradPlan {
  TileGeom g;
  int n_tiles, n_slices, tiles_per_slice, n_co_tiles, n_ci_tiles, khb;
  int tco, tci;
};

// tile configuration table (must match the launch dispatch below)
static int wgrad3_variant() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("SIVAE_WGRAD3_VARIANT");
    v = e ? atoi(e) : 0;
  }
  return v;
}

static int wgrad_plan(int B, int Ci, int Co, int H, int W, int ks, WgradPlan* p) {
  if (ks == 3) {
    p->tco = (Co <= 64 || wgrad3_variant() != 3) ? 64 : 128; p->tci = 64; p->khb = wgrad3_variant() == 2 ? 1 : 3;
  } else if (ks == 1) {
    p->tco = 128; p->tci = 128; p->khb = 1;
  } else if (ks == 5) {
    p->khb = 1;
    if (Co <= 32) { p->tco = 32; p->tci = 64; }
    else { p->tco = 64; p->tci = 32; }
  } else {
    return SIVAE_ERR_KSIZE;
  }
  p->g = make_tile_geom(B, H, W, 64);
  p->n_tiles = p->g.ntb * p->g.nth * p->g.ntw;
  p->n_co_tiles = cdiv(Co, p->tco);
  p->n_ci_tiles = cdiv(Ci, p->tci);
  const int out_tiles = p->n_co_tiles * p->n_ci_tiles * (ks / p->khb);
  // aim for ~1024 blocks, at least 4 pixel tiles per slice when possible
  int want = cdiv(1024, out_tiles);
  if (want < 1) want = 1;
  int tps = cdiv(p->n_tiles, want);
  if (tps < 4) tps = 4;
  if (tps > p->n_tiles) tps = p->n_tiles;
  p->tiles_per_slice = tps;
  p->n_slices = cdiv(p->n_tiles, tps);
  return SIVAE_OK;
}

template <int KS, int KHB, int WM, int WN, int WVM, int WVN, int NPOS>
int launch_wgrad(ConvWgradArgs& a, const WgradPlan& p, hipStream_t stream) {
  constexpr int NT = WVM * WVN * 64;
  constexpr int TCO = WVM * WM * 32;
  constexpr int TCI = WVN * WN * 32;
  constexpr int P = KS / 2;
  const int TW = 1 << p.g.tw_log2, TH = 1 << p.g.th_log2, TB = 1 << p.g.tb_log2;
  const int LW = TW + 2 * P, LH = TH + KHB - 1;
  const int plane = TB * LH * LW;
  if (plane > NPOS * 64) return SIVAE_ERR_SHAPE;
  a.magic_lw = make_magic(LW);
  a.magic_lh = make_magic(LH);
  const size_t lds = (size_t)(TCO * 65 + TCI * (plane | 1)) * sizeof(float);
  auto kern = a.pro_mean ? conv_wgrad_kernel<KS, KHB, WM, WN, WVM, WVN, NPOS, true>
                         : conv_wgrad_kernel<KS, KHB, WM, WN, WVM, WVN, NPOS, false>;
  {
    static size_t lds_hwm[2] = {0, 0};  // per template instantiation, per prologue variant
    const int rc_lds = sivae_ensure_lds(reinterpret_cast<const void*>(kern), lds, &lds_hwm[a.pro_mean ? 1 : 0]);
    if (rc_lds != SIVAE_OK) return rc_lds;
  }
  dim3 grid(p.n_co_tiles * p.n_ci_tiles, p.n_slices, KS / KHB);
  hipLaunchKernelGGL(kern, grid, dim3(NT), lds, stream, a);
  return sivae_launch_status();
}

}  // namespace

extern "C" size_t sivae_conv2d_wgrad_workspace_bytes(int B, int Ci, int Co, int H, int W, int ks) {
  WgradPlan p;
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return 0;
  if (wgrad_plan(B, Ci, Co, H, W, ks, &p) != SIVAE_OK) return 0;
  return (size_t)p.n_slices * Co * Ci * ks * ks * sizeof(float);
}

extern "C" int sivae_conv2d_wgrad(const float* x, const float* dy, float* dw, const float* pro_mean,
                                  const float* pro_invstd, const float* pro_gamma, const float* pro_beta,
                                  float pro_slope, int B, int Ci, int Co, int H, int W, int ks, int upsample,
                                  void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!x || !dy || !dw) return SIVAE_ERR_NULL;
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  if (ks != 1 && ks != 3 && ks != 5) return SIVAE_ERR_KSIZE;
  if (upsample && ((H & 1) || (W & 1))) return SIVAE_ERR_SHAPE;
  if (pro_mean && (!pro_invstd || !pro_gamma || !pro_beta)) return SIVAE_ERR_NULL;
  const long long hw = (long long)H * W;
  if ((long long)Ci * hw * 4 >= 0x7fffffffLL || (long long)Co * hw * 4 >= 0x7fffffffLL) return SIVAE_ERR_RANGE;
  WgradPlan p;
  int rc = wgrad_plan(B, Ci, Co, H, W, ks, &p);
  if (rc != SIVAE_OK) return rc;
  const size_t numel = (size_t)Co * Ci * ks * ks;
  const size_t need = (size_t)p.n_slices * numel * sizeof(float);
  if (!workspace || workspace_bytes < need) return SIVAE_ERR_WORKSPACE;

  ConvWgradArgs a;
  a.x = x;
  a.dy = dy;
  a.part = (float*)workspace;
  a.pro_mean = pro_mean;
  a.pro_invstd = pro_invstd;
  a.pro_gamma = pro_gamma;
  a.pro_beta = pro_beta;
  a.pro_slope = pro_slope;
  a.B = B; a.Ci = Ci; a.Co = Co; a.H = H; a.W = W;
  a.tb_log2 = p.g.tb_log2; a.th_log2 = p.g.th_log2; a.tw_log2 = p.g.tw_log2;
  a.ntb = p.g.ntb; a.nth = p.g.nth; a.ntw = p.g.ntw;
  a.n_co_tiles = p.n_co_tiles; a.n_ci_tiles = p.n_ci_tiles;
  a.n_slices = p.n_slices; a.tiles_per_slice = p.tiles_per_slice; a.n_tiles = p.n_tiles;
  a.upsample = upsample;

  // <KS, KHB, WM, WN, WVM, WVN, NPOS>: NPOS*64 >= largest halo plane of a 64-pixel tile
  if (ks == 3 && wgrad3_variant() == 2) rc = launch_wgrad<3, 1, 1, 1, 2, 2, 2>(a, p, stream);  // one kernel row per block
  else if (ks == 3 && (Co <= 64 || wgrad3_variant() != 3)) rc = launch_wgrad<3, 3, 1, 1, 2, 2, 3>(a, p, stream);  // production: co64 x ci64, 4 waves
  else if (ks == 3) rc = launch_wgrad<3, 3, 1, 1, 4, 2, 3>(a, p, stream);   // co128 x ci64, 8 waves
  else if (ks == 1) rc = launch_wgrad<1, 1, 2, 2, 2, 2, 1>(a, p, stream);   // co128 x ci128, 4 waves
  else if (Co <= 32) rc = launch_wgrad<5, 1, 1, 1, 1, 2, 2>(a, p, stream);  // co32 x ci64, 2 waves
  else rc = launch_wgrad<5, 1, 1, 1, 2, 1, 2>(a, p, stream);                // co64 x ci32, 2 waves
  if (rc != SIVAE_OK) return rc;

  sivae_launch_slice_reduce((const float*)workspace, dw, p.n_slices, numel, stream);
  return sivae_launch_status();
}
