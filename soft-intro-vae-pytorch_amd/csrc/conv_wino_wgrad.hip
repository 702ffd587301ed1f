// Winograd-domain weight gradient of the 3x3 stride-1 "same" convolution (fp32 MFMA, gfx950).
//
// With Y = A^T [ (G g G^T) . (B^T d B) ] A per 2x2 output tile (conv_wino.hip), the gradient with respect
// to the filter is linear in the same transformed operands:
//
//   dU[i][j][co][ci] = sum over tiles  Mg[i][j][co][tile] * V[i][j][ci][tile]     Mg = A dY A^T  (2x2 -> 4x4)
//   dg[co][ci]       = G^T dU[.][.][co][ci] G                                     V  = B^T d  B   (4x4 -> 4x4)
//
// i.e. 16 independent GEMMs with K = number of tiles: 16 multiplies per (tile, co, ci) instead of the 36 of
// the direct form (9 taps x 4 pixels) — 2.25x fewer MFMA passes for the weight half of
// aten::convolution_backward of the nn.Conv2d(k=3) layers (soft_intro_vae/train_soft_intro_vae.py:56-61).
//
// Work split (same idea as the forward kernel): a block is 64 output channels x 32*NGI input channels; wave
// (g, j) owns input-channel group g and frequency column j (4 frequencies x 2 co-subtiles = 128 accumulator
// registers).  MFMA roles: A = Mg (row = co, k = tile), B = V (k = tile, col = ci); a k-step is a pair of
// horizontally adjacent tiles.  Both operands are built in registers from raw LDS reads (4 dY values and
// 8 halo values per lane and k-step); neither transformed tensor is ever stored.
// The tile dimension is cut into stages of 16 tiles (4 x 16 pixels: 64 dY pixels, a 6 x 18 input halo), staged
// global -> registers -> LDS with the next stage in flight during the MFMA phase (double-buffered LDS, one
// barrier per stage).  Channel rows in LDS have odd strides (109 / 65 floats), so the channel-strided operand
// reads are bank-conflict free.
// The sum over tiles is split across blocks ("slices"); every slice writes its partial dU and a second kernel
// adds the slices in a fixed order and applies G^T . G — no atomics, run-to-run reproducible.
//
// Optional prologue (as in conv_wgrad.hip): X' = LeakyReLU((X-mean)*invstd*gamma+beta) recomputed on load, and
// nearest-2x upsample addressing of X.
#include "common.h"
#include <stdlib.h>

struct WinoWgArgs {
  const float* x;
  const float* dy;
  float* ws;  // [n_slices][16][Co_pad][Ci_pad]
  const float* pro_mean;
  const float* pro_invstd;
  const float* pro_gamma;
  const float* pro_beta;
  float pro_slope;
  int pro_seg_images, pro_nseg;  // segments (see conv_wino.hip): pro_mean / pro_invstd are [pro_nseg][Ci]
  int B, Ci, Co, H, W;
  int Ci_pad, Co_pad;
  int nrh, nrw, nstages, sps;
  int n_co_tiles, n_ci_tiles;
  int upsample;
  int nblk, xcd_remap;
};

template <bool PRO, int NGI>
__global__ void __launch_bounds__(NGI * 256, 2) wino_wgrad_kernel(WinoWgArgs a) {
  constexpr int NT = NGI * 256, NW = NGI * 4;
  constexpr int CIT = 32 * NGI, COT = 64;
  constexpr int LWX = 18, NPOSX = 6 * LWX, XP = NPOSX + 1;  // 6 x 18 halo, odd channel stride
  constexpr int YP = 65;                                    // 4 x 16 pixels, odd channel stride
  constexpr int XBUF = CIT * XP, YBUF = COT * YP;
  constexpr int NSUB = NT / 128;    // x staging: 128 halo slots per channel row, NSUB channels per pass
  constexpr int XQ = CIT / NSUB;    // x loads per thread and stage (16)
  constexpr int YQ = COT / NW;      // dY loads per thread and stage (16 / 8)

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;              // [2][CIT][XP]
  float* dys = smem + 2 * XBUF;  // [2][COT][YP]
  // fused-BatchNorm parameters {mean, invstd*gamma, beta, -} of this block's CIT input channels (LDS instead of
  // three scalar loads per channel and stage: those serialise on lgkmcnt(0) in the middle of the MFMA phase)
  float4* pro4 = reinterpret_cast<float4*>(smem + 2 * XBUF + 2 * YBUF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int wj = wave & 3, wgi = wave >> 2;
  const int H = a.H, W = a.W, HW = H * W;
  const int Hs = a.upsample ? (H >> 1) : H, Ws = a.upsample ? (W >> 1) : W;
  const int HWs = Hs * Ws;

  const int ntiles = a.n_co_tiles * a.n_ci_tiles;
  // XCD-aware block order: consecutive blockIdx go round-robin to the 8 XCDs, so logical block L = xcd * M + j puts the
  // (co, ci) tiles of one pixel slice on ONE XCD at the same time — they read the same x / dY stages, which then come
  // out of that XCD's L2 once instead of once per XCD
  int lb = (int)blockIdx.x;
  if (a.xcd_remap) {
    lb = (lb & 7) * ((int)gridDim.x >> 3) + (lb >> 3);
    if (lb >= a.nblk) return;
  }
  const int tile = lb % ntiles, slice = lb / ntiles;
  const int ci0 = (tile % a.n_ci_tiles) * CIT, co0 = (tile / a.n_ci_tiles) * COT;
  const int s_begin = slice * a.sps;
  const int s_end = (s_begin + a.sps < a.nstages) ? (s_begin + a.sps) : a.nstages;

  // ---- staging maps
  const int xsub = __builtin_amdgcn_readfirstlane(tid >> 7);  // channel phase of this wave's x loads
  // halo slot of this thread; slots 108..127 duplicate 0..19 (same address, same value) so that the
  // register -> LDS copy needs no predication and the prologue no branch
  const int xpos = (tid & 127) < NPOSX ? (tid & 127) : (tid & 127) - NPOSX;
  const int xrr = xpos / LWX, xcc = xpos % LWX;
  float xmask = 0.f;
  int pseg = 0;  // table offset of the segment of the stage currently held in xr (wave-uniform)
  const f32x2 pslope2 = {a.pro_slope, a.pro_slope};
  const int ypy = lane >> 4, ypx = lane & 15;

  // ---- operand bases.  k-step kk = tile pair (ty = kk >> 2, tx = 2*(kk & 3) + hh)
  const int ca = (wj == 0) ? 0 : ((wj == 2) ? 2 : 1);
  const int cb = (wj == 0) ? 2 : ((wj == 1) ? 2 : ((wj == 2) ? 1 : 3));
  const float sgn = (wj == 1) ? 1.f : -1.f;
  const float pj = (wj == 3) ? 0.f : 1.f;
  const float qj = (wj == 0) ? 0.f : ((wj == 1) ? 1.f : -1.f);
  const int bx = (wgi * 32 + l31) * XP + 2 * hh;
  const int base_a = bx + ca, base_b = bx + cb;
  const int base_y = l31 * YP + 2 * hh;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][m][r] = 0.f;

  float xr[XQ], yr[YQ];
  unsigned xoff = SIVAE_OOB;  // of the stage currently held in xr (needed by the prologue at store time)

#define WG_LOAD(S)                                                                                  \
  {                                                                                                 \
    const int s_ = (S);                                                                             \
    const int b_ = s_ / (a.nrh * a.nrw);                                                            \
    const int rem_ = s_ - b_ * (a.nrh * a.nrw);                                                     \
    const int ry_ = rem_ / a.nrw, rx_ = rem_ - ry_ * a.nrw;                                         \
    const int r0_ = ry_ * 4, c0_ = rx_ * 16;                                                        \
    if (PRO && a.pro_nseg > 1) pseg = (b_ / a.pro_seg_images) * CIT;                                \
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(a.x + (size_t)b_ * a.Ci * HWs, (unsigned long long)a.Ci * HWs * 4ull); \
    const __amdgpu_buffer_rsrc_t yrs = make_rsrc(a.dy + (size_t)b_ * a.Co * HW, (unsigned long long)a.Co * HW * 4ull);   \
    {                                                                                               \
      const int r = r0_ + xrr - 1, c = c0_ + xcc - 1;                                               \
      xoff = SIVAE_OOB;                                                                             \
      xmask = 0.f;                                                                                  \
      if (r >= 0 && r < H && c >= 0 && c < W) {                                                     \
        xmask = 1.f;                                                                                \
        const int rs = a.upsample ? (r >> 1) : r, cs = a.upsample ? (c >> 1) : c;                   \
        xoff = (unsigned)(rs * Ws + cs) * 4u;                                                       \
      }                                                                                             \
    }                                                                                               \
    unsigned yoff = SIVAE_OOB;                                                                      \
    {                                                                                               \
      const int r = r0_ + ypy, c = c0_ + ypx;                                                       \
      if (r < H && c < W) yoff = (unsigned)(r * W + c) * 4u;                                        \
    }                                                                                               \
    _Pragma("unroll") for (int q = 0; q < XQ; ++q) {                                                \
      const int ci = ci0 + xsub + NSUB * q;                                                         \
      const int cic = ci < a.Ci ? ci : a.Ci - 1;                                                    \
      xr[q] = buf_load_f32(xrs, xoff, (unsigned)cic * (unsigned)HWs * 4u);                          \
    }                                                                                               \
    _Pragma("unroll") for (int q = 0; q < YQ; ++q) {                                                \
      const int co = co0 + wave + NW * q;                                                           \
      const int coc = co < a.Co ? co : a.Co - 1;                                                    \
      yr[q] = buf_load_f32(yrs, yoff, (unsigned)coc * (unsigned)HW * 4u);                           \
    }                                                                                               \
  }
#define WG_STORE(BUF)                                                                               \
  {                                                                                                 \
    if (PRO) { /* two channels per packed-fp32 op */                                                \
      _Pragma("unroll") for (int pq = 0; pq < XQ / 2; ++pq) {                                       \
        const float4 p0 = pro4[pseg + 2 * (pq * NSUB + xsub)];     /* mean mean' scale scale' */    \
        const float4 p1 = pro4[pseg + 2 * (pq * NSUB + xsub) + 1]; /* beta beta' */                 \
        f32x2 v = {xr[2 * pq], xr[2 * pq + 1]};                                                     \
        const f32x2 pm = {p0.x, p0.y}, ps = {p0.z, p0.w}, pb = {p1.x, p1.y};                        \
        v = (v - pm) * ps + pb;                                                                     \
        const f32x2 u = v * pslope2;                                                                \
        v.x = fmaxf(v.x, u.x);                                                                      \
        v.y = fmaxf(v.y, u.y);                                                                      \
        v = v * f32x2{xmask, xmask};                                                                \
        xs[(BUF)*XBUF + (xsub + NSUB * (2 * pq)) * XP + xpos] = v.x;                                \
        xs[(BUF)*XBUF + (xsub + NSUB * (2 * pq + 1)) * XP + xpos] = v.y;                            \
      }                                                                                             \
    } else {                                                                                        \
      _Pragma("unroll") for (int q = 0; q < XQ; ++q) xs[(BUF)*XBUF + (xsub + NSUB * q) * XP + xpos] = xr[q]; \
    }                                                                                               \
    _Pragma("unroll") for (int q = 0; q < YQ; ++q) dys[(BUF)*YBUF + (wave + NW * q) * YP + lane] = yr[q]; \
  }
  // raw operand reads of k-step KK: 2 columns x 4 rows of the halo, and the 2x2 dY patch of both co-subtiles
#define WG_READ(BUF, KK, DA, DB, DY)                                                                \
  {                                                                                                 \
    const float* pa = xs + (BUF)*XBUF + base_a + 2 * ((KK) >> 2) * LWX + 4 * ((KK)&3);             \
    const float* pb_ = xs + (BUF)*XBUF + base_b + 2 * ((KK) >> 2) * LWX + 4 * ((KK)&3);            \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                 \
      DA[r] = pa[r * LWX];                                                                          \
      DB[r] = pb_[r * LWX];                                                                         \
    }                                                                                               \
    const float* py = dys + (BUF)*YBUF + base_y + 2 * ((KK) >> 2) * 16 + 4 * ((KK)&3);              \
    _Pragma("unroll") for (int m = 0; m < 2; ++m) {                                                 \
      DY[m][0] = py[m * 32 * YP + 0];                                                               \
      DY[m][1] = py[m * 32 * YP + 1];                                                               \
      DY[m][2] = py[m * 32 * YP + 16];                                                              \
      DY[m][3] = py[m * 32 * YP + 17];                                                              \
    }                                                                                               \
  }
#define WG_STEP(DA, DB, DY)                                                                         \
  {                                                                                                 \
    const float t0 = DA[0] + sgn * DB[0], t1 = DA[1] + sgn * DB[1];                                 \
    const float t2_ = DA[2] + sgn * DB[2], t3 = DA[3] + sgn * DB[3];                                \
    const float v0 = t0 - t2_, v1 = t1 + t2_, v2 = t2_ - t1, v3 = t1 - t3;                          \
    _Pragma("unroll") for (int m = 0; m < 2; ++m) {                                                 \
      const float u0 = pj * DY[m][0] + qj * DY[m][1];                                               \
      const float u1 = pj * DY[m][2] + qj * DY[m][3];                                               \
      acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0, v0, acc[0][m], 0, 0, 0);                 \
      acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0 + u1, v1, acc[1][m], 0, 0, 0);            \
      acc[2][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0 - u1, v2, acc[2][m], 0, 0, 0);            \
      acc[3][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(-u1, v3, acc[3][m], 0, 0, 0);                \
    }                                                                                               \
  }
#define WG_KSTEP(BUF, KK, DA, DB, DY, DAN, DBN, DYN)                                                \
  {                                                                                                 \
    if ((KK) + 1 < 8) WG_READ(BUF, (KK) + 1, DAN, DBN, DYN)                                         \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    WG_STEP(DA, DB, DY)                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                              \
  }
  // MFMA phase of one stage on buffer BUF; the next stage (already loading into xr / yr) is written to the
  // other buffer late in the phase, so only the barrier separates two phases
#define WG_MMA(S, BUF)                                                                              \
  {                                                                                                 \
    float da0[4], db0[4], dy0[2][4], da1[4], db1[4], dy1[2][4];                                     \
    const bool next_ = (S) + 1 < s_end;                                                             \
    if (next_) WG_LOAD((S) + 1)                                                                     \
    WG_READ(BUF, 0, da0, db0, dy0)                                                                  \
    WG_KSTEP(BUF, 0, da0, db0, dy0, da1, db1, dy1)                                                  \
    WG_KSTEP(BUF, 1, da1, db1, dy1, da0, db0, dy0)                                                  \
    WG_KSTEP(BUF, 2, da0, db0, dy0, da1, db1, dy1)                                                  \
    WG_KSTEP(BUF, 3, da1, db1, dy1, da0, db0, dy0)                                                  \
    WG_KSTEP(BUF, 4, da0, db0, dy0, da1, db1, dy1)                                                  \
    WG_KSTEP(BUF, 5, da1, db1, dy1, da0, db0, dy0)                                                  \
    WG_KSTEP(BUF, 6, da0, db0, dy0, da1, db1, dy1)                                                  \
    if (next_) WG_STORE((BUF) ^ 1)                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    WG_KSTEP(BUF, 7, da1, db1, dy1, da0, db0, dy0)                                                  \
    __syncthreads();                                                                                \
  }

  if (PRO) {
    // pair table: entry e = pq * NSUB + xsub holds the channels (xsub + NSUB*2pq, xsub + NSUB*(2pq+1)) a thread
    // stages in its registers xr[2pq], xr[2pq+1]:  pro4[2e] = {mean, mean', scale, scale'}, pro4[2e+1] = {beta, beta'}
    for (int idx = tid; idx < a.pro_nseg * CIT; idx += NT) {
      const int t_ = idx % CIT, so = (idx / CIT) * a.Ci;  // (segment g's statistics start at g * Ci)
      const int e = t_ >> 1, pq = e / NSUB, xs_ = e % NSUB;
      const int ca_ = ci0 + xs_ + NSUB * (2 * pq), cb_ = ca_ + NSUB;
      const int c0_ = ca_ < a.Ci ? ca_ : a.Ci - 1, c1_ = cb_ < a.Ci ? cb_ : a.Ci - 1;
      if ((t_ & 1) == 0)
        pro4[idx] = make_float4(a.pro_mean[so + c0_], a.pro_mean[so + c1_], a.pro_invstd[so + c0_] * a.pro_gamma[c0_],
                                a.pro_invstd[so + c1_] * a.pro_gamma[c1_]);
      else
        pro4[idx] = make_float4(a.pro_beta[c0_], a.pro_beta[c1_], 0.f, 0.f);
    }
    __syncthreads();
  }
  if (s_begin < s_end) {
    WG_LOAD(s_begin)
    WG_STORE(0)
    __syncthreads();
    int s = s_begin;
    for (; s + 1 < s_end; s += 2) {
      WG_MMA(s, 0)
      WG_MMA(s + 1, 1)
    }
    if (s < s_end) WG_MMA(s, 0)
  }
#undef WG_LOAD
#undef WG_STORE
#undef WG_READ
#undef WG_STEP
#undef WG_KSTEP
#undef WG_MMA

  // ---- partial dU of this slice: acc[i][m][r] -> frequency (i, wj), co = co0 + m*32 + row(r, hh), ci = l31
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float* base = a.ws + ((size_t)(slice * 16 + i * 4 + wj) * a.Co_pad + co0) * a.Ci_pad + ci0 + wgi * 32 + l31;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        base[(size_t)row * a.Ci_pad] = acc[i][m][r];
      }
  }
}

// dW[co][ci] = G^T (sum over slices dU[.][.][co][ci]) G.  Block = one co x 64 ci x 4 slice phases.
__global__ void __launch_bounds__(256) wino_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                                int Co, int Ci, int Co_pad, int Ci_pad,
                                                                int n_slices) {
  __shared__ float red[3][16][64];
  const int cil = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int n_cic = (Ci + 63) / 64;
  const int co = blockIdx.x / n_cic, ci = (blockIdx.x % n_cic) * 64 + cil;
  float u[16];
#pragma unroll
  for (int f = 0; f < 16; ++f) u[f] = 0.f;
  if (ci < Ci) {
    for (int s = ph; s < n_slices; s += 4) {
      const float* p = ws + ((size_t)(s * 16) * Co_pad + co) * Ci_pad + ci;
#pragma unroll
      for (int f = 0; f < 16; ++f) u[f] += p[(size_t)f * Co_pad * Ci_pad];
    }
  }
  if (ph > 0) {
#pragma unroll
    for (int f = 0; f < 16; ++f) red[ph - 1][f][cil] = u[f];
  }
  __syncthreads();
  if (ph == 0 && ci < Ci) {
#pragma unroll
    for (int f = 0; f < 16; ++f) u[f] = ((u[f] + red[0][f][cil]) + red[1][f][cil]) + red[2][f][cil];
    // t[r][j] = sum_i G[i][r] u[i][j];  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
    float t[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float u0 = u[0 * 4 + j], u1 = u[1 * 4 + j], u2 = u[2 * 4 + j], u3 = u[3 * 4 + j];
      t[0][j] = u0 + 0.5f * (u1 + u2);
      t[1][j] = 0.5f * (u1 - u2);
      t[2][j] = 0.5f * (u1 + u2) + u3;
    }
    float* dst = dw + ((size_t)co * Ci + ci) * 9;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      dst[r * 3 + 0] = t[r][0] + 0.5f * (t[r][1] + t[r][2]);
      dst[r * 3 + 1] = 0.5f * (t[r][1] - t[r][2]);
      dst[r * 3 + 2] = 0.5f * (t[r][1] + t[r][2]) + t[r][3];
    }
  }
}

namespace {
struct WinoWgPlan {
  int Ci_pad, Co_pad, n_ci_tiles, n_co_tiles, nrh, nrw, nstages, sps, n_slices;
};
constexpr int WG_NGI = 1;

int wino_wg_plan(int B, int Ci, int Co, int H, int W, WinoWgPlan* p) {
  const int CIT = 32 * WG_NGI;
  p->n_ci_tiles = cdiv(Ci, CIT);
  p->n_co_tiles = cdiv(Co, 64);
  p->Ci_pad = p->n_ci_tiles * CIT;
  p->Co_pad = p->n_co_tiles * 64;
  p->nrh = cdiv(H, 4);
  p->nrw = cdiv(W, 16);
  const long long ns = (long long)B * p->nrh * p->nrw;
  if (ns > 0x3fffffffLL) return SIVAE_ERR_RANGE;
  p->nstages = (int)ns;
  const int ntiles = p->n_ci_tiles * p->n_co_tiles;
  // enough blocks for two full rounds of the 512 block slots, but at least 16 stages per slice so the
  // 128 KB partial-dU write-out of a block stays small next to its MFMA work
  // SIVAE_WG_SLOTS: target number of blocks (default 1024 = two rounds of the 512 block slots)
  static int slots = 0;
  if (slots == 0) {
    const char* e = getenv("SIVAE_WG_SLOTS");
    slots = e ? atoi(e) : 1024;
    if (slots <= 0) slots = 1024;
  }
  int n_slices = cdiv(slots, ntiles);
  const int max_slices = p->nstages / 16 > 0 ? p->nstages / 16 : 1;
  if (n_slices > max_slices) n_slices = max_slices;
  p->sps = cdiv(p->nstages, n_slices);
  p->n_slices = cdiv(p->nstages, p->sps);
  return SIVAE_OK;
}
}  // namespace

// maps the Winograd-domain weight gradient takes (its stage is a 4 x 16 pixel region)
extern "C" int sivae_conv2d_wino_wgrad_supported(int H, int W) {
  return (H >= 8 && W >= 16 && !(H & 1) && !(W & 1)) ? 1 : 0;
}

extern "C" size_t sivae_conv2d_wino_wgrad_workspace_bytes(int B, int Ci, int Co, int H, int W) {
  WinoWgPlan p;
  if (B <= 0 || Ci <= 0 || Co <= 0 || !sivae_conv2d_wino_wgrad_supported(H, W)) return 0;
  if (wino_wg_plan(B, Ci, Co, H, W, &p) != SIVAE_OK) return 0;
  return (size_t)p.n_slices * 16 * p.Co_pad * p.Ci_pad * sizeof(float);
}

static int wino_wgrad_impl(const float* x, const float* dy, float* dw, const float* pro_mean,
                           const float* pro_invstd, const float* pro_gamma, const float* pro_beta,
                           float pro_slope, int B, int Ci, int Co, int H, int W, int upsample,
                           void* workspace, size_t workspace_bytes, hipStream_t stream, int pro_seg_images) {
  if (!x || !dy || !dw || !workspace) return SIVAE_ERR_NULL;
  if (pro_seg_images < 0 || (pro_seg_images > 0 && B % pro_seg_images != 0)) return SIVAE_ERR_SHAPE;
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  if (!sivae_conv2d_wino_wgrad_supported(H, W)) return SIVAE_ERR_SHAPE;
  if (pro_mean && (!pro_invstd || !pro_gamma || !pro_beta)) return SIVAE_ERR_NULL;
  if (pro_mean && !(pro_slope >= 0.f && pro_slope <= 1.f)) return SIVAE_ERR_MODE;  // prologue uses max(v, v*slope)
  const long long hw = (long long)H * W;
  if ((long long)Ci * hw * 4 >= 0x7fffffffLL || (long long)Co * hw * 4 >= 0x7fffffffLL) return SIVAE_ERR_RANGE;
  WinoWgPlan p;
  int rc = wino_wg_plan(B, Ci, Co, H, W, &p);
  if (rc != SIVAE_OK) return rc;
  const size_t need = (size_t)p.n_slices * 16 * p.Co_pad * p.Ci_pad * sizeof(float);
  if (workspace_bytes < need) return SIVAE_ERR_WORKSPACE;
  WinoWgArgs a;
  a.x = x;
  a.dy = dy;
  a.ws = static_cast<float*>(workspace);
  a.pro_mean = pro_mean;
  a.pro_invstd = pro_invstd;
  a.pro_gamma = pro_gamma;
  a.pro_beta = pro_beta;
  a.pro_slope = pro_slope;
  a.pro_seg_images = pro_seg_images > 0 ? pro_seg_images : B;
  a.pro_nseg = B / a.pro_seg_images;
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
  a.upsample = upsample;
  const long long nblk = (long long)p.n_ci_tiles * p.n_co_tiles * p.n_slices;
  if (nblk > 0x7fffffffLL) return SIVAE_ERR_RANGE;
  constexpr int CIT = 32 * WG_NGI;
  const size_t lds = (size_t)2 * (CIT * 109 + 64 * 65) * sizeof(float) + (pro_mean ? (size_t)a.pro_nseg * CIT * 16 : 0);
  auto kern = pro_mean ? wino_wgrad_kernel<true, WG_NGI> : wino_wgrad_kernel<false, WG_NGI>;
  {
    static size_t lds_hwm[2] = {0, 0};  // per template instantiation, per prologue variant
    const int rc_lds = sivae_ensure_lds(reinterpret_cast<const void*>(kern), lds, &lds_hwm[pro_mean ? 1 : 0]);
    if (rc_lds != SIVAE_OK) return rc_lds;
  }
  a.nblk = (int)nblk;
  a.xcd_remap = sivae_xcd_remap();
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.xcd_remap ? (nblk + 7) / 8 * 8 : nblk)), dim3(WG_NGI * 256), lds, stream, a);
  rc = sivae_launch_status();
  if (rc != SIVAE_OK) return rc;
  const int n_cic = (Ci + 63) / 64;
  hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3((unsigned)(Co * n_cic)), dim3(256), 0, stream,
                     static_cast<const float*>(workspace), dw, Co, Ci, p.Co_pad, p.Ci_pad, p.n_slices);
  return sivae_launch_status();
}

extern "C" int sivae_conv2d_wino_wgrad(const float* x, const float* dy, float* dw, const float* pro_mean,
                                       const float* pro_invstd, const float* pro_gamma, const float* pro_beta,
                                       float pro_slope, int B, int Ci, int Co, int H, int W, int upsample,
                                       void* workspace, size_t workspace_bytes, hipStream_t stream) {
  return wino_wgrad_impl(x, dy, dw, pro_mean, pro_invstd, pro_gamma, pro_beta, pro_slope, B, Ci, Co, H, W, upsample,
                         workspace, workspace_bytes, stream, 0);
}

// segmented batch (B = nseg * seg_images, pro_mean / pro_invstd [nseg][Ci]): one weight gradient summed over all passes
extern "C" int sivae_conv2d_wino_wgrad_seg(const float* x, const float* dy, float* dw, const float* pro_mean,
                                           const float* pro_invstd, const float* pro_gamma, const float* pro_beta,
                                           float pro_slope, int B, int Ci, int Co, int H, int W, int upsample,
                                           int seg_images, void* workspace, size_t workspace_bytes,
                                           hipStream_t stream) {
  if (seg_images <= 0) return SIVAE_ERR_SHAPE;
  return wino_wgrad_impl(x, dy, dw, pro_mean, pro_invstd, pro_gamma, pro_beta, pro_slope, B, Ci, Co, H, W, upsample,
                         workspace, workspace_bytes, stream, seg_images);
}
