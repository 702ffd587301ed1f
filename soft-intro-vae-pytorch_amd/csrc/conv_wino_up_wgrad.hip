// Weight gradient of "3x3 conv of a nearest-2x-upsampled input" (ResidualBlock.conv1 after nn.Upsample in the
// decoder, soft_intro_vae/train_soft_intro_vae.py:153-156 + :56) in the phase form of conv_wino_up.hip:
//
//   y_pq[i][j] = sum_ab g_pq[a][b] x[i-1+p+a][j-1+q+b]     (x at LOW resolution, y_pq[i][j] = y[2i+p][2j+q])
//   dU_pq[i][j][co][ci] = sum over 2x2 low-res tiles  (A dY_pq A^T)[i][j] * (B^T d_pq B)[i][j]      F(2x2,2x2)
//   dg_pq = G^T dU_pq G   (3x3 -> 2x2),     dW[r][c] = sum_pq dg_pq[a_p(r)][b_q(c)]
//
// 36 multiplies per 4x4 block of dy pixels and (co, ci) instead of 64 for the F(2x2,3x3) weight gradient on the
// upsampled map (conv_wino_wgrad.hip with its upsample flag) and 144 for the direct form.
//
// Block = 8 waves = 4 phases x 2 input-channel groups: wave (ph, cg) owns the 9 frequencies of its phase for
// 32 output x 32 input channels (144 accumulator registers).  MFMA roles as in conv_wino_wgrad.hip: A = A dY A^T
// (row = co, k = tile), B = B^T d B (k = tile, col = ci), a k-step = two horizontally adjacent tiles; both
// operands are built in registers from raw LDS reads (4 dy values of the wave's parity class, a 3x3 patch of x).
// Stages of 16 tiles (4 x 16 low-res pixels: an 8 x 32 block of dy, a 6 x 18 halo of x), double-buffered LDS
// (122 KB, one block per CU), one barrier per stage, next stage's global loads in flight during the MFMA phase.
// Deterministic split over tiles; a slice's partial is G^T dU G already (formed in registers: 16 instead of 36 values per
// (co, ci)); the fixed-order reduce kernel adds the slices and folds the four phase filters back into the 3x3 filter.
#include "common.h"
#include <stdlib.h>

struct WinoUpWgArgs {
  const float* x;   // [B][Ci][Hs][Ws]
  const float* dy;  // [B][Co][2Hs][2Ws]
  float* ws;        // [n_slices][4 phases][2][2][Co_pad][Ci_pad]: transformed partials G^T dU G
  int B, Ci, Co, Hs, Ws;
  int Ci_pad, Co_pad;
  int nrh, nrw, nstages, sps;
  int n_co_tiles, n_ci_tiles;
  int nblk, xcd_remap;
};

#define WUW_COT 32
#define WUW_CIT 64

__global__ void __launch_bounds__(512, 2) wino_up_wgrad_kernel(WinoUpWgArgs a) {
  constexpr int NT = 512;
  constexpr int LWX = 18, NPOSX = 6 * LWX, XP = NPOSX + 1;  // 6 x 18 low-res halo, odd channel stride
  constexpr int YP = 257;                                   // 8 x 32 dy pixels, odd channel stride
  constexpr int XBUF = WUW_CIT * XP, YBUF = WUW_COT * YP;
  constexpr int XQ = 16, YQ = 16;  // loads per thread and stage

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;              // [2][64][XP]
  float* dys = smem + 2 * XBUF;  // [2][32][YP]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int ph = wave & 3, cg = wave >> 2;
  const int pp = ph >> 1, pq = ph & 1;
  const int Hs = a.Hs, Ws = a.Ws, HWs = Hs * Ws, H = 2 * Hs, W = 2 * Ws, HW = H * W;

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
  const int ci0 = (tile % a.n_ci_tiles) * WUW_CIT, co0 = (tile / a.n_ci_tiles) * WUW_COT;
  const int s_begin = slice * a.sps;
  const int s_end = (s_begin + a.sps < a.nstages) ? (s_begin + a.sps) : a.nstages;

  // ---- staging maps.  x: 128 halo slots per channel row (108 used, the rest duplicate the first), 4 channel
  // phases; dy: 256 pixels per channel row, 2 channel phases
  const int xsub = __builtin_amdgcn_readfirstlane(tid >> 7);
  const int xpos = (tid & 127) < NPOSX ? (tid & 127) : (tid & 127) - NPOSX;
  const int xrr = xpos / LWX, xcc = xpos % LWX;
  const int ysub = __builtin_amdgcn_readfirstlane(tid >> 8);
  const int ypix = tid & 255, ypy = ypix >> 5, ypx = ypix & 31;

  // ---- operand bases.  k-step kk = tile pair (ty = kk >> 2, tx = 2*(kk & 3) + hh)
  //   x patch rows 2*ty + pp + r, cols 2*tx + pq + c;  dy pixels (4*ty + 2a + pp, 4*tx + 2b + pq)
  const int base_x = (cg * 32 + l31) * XP + 2 * hh + pp * LWX + pq;
  const int base_y = l31 * YP + 4 * hh + pp * 32 + pq;

  f32x16 acc[9];
#pragma unroll
  for (int f = 0; f < 9; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

  float xr[XQ], yr[YQ];

#define WUW_LOAD(S)                                                                                 \
  {                                                                                                 \
    const int s_ = (S);                                                                             \
    const int b_ = s_ / (a.nrh * a.nrw);                                                            \
    const int rem_ = s_ - b_ * (a.nrh * a.nrw);                                                     \
    const int ry_ = rem_ / a.nrw, rx_ = rem_ - ry_ * a.nrw;                                         \
    const int r0_ = ry_ * 4, c0_ = rx_ * 16; /* low-res origin of the region */                     \
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(a.x + (size_t)b_ * a.Ci * HWs, (unsigned long long)a.Ci * HWs * 4ull); \
    const __amdgpu_buffer_rsrc_t yrs = make_rsrc(a.dy + (size_t)b_ * a.Co * HW, (unsigned long long)a.Co * HW * 4ull);   \
    unsigned xoff = SIVAE_OOB, yoff = SIVAE_OOB;                                                    \
    {                                                                                               \
      const int r = r0_ + xrr - 1, c = c0_ + xcc - 1;                                               \
      if (r >= 0 && r < Hs && c >= 0 && c < Ws) xoff = (unsigned)(r * Ws + c) * 4u;                 \
    }                                                                                               \
    {                                                                                               \
      const int r = 2 * r0_ + ypy, c = 2 * c0_ + ypx;                                               \
      if (r < H && c < W) yoff = (unsigned)(r * W + c) * 4u;                                        \
    }                                                                                               \
    _Pragma("unroll") for (int q = 0; q < XQ; ++q) {                                                \
      const int ci = ci0 + xsub + 4 * q;                                                            \
      const int cic = ci < a.Ci ? ci : a.Ci - 1;                                                    \
      xr[q] = buf_load_f32(xrs, xoff, (unsigned)cic * (unsigned)HWs * 4u);                          \
    }                                                                                               \
    _Pragma("unroll") for (int q = 0; q < YQ; ++q) {                                                \
      const int co = co0 + ysub + 2 * q;                                                            \
      const int coc = co < a.Co ? co : a.Co - 1;                                                    \
      yr[q] = buf_load_f32(yrs, yoff, (unsigned)coc * (unsigned)HW * 4u);                           \
    }                                                                                               \
  }
#define WUW_STORE(BUF)                                                                              \
  {                                                                                                 \
    _Pragma("unroll") for (int q = 0; q < XQ; ++q) xs[(BUF)*XBUF + (xsub + 4 * q) * XP + xpos] = xr[q]; \
    _Pragma("unroll") for (int q = 0; q < YQ; ++q) dys[(BUF)*YBUF + (ysub + 2 * q) * YP + ypix] = yr[q]; \
  }
#define WUW_READ(BUF, KK, D, DY)                                                                    \
  {                                                                                                 \
    const float* px_ = xs + (BUF)*XBUF + base_x + 2 * ((KK) >> 2) * LWX + 4 * ((KK)&3);            \
    _Pragma("unroll") for (int r = 0; r < 3; ++r) {                                                 \
      D[r][0] = px_[r * LWX + 0];                                                                   \
      D[r][1] = px_[r * LWX + 1];                                                                   \
      D[r][2] = px_[r * LWX + 2];                                                                   \
    }                                                                                               \
    const float* py_ = dys + (BUF)*YBUF + base_y + 4 * ((KK) >> 2) * 32 + 8 * ((KK)&3);            \
    DY[0] = py_[0];                                                                                 \
    DY[1] = py_[2];                                                                                 \
    DY[2] = py_[64];                                                                                \
    DY[3] = py_[66];                                                                                \
  }
#define WUW_STEP(D, DY)                                                                             \
  {                                                                                                 \
    float t[3][3];                                                                                  \
    _Pragma("unroll") for (int r = 0; r < 3; ++r) {                                                 \
      t[r][0] = D[r][0] - D[r][1];                                                                  \
      t[r][1] = D[r][1];                                                                            \
      t[r][2] = D[r][1] - D[r][2];                                                                  \
    }                                                                                               \
    /* Mg = A dY A^T, A = [[1,0],[1,1],[0,-1]]: columns (d0, d0 + d1, -d1), then the same over rows */ \
    const float m00 = DY[0], m01 = DY[0] + DY[1], m02 = -DY[1];                                     \
    const float n0 = DY[2], n1 = DY[2] + DY[3], n2 = -DY[3];                                        \
    const float mg[3][3] = {{m00, m01, m02}, {m00 + n0, m01 + n1, m02 + n2}, {-n0, -n1, -n2}};      \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                 \
      const float v0 = t[0][j] - t[1][j], v1 = t[1][j], v2 = t[1][j] - t[2][j];                     \
      acc[0 * 3 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(mg[0][j], v0, acc[0 * 3 + j], 0, 0, 0); \
      acc[1 * 3 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(mg[1][j], v1, acc[1 * 3 + j], 0, 0, 0); \
      acc[2 * 3 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(mg[2][j], v2, acc[2 * 3 + j], 0, 0, 0); \
    }                                                                                               \
  }
#define WUW_KSTEP(BUF, KK, D, DY, DN, DYN)                                                          \
  {                                                                                                 \
    if ((KK) + 1 < 8) WUW_READ(BUF, (KK) + 1, DN, DYN)                                              \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    WUW_STEP(D, DY)                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                              \
  }
#define WUW_MMA(S, BUF)                                                                             \
  {                                                                                                 \
    float d0[3][3], d1[3][3], e0[4], e1[4];                                                         \
    const bool next_ = (S) + 1 < s_end;                                                             \
    if (next_) WUW_LOAD((S) + 1)                                                                    \
    WUW_READ(BUF, 0, d0, e0)                                                                        \
    WUW_KSTEP(BUF, 0, d0, e0, d1, e1)                                                               \
    WUW_KSTEP(BUF, 1, d1, e1, d0, e0)                                                               \
    WUW_KSTEP(BUF, 2, d0, e0, d1, e1)                                                               \
    WUW_KSTEP(BUF, 3, d1, e1, d0, e0)                                                               \
    WUW_KSTEP(BUF, 4, d0, e0, d1, e1)                                                               \
    WUW_KSTEP(BUF, 5, d1, e1, d0, e0)                                                               \
    WUW_KSTEP(BUF, 6, d0, e0, d1, e1)                                                               \
    if (next_) WUW_STORE((BUF) ^ 1)                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    WUW_KSTEP(BUF, 7, d1, e1, d0, e0)                                                               \
    __syncthreads();                                                                                \
  }

  if (s_begin < s_end) {
    WUW_LOAD(s_begin)
    WUW_STORE(0)
    __syncthreads();
    int s = s_begin;
    for (; s + 1 < s_end; s += 2) {
      WUW_MMA(s, 0)
      WUW_MMA(s + 1, 1)
    }
    if (s < s_end) WUW_MMA(s, 0)
  }
#undef WUW_LOAD
#undef WUW_STORE
#undef WUW_READ
#undef WUW_STEP
#undef WUW_KSTEP
#undef WUW_MMA

  // ---- partial of this slice.  acc[f][r] = dU of frequency f = i*3 + j of phase ph, co = co0 + row(r, hh), ci = ci0 +
  // cg*32 + l31: the wave holds all nine frequencies of its phase, so dg = G^T dU G (3x3 -> 2x2, G = [[1,0],[1,1],[0,1]],
  // all +1: 8 additions) is formed here in registers — linear, so it commutes with the sum over slices — and the partials
  // and everything the reducer reads shrink from 36 to 16 values per (co, ci): plane ph*4 + a*2 + b.
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float t00 = acc[0][r] + acc[3][r], t01 = acc[1][r] + acc[4][r], t02 = acc[2][r] + acc[5][r];
    const float t10 = acc[3][r] + acc[6][r], t11 = acc[4][r] + acc[7][r], t12 = acc[5][r] + acc[8][r];
    acc[0][r] = t00 + t01;
    acc[1][r] = t01 + t02;
    acc[2][r] = t10 + t11;
    acc[3][r] = t11 + t12;
  }
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    float* base = a.ws + ((size_t)(slice * 16 + ph * 4 + f) * a.Co_pad + co0) * a.Ci_pad + ci0 + cg * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
      base[(size_t)row * a.Ci_pad] = acc[f][r];
    }
  }
}

// dW[co][ci][3][3] from the slice partials dg_pq[a][b] ([n_slices][4 phases][2][2], transformed in the kernel above):
// dW[r][c] = sum_pq dg_pq[a_p(r)][b_q(c)],  a_0 = (0,1,1), a_1 = (0,0,1).  Block = one co x 64 ci x 4 slice phases.
__global__ void __launch_bounds__(256) wino_up_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                                   int Co, int Ci, int Co_pad, int Ci_pad,
                                                                   int n_slices) {
  __shared__ float red[3][16][64];
  const int cil = threadIdx.x & 63, sp = threadIdx.x >> 6;
  const int n_cic = (Ci + 63) / 64;
  const int co = blockIdx.x / n_cic, ci = (blockIdx.x % n_cic) * 64 + cil;
  float u[16];
#pragma unroll
  for (int f = 0; f < 16; ++f) u[f] = 0.f;
  if (ci < Ci) {
    for (int s = sp; s < n_slices; s += 4) {
      const float* p = ws + ((size_t)(s * 16) * Co_pad + co) * Ci_pad + ci;
#pragma unroll
      for (int f = 0; f < 16; ++f) u[f] += p[(size_t)f * Co_pad * Ci_pad];
    }
  }
  if (sp > 0) {
#pragma unroll
    for (int f = 0; f < 16; ++f) red[sp - 1][f][cil] = u[f];
  }
  __syncthreads();
  if (sp == 0 && ci < Ci) {
#pragma unroll
    for (int f = 0; f < 16; ++f) u[f] = ((u[f] + red[0][f][cil]) + red[1][f][cil]) + red[2][f][cil];
    float dwv[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) dwv[r][c] = 0.f;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float* g = u + (p * 2 + q) * 4;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const int ar = p == 0 ? (r == 0 ? 0 : 1) : (r == 2 ? 1 : 0);
            const int bc = q == 0 ? (c == 0 ? 0 : 1) : (c == 2 ? 1 : 0);
            dwv[r][c] += g[ar * 2 + bc];
          }
      }
    float* dst = dw + ((size_t)co * Ci + ci) * 9;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) dst[r * 3 + c] = dwv[r][c];
  }
}

namespace {
struct WuwPlan {
  int Ci_pad, Co_pad, n_ci_tiles, n_co_tiles, nrh, nrw, nstages, sps, n_slices;
};

int wuw_plan(int B, int Ci, int Co, int Hs, int Ws, WuwPlan* p) {
  p->n_ci_tiles = cdiv(Ci, WUW_CIT);
  p->n_co_tiles = cdiv(Co, WUW_COT);
  p->Ci_pad = p->n_ci_tiles * WUW_CIT;
  p->Co_pad = p->n_co_tiles * WUW_COT;
  p->nrh = cdiv(Hs, 4);
  p->nrw = cdiv(Ws, 16);
  const long long ns = (long long)B * p->nrh * p->nrw;
  if (ns > 0x3fffffffLL) return SIVAE_ERR_RANGE;
  p->nstages = (int)ns;
  const int ntiles = p->n_ci_tiles * p->n_co_tiles;
  // one 8-wave block per CU: two rounds of 256 blocks, at least 16 stages per slice
  int n_slices = cdiv(512, ntiles);
  const int max_slices = p->nstages / 16 > 0 ? p->nstages / 16 : 1;
  if (n_slices > max_slices) n_slices = max_slices;
  p->sps = cdiv(p->nstages, n_slices);
  p->n_slices = cdiv(p->nstages, p->sps);
  return SIVAE_OK;
}
}  // namespace

// Hs, Ws = low-resolution (x) size
extern "C" int sivae_conv2d_wino_up_wgrad_supported(int Hs, int Ws) { return (Hs >= 4 && Ws >= 16) ? 1 : 0; }

extern "C" size_t sivae_conv2d_wino_up_wgrad_workspace_bytes(int B, int Ci, int Co, int Hs, int Ws) {
  WuwPlan p;
  if (B <= 0 || Ci <= 0 || Co <= 0 || !sivae_conv2d_wino_up_wgrad_supported(Hs, Ws)) return 0;
  if (wuw_plan(B, Ci, Co, Hs, Ws, &p) != SIVAE_OK) return 0;
  return (size_t)p.n_slices * 16 * p.Co_pad * p.Ci_pad * sizeof(float);
}

extern "C" int sivae_conv2d_wino_up_wgrad(const float* x_half, const float* dy, float* dw, int B, int Ci, int Co,
                                          int Hs, int Ws, void* workspace, size_t workspace_bytes,
                                          hipStream_t stream) {
  if (!x_half || !dy || !dw || !workspace) return SIVAE_ERR_NULL;
  if (B <= 0 || Ci <= 0 || Co <= 0 || Hs <= 0 || Ws <= 0) return SIVAE_ERR_SHAPE;
  if (!sivae_conv2d_wino_up_wgrad_supported(Hs, Ws)) return SIVAE_ERR_SHAPE;
  const long long hw = 4ll * Hs * Ws;
  if ((long long)Ci * (hw / 4) * 4 >= 0x7fffffffLL || (long long)Co * hw * 4 >= 0x7fffffffLL) return SIVAE_ERR_RANGE;
  WuwPlan p;
  int rc = wuw_plan(B, Ci, Co, Hs, Ws, &p);
  if (rc != SIVAE_OK) return rc;
  const size_t need = (size_t)p.n_slices * 16 * p.Co_pad * p.Ci_pad * sizeof(float);
  if (workspace_bytes < need) return SIVAE_ERR_WORKSPACE;
  WinoUpWgArgs a;
  a.x = x_half;
  a.dy = dy;
  a.ws = static_cast<float*>(workspace);
  a.B = B;
  a.Ci = Ci;
  a.Co = Co;
  a.Hs = Hs;
  a.Ws = Ws;
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
  const size_t lds = (size_t)2 * (WUW_CIT * 109 + WUW_COT * 257) * sizeof(float);
  {
    static size_t lds_hwm = 0;
    const int rc_lds = sivae_ensure_lds(reinterpret_cast<const void*>(wino_up_wgrad_kernel), lds, &lds_hwm);
    if (rc_lds != SIVAE_OK) return rc_lds;
  }
  a.nblk = (int)nblk;
  a.xcd_remap = sivae_xcd_remap();
  hipLaunchKernelGGL(wino_up_wgrad_kernel, dim3((unsigned)(a.xcd_remap ? (nblk + 7) / 8 * 8 : nblk)), dim3(512), lds, stream, a);
  rc = sivae_launch_status();
  if (rc != SIVAE_OK) return rc;
  const int n_cic = (Ci + 63) / 64;
  hipLaunchKernelGGL(wino_up_wgrad_reduce_kernel, dim3((unsigned)(Co * n_cic)), dim3(256), 0, stream,
                     static_cast<const float*>(workspace), dw, Co, Ci, p.Co_pad, p.Ci_pad, p.n_slices);
  return sivae_launch_status();
}
