// 3x3 convolution of a nearest-2x-upsampled input (nn.Upsample(2,'nearest') -> ResidualBlock.conv1 in the
// decoder, soft_intro_vae/train_soft_intro_vae.py:153-156 with :56), computed on the LOW-resolution tensor.
//
// With xu[u][v] = x[u>>1][v>>1], every output parity class ("phase" (p, q) = (row & 1, col & 1)) is a 2x2
// convolution of x with its own summed filter
//     y[2i+p][2j+q] = sum_{a,b in {0,1}} g_pq[a][b] * x[i-1+p+a][j-1+q+b],
//     g_pq[a][b] = sum_{r in R_p(a), c in R_q(b)} w[r][c],   R_0 = ({0}, {1,2}),  R_1 = ({0,1}, {2})
// and each phase runs as Winograd F(2x2, 2x2): 9 multiplies per 2x2 outputs of the phase (all-(+-1) transforms
//     B^T = [[1,-1,0],[0,1,0],[0,1,-1]],  G = [[1,0],[1,1],[0,1]],  A^T = [[1,1,0],[0,1,-1]]).
// Per 4x4 block of output pixels that is 4 x 9 = 36 multiplies, against 64 for F(2x2,3x3) on the upsampled map
// (conv_wino.hip) and 144 for the direct form: 1.78x fewer MFMA passes than the kernel it replaces.
//
// Work split: a block is 4 waves = the 4 phases; wave (p, q) owns all 9 frequencies of its phase for 32 output
// channels x 32 low-resolution tiles (2x2 low-res = 4x4 output pixels each): 144 accumulator registers.  The four
// waves share one zero-padded low-resolution halo tile in LDS (each reads its own shifted 3x3 patch: 9
// ds_read_b32 + 12 VALU ops per k-step for 9 MFMAs); U_pq = G g_pq G^T comes straight from L2 as three 16-byte
// loads per k-step (packed [phase][ci][co][12]).  All 9 frequencies of an output live in ONE wave, so the output
// transform is entirely in registers — no LDS exchange, no barrier — and the wave stores its phase's pixels.
// K loop, halo staging, persistent work items and fused BatchNorm+LeakyReLU prologue are those of conv_wino.hip.
#include "common.h"
#include "pack_batch.h"
#include <stdlib.h>

struct WinoUpArgs {
  const float* x;   // [B][Ci][H/2][W/2]
  const float* up;  // packed U [4 phases][Ci_pad][Co_pad][12]
  float* y;         // [B][Co][H][W]
  const float* pro_mean;
  const float* pro_invstd;
  const float* pro_gamma;
  const float* pro_beta;
  float pro_slope;
  float* stats;  // [4 * n_px_tiles][Co][2] or null
  int B, Ci, Co, H, W;
  int Ci_pad, Co_pad;
  int nbh, nbw;
  int n_co_tiles;
  int n_items;
  int xcd_group;  // 1: the blocks of an XCD start on contiguous items (conv_wino.hip)
};

#define WUP_CK 16
#define WUP_TCO 32

template <int TTH_L2, int TTW_L2, bool PRO>
__global__ void __launch_bounds__(256, 2) conv_wino_up_kernel(WinoUpArgs a) {
  constexpr int TTH = 1 << TTH_L2, TTW = 1 << TTW_L2;
  static_assert(TTH * TTW == 32, "a block is 32 low-resolution tiles");
  constexpr int NT = 256;
  constexpr int PXH = 2 * TTH, PXW = 2 * TTW;  // low-resolution pixels per block
  constexpr int LH = PXH + 2, LWU = PXW + 2;
  constexpr int PH = TTW + TTW / 4, RS = 2 * PH, PLANE = LH * RS;  // same conflict-free halo layout as conv_wino
  constexpr int NPOS = LH * LWU;
  static_assert(NPOS <= NT, "one halo position per thread");
  constexpr int CK = WUP_CK;
  constexpr int XBUF = CK * PLANE;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;  // [2][CK][PLANE]
  float4* pro4 = reinterpret_cast<float4*>(smem + 2 * XBUF);
  float* ex = smem + 2 * XBUF + (PRO ? 4 * a.Ci_pad : 0);  // [4 waves][16 slots][2][64 lanes]: the epilogue's exchange

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int pp = wave >> 1, pq = wave & 1;  // this wave's phase
  const int H = a.H, W = a.W;
  const int Hs = H >> 1, Ws = W >> 1, HWs = Hs * Ws;

  const int n_items = a.n_items;
  // Consecutive blockIdx go round-robin to the 8 XCDs (each with its own L2).  Items are ordered output-channel tile
  // fastest, then tile column, tile row: with xcd_group XCD x starts on the contiguous items x * grid/8 + slot, so the
  // readers of one halo — the channel-tile siblings of a tile block and its row / column neighbours — meet in one L2
  // (round 4, FETCH_SIZE: 4.0 GB read per launch on average against 0.63 GB of inputs with the plain order)
  int item = blockIdx.x;
  if (a.xcd_group) item = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
  int pt, b, r0, c0, co0;  // r0, c0: low-resolution origin of the tile block
  __amdgpu_buffer_rsrc_t xrsrc;
  const __amdgpu_buffer_rsrc_t ursrc = make_rsrc(a.up, 4ull * a.Ci_pad * a.Co_pad * 48ull);
  unsigned xo, ua_base;
  static_assert(2 * NPOS >= NT, "duplicate-owner mapping");
  const int teff = tid < NPOS ? tid : tid - NPOS;  // (threads beyond the halo duplicate the first slots: no predication)
  const int xrr = teff / LWU, xcc = teff % LWU;
  const int xl = xrr * RS + (xcc & 1) * PH + (xcc >> 1);
  float xmask = 0.f;
#define WUP_SETUP(ITEM)                                                  \
  {                                                                      \
    const int co_tile = (ITEM) % a.n_co_tiles;                           \
    pt = (ITEM) / a.n_co_tiles;                                          \
    const int tbx = pt % a.nbw;                                          \
    const int t2 = pt / a.nbw;                                           \
    const int tby = t2 % a.nbh;                                          \
    b = t2 / a.nbh;                                                      \
    r0 = tby * PXH;                                                      \
    c0 = tbx * PXW;                                                      \
    co0 = co_tile * WUP_TCO;                                             \
    xrsrc = make_rsrc(a.x + (size_t)b * a.Ci * HWs, (unsigned long long)a.Ci * HWs * 4ull); \
    const int r = r0 + xrr - 1, c = c0 + xcc - 1;                        \
    xo = SIVAE_OOB;                                                      \
    xmask = 0.f;                                                         \
    if (r >= 0 && r < Hs && c >= 0 && c < Ws) {                          \
      xo = (unsigned)(r * Ws + c) * 4u;                                  \
      xmask = 1.f;                                                       \
    }                                                                    \
    ua_base = (unsigned)((wave * a.Ci_pad) * a.Co_pad + co0) * 48u;      \
  }

  // A operand: lane -> (ci = k-step*2 + hh, co = co0 + l31): 12 floats (9 used) = three 16-byte loads
  const unsigned va0 = (unsigned)(hh * a.Co_pad + l31) * 48u;
  const unsigned ua_step = (unsigned)a.Co_pad * 48u;  // bytes per input channel

  // B operand: tile (ty, tx); patch rows (2*ty + pp + r), columns (2*tx + pq + c), r, c = 0..2, in the
  // [row][col parity][col/2] halo layout
  const int tx = l31 & (TTW - 1), ty = l31 >> TTW_L2;
  const int bb = hh * PLANE + (2 * ty + pp) * RS + tx;
  const int oc0 = ((pq + 0) & 1) * PH + ((pq + 0) >> 1);
  const int oc1 = ((pq + 1) & 1) * PH + ((pq + 1) >> 1);
  const int oc2 = ((pq + 2) & 1) * PH + ((pq + 2) >> 1);

  f32x16 acc[9];
  float xr[CK];
  float4 AR[4][3];  // U operand ring: slot (k-step & 3), refilled 4 k-steps ahead
  const int nksteps = a.Ci_pad / 2;

#define WUP_LOAD_X(CH)                                                   \
  {                                                                      \
    _Pragma("unroll") for (int ck = 0; ck < CK; ++ck) {                  \
      const int ci = (CH)*CK + ck;                                       \
      const int cic = ci < a.Ci ? ci : a.Ci - 1;                         \
      xr[ck] = buf_load_f32(xrsrc, xo, (unsigned)cic * (unsigned)HWs * 4u); \
    }                                                                    \
  }
#define WUP_LOAD_A(KS_ABS, SLOT)                                         \
  {                                                                      \
    const unsigned so = ua_base + (unsigned)(2 * (KS_ABS)) * ua_step;    \
    AR[SLOT][0] = buf_load_f32x4(ursrc, va0, so);                        \
    AR[SLOT][1] = buf_load_f32x4(ursrc, va0 + 16u, so);                  \
    AR[SLOT][2] = buf_load_f32x4(ursrc, va0 + 32u, so);                  \
  }
#define WUP_STORE_X(CH, BUF)                                             \
  {                                                                      \
    _Pragma("unroll") for (int ck = 0; ck < CK; ++ck) {                  \
      const int ci = (CH)*CK + ck;                                       \
      float v = xr[ck];                                                  \
      if (PRO) {                                                         \
        const float4 p4 = pro4[ci];                                      \
        v = lrelu01((v - p4.x) * p4.y + p4.z, a.pro_slope) * xmask;      \
      } else {                                                           \
        v = ci < a.Ci ? v : 0.f;                                         \
      }                                                                  \
      xs[(BUF)*XBUF + ck * PLANE + xl] = v;                              \
    }                                                                    \
  }
#define WUP_READ(BUF, KK, D)                                             \
  {                                                                      \
    const float* pb_ = xs + (BUF)*XBUF + 2 * (KK)*PLANE + bb;            \
    _Pragma("unroll") for (int r = 0; r < 3; ++r) {                      \
      D[r][0] = pb_[r * RS + oc0];                                       \
      D[r][1] = pb_[r * RS + oc1];                                       \
      D[r][2] = pb_[r * RS + oc2];                                       \
    }                                                                    \
  }
#define WUP_STEP(SLOT, D)                                                \
  {                                                                      \
    float t[3][3];                                                       \
    _Pragma("unroll") for (int r = 0; r < 3; ++r) {                      \
      t[r][0] = D[r][0] - D[r][1];                                       \
      t[r][1] = D[r][1];                                                 \
      t[r][2] = D[r][1] - D[r][2];                                       \
    }                                                                    \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) {                      \
      const float v0 = t[0][j] - t[1][j], v1 = t[1][j], v2 = t[1][j] - t[2][j]; \
      const float u0 = (j == 0) ? AR[SLOT][0].x : ((j == 1) ? AR[SLOT][0].y : AR[SLOT][0].z); \
      const float u1 = (j == 0) ? AR[SLOT][0].w : ((j == 1) ? AR[SLOT][1].x : AR[SLOT][1].y); \
      const float u2 = (j == 0) ? AR[SLOT][1].z : ((j == 1) ? AR[SLOT][1].w : AR[SLOT][2].x); \
      acc[0 * 3 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0, v0, acc[0 * 3 + j], 0, 0, 0); \
      acc[1 * 3 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1, v1, acc[1 * 3 + j], 0, 0, 0); \
      acc[2 * 3 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(u2, v2, acc[2 * 3 + j], 0, 0, 0); \
    }                                                                    \
  }
#define WUP_KSTEP(CH, BUF, KK, D, DN)                                    \
  {                                                                      \
    if ((KK) + 1 < CK / 2) WUP_READ(BUF, (KK) + 1, DN)                   \
    __builtin_amdgcn_sched_barrier(0);                                   \
    WUP_STEP((KK)&3, D)                                                  \
    __builtin_amdgcn_sched_barrier(0);                                   \
    if ((CH) * (CK / 2) + (KK) + 4 < nksteps) WUP_LOAD_A((CH) * (CK / 2) + (KK) + 4, (KK)&3) \
  }
#define WUP_MMA(CH, BUF, NEXT)                                           \
  {                                                                      \
    float d0[3][3], d1[3][3];                                            \
    if (NEXT) WUP_LOAD_X((CH) + 1)                                       \
    WUP_READ(BUF, 0, d0)                                                 \
    WUP_KSTEP(CH, BUF, 0, d0, d1)                                        \
    WUP_KSTEP(CH, BUF, 1, d1, d0)                                        \
    WUP_KSTEP(CH, BUF, 2, d0, d1)                                        \
    WUP_KSTEP(CH, BUF, 3, d1, d0)                                        \
    WUP_KSTEP(CH, BUF, 4, d0, d1)                                        \
    WUP_KSTEP(CH, BUF, 5, d1, d0)                                        \
    WUP_KSTEP(CH, BUF, 6, d0, d1)                                        \
    if (NEXT) WUP_STORE_X((CH) + 1, (BUF) ^ 1)                           \
    __builtin_amdgcn_sched_barrier(0);                                   \
    WUP_KSTEP(CH, BUF, 7, d1, d0)                                        \
    __syncthreads();                                                     \
  }

  const int nchunks = a.Ci_pad / CK;
  if (PRO) {
    for (int c = tid; c < a.Ci_pad; c += NT) {
      pro4[c] = c < a.Ci ? make_float4(a.pro_mean[c], a.pro_invstd[c] * a.pro_gamma[c], a.pro_beta[c], 0.f)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
  }
  WUP_SETUP(item)
  WUP_LOAD_X(0)
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) WUP_LOAD_A(kk, kk)
  for (;;) {
#pragma unroll
    for (int f = 0; f < 9; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
    WUP_STORE_X(0, 0)
    __syncthreads();
    int ch = 0;
    for (; ch + 1 < nchunks; ch += 2) {
      WUP_MMA(ch, 0, true)
      const bool more = ch + 2 < nchunks;
      WUP_MMA(ch + 1, 1, more)
    }
    if (ch < nchunks) WUP_MMA(ch, 0, false)

    __builtin_amdgcn_s_setprio(1);  // serial tail at raised priority (see conv_wino.hip)
    const int e_pt = pt, e_b = b, e_r0 = r0, e_c0 = c0, e_co0 = co0;
    const int next = item + (int)gridDim.x;
    const bool has_next = next < n_items;
    if (has_next) {
      WUP_SETUP(next)
      WUP_LOAD_X(0)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) WUP_LOAD_A(kk, kk)
    }
    // ---- output transform in this wave's registers: acc[i*3+j][r] -> Y[2][2] of tile l31, channel
    // e_co0 + (r&3) + 8*(r>>2) + 4*hh; the phase's pixels are (2*(e_r0 + 2*ty + a) + pp, 2*(e_c0 + 2*tx + b) + pq).
    // A phase holds every OTHER pixel of its rows: stored from here they would be 4-byte writes at an 8-byte stride —
    // partial lines that L2 completes by FETCHING them (round 4, FETCH_SIZE: this kernel read ~1.3 bytes of y per byte it
    // wrote).  So the two column phases of a row phase swap halves through LDS: wave (pp, 0) stores row a = 0 of every
    // tile, wave (pp, 1) row a = 1, each as ONE 16-byte store of four consecutive pixels per lane and channel slot.
    {
      const __amdgpu_buffer_rsrc_t yrsrc =
          make_rsrc(a.y + (size_t)e_b * a.Co * H * W, (unsigned long long)a.Co * H * W * 4ull);
      const int li = e_r0 + 2 * ty, lj = e_c0 + 2 * tx;  // low-resolution coordinates of the tile
      const bool ok00 = li < Hs && lj < Ws, ok01 = li < Hs && lj + 1 < Ws;
      const bool ok10 = li + 1 < Hs && lj < Ws, ok11 = li + 1 < Hs && lj + 1 < Ws;
      float* exw = ex + wave * (16 * 2 * 64) + lane;
      const float* exr = ex + (wave ^ 1) * (16 * 2 * 64) + lane;
      float keep[16][2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int chn = e_co0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const float s00 = acc[0][r] + acc[3][r], s01 = acc[1][r] + acc[4][r], s02 = acc[2][r] + acc[5][r];
        const float s10 = acc[3][r] - acc[6][r], s11 = acc[4][r] - acc[7][r], s12 = acc[5][r] - acc[8][r];
        const float y00 = s00 + s01, y01 = s01 - s02, y10 = s10 + s11, y11 = s11 - s12;
        if (pq == 0) {  // (wave-uniform)
          keep[r][0] = y00; keep[r][1] = y01;
          exw[(r * 2 + 0) * 64] = y10; exw[(r * 2 + 1) * 64] = y11;
        } else {
          keep[r][0] = y10; keep[r][1] = y11;
          exw[(r * 2 + 0) * 64] = y00; exw[(r * 2 + 1) * 64] = y01;
        }
        if (a.stats != nullptr) {
          const bool cok = chn < a.Co;
          float s = (ok00 ? y00 : 0.f) + (ok01 ? y01 : 0.f) + (ok10 ? y10 : 0.f) + (ok11 ? y11 : 0.f);
          float q = (ok00 ? y00 * y00 : 0.f) + (ok01 ? y01 * y01 : 0.f) + (ok10 ? y10 * y10 : 0.f) +
                    (ok11 ? y11 * y11 : 0.f);
          s = half_wave_sum_hi(s);
          q = half_wave_sum_hi(q);
          if (l31 == 31 && cok) {
            float* dst = a.stats + ((size_t)(e_pt * 4 + wave) * a.Co + chn) * 2;
            dst[0] = s;
            dst[1] = q;
          }
        }
      }
      // (LDS-only barrier: __syncthreads() would also wait out the next item's operand loads requested above)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      // row a = pq of the tile: output row 2*(li + pq) + pp, columns 2*lj .. 2*lj + 3 = (b, q) = (0,0) (0,1) (1,0) (1,1)
      const bool rok = pq == 0 ? li < Hs : li + 1 < Hs;
      const unsigned base = (unsigned)((2 * (li + pq) + pp) * W + 2 * lj) * 4u;
      f32x4 f_prev = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int chn = e_co0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const float o0 = exr[(r * 2 + 0) * 64], o1 = exr[(r * 2 + 1) * 64];  // the other column phase, b = 0, 1
        const unsigned cb = base + (unsigned)chn * (unsigned)(H * W) * 4u;
        const float c0_ = pq == 0 ? keep[r][0] : o0, c1_ = pq == 0 ? o0 : keep[r][0];
        const float c2_ = pq == 0 ? keep[r][1] : o1, c3_ = pq == 0 ? o1 : keep[r][1];
        if (rok && lj < Ws && chn < a.Co) {  // (W % 4 == 0: a tile's two low-resolution columns are inside together)
          typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
          const f32x4 f = {c0_, c1_, c2_, c3_};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f), yrsrc, (int)cb, 0, 0);
          SIVAE_PIN4(f_prev)  // (the previous store's data registers stay untouched until this store is issued)
          f_prev = f;
        }
      }
      SIVAE_PIN4(f_prev)
    }
    __builtin_amdgcn_s_setprio(0);
    if (!has_next) break;
    item = next;
  }
#undef WUP_SETUP
#undef WUP_LOAD_X
#undef WUP_LOAD_A
#undef WUP_STORE_X
#undef WUP_READ
#undef WUP_STEP
#undef WUP_KSTEP
#undef WUP_MMA
}

// ---- filter transform: g_pq = P_p w Q_q^T (3x3 -> 2x2 per phase), U_pq = G g_pq G^T (2x2 -> 3x3),
// packed [phase][ci_pad][co_pad][12] (row-major 3x3 in the first 9 floats), padding entries zero
__device__ __forceinline__ void pack_wino_up_body(const float* __restrict__ w, float* __restrict__ up, int Co,
                                                           int Ci, int kpad, int npad, size_t idx0_, const size_t stride_) {
  const size_t total = (size_t)kpad * npad;
  for (size_t idx = idx0_; idx < total; idx += stride_) {
    const int n = (int)(idx % npad), k = (int)(idx / npad);
    float g[3][3];
    const bool ok = k < Ci && n < Co;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) g[r][c] = ok ? w[((size_t)n * Ci + k) * 9 + r * 3 + c] : 0.f;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        // rows: p = 0 -> (w0, w1 + w2), p = 1 -> (w0 + w1, w2); same for columns with q
        float rw[2][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          rw[0][c] = p == 0 ? g[0][c] : g[0][c] + g[1][c];
          rw[1][c] = p == 0 ? g[1][c] + g[2][c] : g[2][c];
        }
        float gp[2][2];
#pragma unroll
        for (int a_ = 0; a_ < 2; ++a_) {
          gp[a_][0] = q == 0 ? rw[a_][0] : rw[a_][0] + rw[a_][1];
          gp[a_][1] = q == 0 ? rw[a_][1] + rw[a_][2] : rw[a_][2];
        }
        // U = G gp G^T, G = [[1,0],[1,1],[0,1]]
        float gr[3][2];
#pragma unroll
        for (int b_ = 0; b_ < 2; ++b_) {
          gr[0][b_] = gp[0][b_];
          gr[1][b_] = gp[0][b_] + gp[1][b_];
          gr[2][b_] = gp[1][b_];
        }
        float u[12];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          u[i * 3 + 0] = gr[i][0];
          u[i * 3 + 1] = gr[i][0] + gr[i][1];
          u[i * 3 + 2] = gr[i][1];
        }
        u[9] = u[10] = u[11] = 0.f;
        float4* dst = reinterpret_cast<float4*>(up + (((size_t)(p * 2 + q) * kpad + k) * npad + n) * 12);
        dst[0] = make_float4(u[0], u[1], u[2], u[3]);
        dst[1] = make_float4(u[4], u[5], u[6], u[7]);
        dst[2] = make_float4(u[8], u[9], u[10], u[11]);
      }
  }
}

__global__ void __launch_bounds__(256) pack_wino_up_kernel(const float* __restrict__ w, float* __restrict__ up, int Co,
                                                           int Ci, int kpad, int npad) {
  pack_wino_up_body(w, up, Co, Ci, kpad, npad, (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)gridDim.x * 256);
}

__global__ void __launch_bounds__(256) pack_wino_up_batch_kernel(const SivaePackJob* __restrict__ jobs,
                                                                  const unsigned short* __restrict__ block_job) {
  const SivaePackJob j = jobs[block_job[blockIdx.x]];
  pack_wino_up_body(j.w, j.dst, j.Co, j.Ci, j.kpad, j.npad, (size_t)(blockIdx.x - j.blk0) * 256 + threadIdx.x, (size_t)j.nblk * 256);
}


static inline int wup_kpad(int k) { return ((k + WUP_CK - 1) / WUP_CK) * WUP_CK; }
static inline int wup_npad(int n) { return ((n + WUP_TCO - 1) / WUP_TCO) * WUP_TCO; }

extern "C" size_t sivae_pack_wino_up_weight_bytes(int Co, int Ci) {
  if (Co <= 0 || Ci <= 0) return 0;
  return (size_t)4 * wup_kpad(Ci) * wup_npad(Co) * 12 * sizeof(float);
}

extern "C" int sivae_pack_wino_up_weight(const float* w, float* up, int Co, int Ci, hipStream_t stream) {
  if (!w || !up) return SIVAE_ERR_NULL;
  if (Co <= 0 || Ci <= 0) return SIVAE_ERR_SHAPE;
  const int kpad = wup_kpad(Ci), npad = wup_npad(Co);
  int nb = cdiv((long long)kpad * npad, 256);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(pack_wino_up_kernel, dim3(nb), dim3(256), 0, stream, w, up, Co, Ci, kpad, npad);
  return sivae_launch_status();
}

// H, W = OUTPUT size.  Low-resolution width >= 16 (tile blocks of 8x16 or 4x32 low-res pixels), even low-res size
// not required; smaller maps use sivae_conv2d_wino_fwd with its upsample flag.
extern "C" int sivae_conv2d_wino_up_supported(int H, int W) {
  return (H >= 16 && W >= 32 && !(H & 1) && !(W & 3)) ? 1 : 0;  // (W % 4: 16-byte stores of four output pixels)
}

static inline bool wup_wide(int W) { return (W >> 1) >= 32; }

extern "C" int sivae_conv2d_wino_up_num_px_tiles(int B, int H, int W) {
  if (B <= 0 || !sivae_conv2d_wino_up_supported(H, W)) return SIVAE_ERR_SHAPE;
  const int pxh = wup_wide(W) ? 4 : 8, pxw = wup_wide(W) ? 32 : 16;
  return 4 * B * cdiv(H >> 1, pxh) * cdiv(W >> 1, pxw);
}

static int wup_grid_blocks() {
  static int g = 0;
  if (g == 0) {
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    g = 2 * cus;
  }
  return g;
}

template <int TTH_L2, int TTW_L2>
static int wup_launch(WinoUpArgs& a, hipStream_t stream) {
  constexpr int PXH = 2 << TTH_L2, PXW = 2 << TTW_L2;
  constexpr int PH = (1 << TTW_L2) + (1 << TTW_L2) / 4, PLANE = (PXH + 2) * 2 * PH;
  a.nbh = cdiv(a.H >> 1, PXH);
  a.nbw = cdiv(a.W >> 1, PXW);
  a.n_co_tiles = cdiv(a.Co, WUP_TCO);
  const long long nblk = (long long)a.B * a.nbh * a.nbw * a.n_co_tiles;
  if (nblk > 0x7fffffffLL) return SIVAE_ERR_RANGE;
  const size_t lds = (size_t)2 * WUP_CK * PLANE * sizeof(float) + (a.pro_mean ? (size_t)a.Ci_pad * 16 : 0) +
                     (size_t)4 * 16 * 2 * 64 * sizeof(float);
  auto kern = a.pro_mean ? conv_wino_up_kernel<TTH_L2, TTW_L2, true> : conv_wino_up_kernel<TTH_L2, TTW_L2, false>;
  {
    static size_t lds_hwm[2] = {0, 0};
    const int rc_lds = sivae_ensure_lds(reinterpret_cast<const void*>(kern), lds, &lds_hwm[a.pro_mean ? 1 : 0]);
    if (rc_lds != SIVAE_OK) return rc_lds;
  }
  a.n_items = (int)nblk;
  const int grid = nblk < wup_grid_blocks() ? (int)nblk : wup_grid_blocks();
  // (only while the packed filter stays L2-resident whichever channel tiles an XCD walks: with the plain order XCD x
  // sees channel tile x mod n_co_tiles only — the better deal for the wide deep layers, whose filter is 6-25 MB)
  a.xcd_group = (sivae_xcd_remap() && !(grid & 7) && 4ull * a.Ci_pad * a.Co_pad * 48ull <= (2ull << 20)) ? 1 : 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, a);
  return sivae_launch_status();
}

extern "C" int sivae_conv2d_wino_up_fwd(const float* x_half, const float* up, float* y, const float* pro_mean,
                                        const float* pro_invstd, const float* pro_gamma, const float* pro_beta,
                                        float pro_slope, float* stats_partial, int B, int Ci, int Co, int H, int W,
                                        hipStream_t stream) {
  if (!x_half || !up || !y) return SIVAE_ERR_NULL;
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  if (!sivae_conv2d_wino_up_supported(H, W)) return SIVAE_ERR_SHAPE;
  if (pro_mean && (!pro_invstd || !pro_gamma || !pro_beta)) return SIVAE_ERR_NULL;
  if (pro_mean && !(pro_slope >= 0.f && pro_slope <= 1.f)) return SIVAE_ERR_MODE;
  const long long hw = (long long)H * W;
  if ((long long)Ci * (hw / 4) * 4 >= 0x7fffffffLL || (long long)Co * hw * 4 >= 0x7fffffffLL) return SIVAE_ERR_RANGE;
  if (((uintptr_t)y & 15u) != 0) return SIVAE_ERR_SHAPE;  // 16-byte row stores
  WinoUpArgs a;
  a.x = x_half;
  a.up = up;
  a.y = y;
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
  a.Ci_pad = wup_kpad(Ci);
  a.Co_pad = wup_npad(Co);
  if (4ull * a.Ci_pad * a.Co_pad * 48ull >= 0xffffffffull) return SIVAE_ERR_RANGE;
  return wup_wide(W) ? wup_launch<1, 4>(a, stream) : wup_launch<2, 3>(a, stream);
}

// ---- batched packing (pack_batch.h)
int sivae_packjob_wino_up(SivaePackJob* j, int Co, int Ci) {
  j->kdim = Ci;
  j->ndim = Co;
  j->kpad = wup_kpad(Ci);
  j->npad = wup_npad(Co);
  j->total = (unsigned long long)j->kpad * j->npad;
  return SIVAE_OK;
}
void sivae_packbatch_wino_up(const SivaePackJob* jobs, const unsigned short* block_job, int nblocks, hipStream_t s) {
  hipLaunchKernelGGL(pack_wino_up_batch_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, jobs, block_job);
}
