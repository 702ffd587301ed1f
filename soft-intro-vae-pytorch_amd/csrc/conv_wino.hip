// Winograd F(2x2, 3x3) stride-1 "same" convolution for gfx950 on the exact-fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32): 16 multiplies per 2x2 output tile instead of 36 -> 2.25x fewer MFMA
// passes than the direct implicit GEMM of conv_fwd.hip for the same nn.Conv2d(k=3, s=1, p=1)
// (reference: soft_intro_vae/train_soft_intro_vae.py:56-61).  Forward and — fed the flipped /
// transposed weight transform — the data gradient.
//
//   V = B^T d B   (input 4x4 patch d of a tile, per input channel)
//   U = G g G^T   (3x3 filter g, per (co, ci); done once per optimizer step by sivae_pack_wino_weight)
//   M[i][j] = sum_ci U[i][j][co][ci] * V[i][j][ci][tile]      <- 16 independent GEMMs, K = Ci
//   Y = A^T M A   (2x2 outputs of the tile)
//
// Work split: a block is 4 waves = 64 output channels x 32 tiles (128 pixels) x 16 frequencies.
// Wave j owns frequency COLUMN j (i = 0..3): 4 freq x 2 co-subtiles of 32x32 accumulators = 128
// registers.  Nothing the MFMA loop consumes is shared between waves except the raw input halo:
//   * B operand: a lane (tile t = lane&31, channel k = lane>>5) reads the two raw columns its
//     frequency column needs (8 ds_read_b32 of the zero-padded LDS halo tile), forms
//     t = d[:,ca] +- d[:,cb] and V[0..3] = B^T t with 8 VALU ops — the transformed tile V is never
//     stored anywhere.
//   * A operand: U is packed [j][ci][co][i], so one 16-byte buffer load per (co-subtile, k-step)
//     gives the lane its four frequencies; loaded straight to registers one chunk (8 channels)
//     ahead (no LDS: no other wave wants it).
// After the K loop the row transform A^T M is done in registers, the column transform exchanges
// 2x32 KB through LDS (aliasing the halo buffers) and writes float2 pixel pairs, with the same
// fused epilogues as the direct kernel (bias, accumulate, BatchNorm sum/sumsq partials).
//
// LDS halo tile layout: [ck][row][parity][col/2] with row stride RS = 2*PH chosen so that the 32
// tiles of a half-wave hit 32 distinct banks (ds_read_b32 banks = dword address mod 32):
// 4x8 tiles: RS = 20 (row pair = 40 = 8 mod 32), 2x16 tiles: RS = 40 (row pair = 80 = 16 mod 32).
#include "common.h"
#include "pack_batch.h"
#include <stdlib.h>

struct WinoArgs {
  const float* x;
  const float* up;  // packed U [4(j)][Ci_pad][Co_pad][4(i)]
  float* y;
  const float* bias;
  const float* pro_mean;
  const float* pro_invstd;
  const float* pro_gamma;
  const float* pro_beta;
  float pro_slope;
  // segments: images [g*pro_seg_images, (g+1)*pro_seg_images) are pass g of the network with its own BatchNorm
  // statistics pro_mean / pro_invstd [pro_nseg][Ci] (gamma / beta shared); pro_nseg == 1: the whole batch is one pass
  int pro_seg_images, pro_nseg;
  float* stats;  // [n_px_tiles][Co][2] or null
  int B, Ci, Co, H, W;
  int Ci_pad, Co_pad;
  int nbh, nbw;
  int n_co_tiles;
  int accumulate;
  int upsample;
  int n_items;
  int xcd_group;  // 1: start items grouped per XCD (see the kernel)
  // split-K (small batches on the deep 4x4 / 8x8 maps: a handful of work items each walking 512 input channels):
  // n_items = k_splits * n_items_base; item (slice s, base item) accumulates chunks [s*chunks_per_split, ...) and
  // writes its partial output to y + s*y_split_stride (a workspace); a reduce kernel sums the slices
  int n_items_base;
  int chunks_per_split;
  long long y_split_stride;
  // optional: the output y is the gradient flowing into LeakyReLU(BatchNorm(bnb_x)) (the data gradient of conv2 feeding
  // BatchNorm-1's backward): the `stats` partials then hold {sum g, sum g * xhat}, g = y * LeakyReLU'(z), instead of
  // {sum y, sum y^2} — the first reduction pass of the BatchNorm backward disappears
  const float* bnb_x;
  const float* bnb_mean;
  const float* bnb_invstd;
  const float* bnb_gamma;
  const float* bnb_beta;
  float bnb_slope;
};

#define WINO_CK 16
// Two blocks share a CU (two waves per SIMD).  While one is in its MFMA phase — 64-cycle matrix instructions with
// plenty of issue slack — the other's serial tail (output transform, LDS exchange, stores, next item's first
// loads) runs at raised wave priority so that it is back in ITS MFMA phase sooner: +2-3% on the 64..128-channel
// layers; raising the MFMA phase instead measured +-0.
#define WINO_EPI_PRIO_UP __builtin_amdgcn_s_setprio(1);
#define WINO_EPI_PRIO_DOWN __builtin_amdgcn_s_setprio(0);
#define WINO_EX_FLOATS (2 * 4 * 2 * 16 * 64)  // output-transform exchange area (64 KB), aliases the halo buffers
#define WINO_TCO 64

// NG = output-channel groups per block (waves = 4*NG: group g, frequency column j), WM = 32-channel
// subtiles per wave; NG*WM*32 = 64 output channels per block either way.
//   <NG=1, WM=2>: 4 waves, 128 accumulator registers per wave, two blocks (8 waves) per CU
//   <NG=2, WM=1>: 8 waves,  64 accumulator registers per wave, two blocks (16 waves) per CU
template <int TTH_L2, int TTW_L2, bool PRO, int NG, int WM>
__global__ void __launch_bounds__(NG * 256, 2 * NG) conv_wino_kernel(WinoArgs a) {
  // tile block = TB images x TTH x TTW tiles.  The 32 tile columns of the MFMA hold 32/(TTH*TTW) images; that is
  // 2 for the 8x8 maps (one image = 16 tiles) and 8 for the 4x4 maps, where only TB = 4 are staged and computed
  // (their halos fill the 256 staging threads; the other 16 columns are discarded — it doubles the number of work
  // items of these tiny, otherwise machine-starving layers: 512->512 @4x4 bs128 is 256 items instead of 128)
  constexpr int TTH = 1 << TTH_L2, TTW = 1 << TTW_L2;
  constexpr int TBL = 32 / (TTH * TTW);
  constexpr int TB = TBL > 4 ? 4 : TBL, TB_L2 = TB == 4 ? 2 : (TB == 2 ? 1 : 0);
  static_assert(TBL * TTH * TTW == 32, "a block is 32 tile columns");
  static_assert(NG * WM * 32 == WINO_TCO, "64 output channels per block");
  constexpr int NT = NG * 256, NW = NG * 4;
  constexpr int PXH = 2 * TTH, PXW = 2 * TTW;
  constexpr int LH = PXH + 2, LWU = PXW + 2;
  // bank-conflict-free strides: 4x8 tiles RS 20, 2x16 tiles RS 40; 2 images x 4x4 tiles: RS 18 (a tile-row pair =
  // 36 = 4 mod 32) and an image stride of 208 (= 16 mod 32) -> bank = 16*tb + 4*ty + tx
  constexpr int PH = (TTW == 4) ? 9 : (TTW == 2 ? 3 : TTW + TTW / 4), RS = 2 * PH;
  constexpr int PLANE_IMG = (TB == 2) ? 208 : (TB == 4 ? 40 : LH * RS), PLANE = TB * PLANE_IMG;
  static_assert(PLANE_IMG >= LH * RS, "image plane holds the halo rows");
  constexpr int NPOS = TB * LH * LWU;
  static_assert(NPOS <= NT, "one halo position per thread");
  constexpr int CK = WINO_CK;
  constexpr int XBUF = CK * PLANE;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;
  // fused-BatchNorm parameters {mean, invstd*gamma, beta, -} per input channel, staged once per block behind the
  // 64 KB exchange area (reading them with scalar loads at halo-store time put 48 SMEM round trips and their
  // lgkmcnt(0) waits into every chunk: 159 instead of 209 TF on 512->512)
  float4* pro4 = reinterpret_cast<float4*>(smem + WINO_EX_FLOATS);

  const int tid = threadIdx.x, lane = tid & 63;
  // wave index as an SGPR: the U loads use it in their scalar offset (a VGPR there costs a waterfall loop per load)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int wj = wave & 3, wg = wave >> 2;
  const int H = a.H, W = a.W;
  const int Hs = a.upsample ? (H >> 1) : H, Ws = a.upsample ? (W >> 1) : W;
  const int HWs = Hs * Ws;

  // Work items (co tile, tile block), co tile fastest.  A block walks items bid, bid + gridDim, ... (the grid is
  // two blocks per CU): the first halo chunk and U operands of the NEXT item are requested before the output
  // transform of the current one, so launch latency, address set-up and the first HBM round trip are paid
  // once per block instead of once per 128 output pixels.
  const int n_items = a.n_items;
  const int nchunks = a.Ci_pad / WINO_CK;
  // Consecutive blockIdx go round-robin to the 8 XCDs.  With xcd_group the block on XCD x, slot j starts at item
  // x * (grid / 8) + j: the co-tiles of one pixel tile (consecutive items) then run on ONE XCD and share the halo in its
  // L2.  Only for layers whose whole U (all co-tiles) fits an L2 next to the halos; with 8 co-tiles of 2 MB each the
  // plain order is the better one (co-tile c = XCD c keeps ONE U slab resident).
  int item = blockIdx.x;
  if (a.xcd_group) item = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
  int pt, b, r0, c0, co0;
  int kc0, kc1, ysl;  // chunk range and output slice of the current item (split-K; the whole K range otherwise)
  __amdgpu_buffer_rsrc_t xrsrc;
  const __amdgpu_buffer_rsrc_t ursrc = make_rsrc(a.up, 16ull * a.Ci_pad * a.Co_pad * 4ull);
  unsigned xo, ua_base;
  // every thread stages a halo slot: threads beyond the NPOS slots duplicate the first ones (same address, same
  // value), so the register -> LDS copy is straight-line code — with predicated stores and a select around the
  // prologue the compiler emitted one exec-mask branch and one serialised ds_read + lgkmcnt(0) per element
  const int teff = tid % NPOS;
  const int xtb = teff / (LH * LWU), xrr = (teff / LWU) % LH, xcc = teff % LWU;
  float xmask = 0.f;  // 1 inside the image, 0 on the zero padding (applied after the fused BatchNorm+LeakyReLU)
  int pseg = 0;       // offset of this thread's segment in the staged prologue table (PRO with pro_nseg > 1)
  const f32x2 pslope2 = {a.pro_slope, a.pro_slope};
#define xmask2 (f32x2{xmask, xmask})
  const int xl = xtb * PLANE_IMG + xrr * RS + (xcc & 1) * PH + (xcc >> 1);
  int nb_here;  // images of this item that exist (TB == 2 and odd batch: the last item has one)
#define WINO_SETUP(ITEM)                                                 \
  {                                                                      \
    ysl = (ITEM) / a.n_items_base;                                       \
    const int it_ = (ITEM) - ysl * a.n_items_base;                       \
    kc0 = ysl * a.chunks_per_split;                                      \
    kc1 = kc0 + a.chunks_per_split < nchunks ? kc0 + a.chunks_per_split : nchunks; \
    const int co_tile = it_ % a.n_co_tiles;                              \
    pt = it_ / a.n_co_tiles;                                             \
    const int tbx = pt % a.nbw;                                          \
    const int t2 = pt / a.nbw;                                           \
    const int tby = t2 % a.nbh;                                          \
    b = (t2 / a.nbh) << TB_L2;                                           \
    nb_here = a.B - b < TB ? a.B - b : TB;                               \
    r0 = tby * PXH;                                                      \
    c0 = tbx * PXW;                                                      \
    co0 = co_tile * WINO_TCO;                                            \
    xrsrc = make_rsrc(a.x + (size_t)b * a.Ci * HWs, (unsigned long long)nb_here * a.Ci * HWs * 4ull); \
    const int r = r0 + xrr - 1, c = c0 + xcc - 1;                        \
    xo = SIVAE_OOB;                                                      \
    xmask = 0.f;                                                         \
    if (xtb < nb_here && r >= 0 && r < H && c >= 0 && c < W) {           \
      xmask = 1.f;                                                       \
      const int rs = a.upsample ? (r >> 1) : r, cs = a.upsample ? (c >> 1) : c; \
      xo = (unsigned)((xtb * a.Ci * Hs + rs) * Ws + cs) * 4u;            \
    }                                                                    \
    ua_base = (unsigned)((wj * a.Ci_pad) * a.Co_pad + co0) * 16u;        \
    if (PRO && a.pro_nseg > 1) {                                         \
      const int bi_ = b + xtb < a.B ? b + xtb : a.B - 1;                 \
      pseg = (bi_ / a.pro_seg_images) * a.Ci_pad;                        \
    }                                                                    \
  }

  // ---- A operand (U) addressing: lane -> (ci = k-step*2 + hh, co = co0 + (wg*WM + m)*32 + l31), 16 B each
  const unsigned va0 = (unsigned)(hh * a.Co_pad + wg * WM * 32 + l31) * 16u;
  const unsigned ua_step = (unsigned)a.Co_pad * 16u;                            // bytes per input channel

  // ---- B operand: raw columns (ca, cb) and sign of frequency column j
  const int tx = l31 & (TTW - 1), ty = (l31 >> TTW_L2) & (TTH - 1), tb = l31 >> (TTW_L2 + TTH_L2);
  const int ca = (wj == 0) ? 0 : ((wj == 2) ? 2 : 1);
  const int cb = (wj == 0) ? 2 : ((wj == 1) ? 2 : ((wj == 2) ? 1 : 3));
  const float sgn = (wj == 1) ? 1.f : -1.f;
  const int bb = hh * PLANE + tb * PLANE_IMG + 2 * ty * RS + tx;
  const int base_a = bb + (ca & 1) * PH + (ca >> 1);
  const int base_b = bb + (cb & 1) * PH + (cb >> 1);

  f32x16 acc[4][WM];
  float xr[CK];
  float4 AR[4][WM];  // ring of U operands: slot (k-step & 3); a slot is refilled with k-step + 4 right after the
                     // MFMAs that consumed it have been issued (prefetch distance = 4 k-steps, across chunks)

#define WINO_LOAD_X(CH)                                                  \
  {                                                                      \
    _Pragma("unroll") for (int ck = 0; ck < CK; ++ck) {                  \
      const int ci = (CH)*CK + ck;                                       \
      const int cic = ci < a.Ci ? ci : a.Ci - 1;                         \
      xr[ck] = buf_load_f32(xrsrc, xo, (unsigned)cic * (unsigned)HWs * 4u); \
    }                                                                    \
  }
#define WINO_LOAD_A(KS_ABS, SLOT)                                        \
  {                                                                      \
    const unsigned so = ua_base + (unsigned)(2 * (KS_ABS)) * ua_step;    \
    _Pragma("unroll") for (int m = 0; m < WM; ++m) AR[SLOT][m] = buf_load_f32x4(ursrc, va0 + m * 512u, so); \
  }
#define WINO_STORE_X(CH, BUF)                                            \
  {                                                                      \
    if (PRO) { /* two channels per packed-fp32 op; padded channels carry all-zero parameters -> 0 */ \
      _Pragma("unroll") for (int cp = 0; cp < CK / 2; ++cp) {            \
        const float4 p0 = pro4[pseg + (CH)*CK + 2 * cp];     /* mean0 mean1 scale0 scale1 */ \
        const float4 p1 = pro4[pseg + (CH)*CK + 2 * cp + 1]; /* beta0 beta1 */   \
        f32x2 v = {xr[2 * cp], xr[2 * cp + 1]};                          \
        const f32x2 pm = {p0.x, p0.y}, ps = {p0.z, p0.w}, pb = {p1.x, p1.y}; \
        v = (v - pm) * ps + pb;                                          \
        const f32x2 u = v * pslope2;                                     \
        v.x = fmaxf(v.x, u.x);                                           \
        v.y = fmaxf(v.y, u.y);                                           \
        v = v * xmask2;                                                  \
        xs[(BUF)*XBUF + (2 * cp) * PLANE + xl] = v.x;                    \
        xs[(BUF)*XBUF + (2 * cp + 1) * PLANE + xl] = v.y;                \
      }                                                                  \
    } else {                                                             \
      _Pragma("unroll") for (int ck = 0; ck < CK; ++ck) {                \
        const int ci = (CH)*CK + ck;                                     \
        xs[(BUF)*XBUF + ck * PLANE + xl] = ci < a.Ci ? xr[ck] : 0.f;     \
      }                                                                  \
    }                                                                    \
  }
  // raw halo reads of k-step KK (two columns x four rows) — issued one k-step ahead of their use so the LDS
  // latency and the 8 transform VALU ops sit under the previous k-step's MFMAs
#define WINO_READ(BUF, KK, DA, DB)                                       \
  {                                                                      \
    const float* pa = xs + (BUF)*XBUF + 2 * (KK)*PLANE + base_a;         \
    const float* pb_ = xs + (BUF)*XBUF + 2 * (KK)*PLANE + base_b;        \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                      \
      DA[r] = pa[r * RS];                                                \
      DB[r] = pb_[r * RS];                                               \
    }                                                                    \
  }
#define WINO_STEP(SLOT, DA, DB)                                          \
  {                                                                      \
    const float t0 = DA[0] + sgn * DB[0], t1 = DA[1] + sgn * DB[1];      \
    const float t2_ = DA[2] + sgn * DB[2], t3 = DA[3] + sgn * DB[3];     \
    const float v0 = t0 - t2_, v1 = t1 + t2_, v2 = t2_ - t1, v3 = t1 - t3; \
    _Pragma("unroll") for (int m = 0; m < WM; ++m) {                     \
      acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(AR[SLOT][m].x, v0, acc[0][m], 0, 0, 0); \
      acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(AR[SLOT][m].y, v1, acc[1][m], 0, 0, 0); \
      acc[2][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(AR[SLOT][m].z, v2, acc[2][m], 0, 0, 0); \
      acc[3][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(AR[SLOT][m].w, v3, acc[3][m], 0, 0, 0); \
    }                                                                    \
  }
  // one k-step: prefetch the raw reads of k-step KK+1 (into DAN/DBN), run k-step KK, refill its U slot
#define WINO_KSTEP(CH, BUF, KK, DA, DB, DAN, DBN)                        \
  {                                                                      \
    if ((KK) + 1 < CK / 2) WINO_READ(BUF, (KK) + 1, DAN, DBN)            \
    __builtin_amdgcn_sched_barrier(0);                                   \
    WINO_STEP((KK)&3, DA, DB)                                            \
    __builtin_amdgcn_sched_barrier(0);                                   \
    if ((CH) * (CK / 2) + (KK) + 4 < kc1 * (CK / 2)) WINO_LOAD_A((CH) * (CK / 2) + (KK) + 4, (KK)&3) \
  }
  // MFMA phase of chunk CH on halo buffer BUF.  The staged registers of chunk CH+1 (loaded at the top) are
  // written to the OTHER halo buffer late in the phase, so the only thing between two MFMA phases is the barrier.
#define WINO_MMA(CH, BUF, NEXT)                                          \
  {                                                                      \
    float da0[4], db0[4], da1[4], db1[4];                                \
    if (NEXT) WINO_LOAD_X((CH) + 1)                                      \
    WINO_READ(BUF, 0, da0, db0)                                          \
    WINO_KSTEP(CH, BUF, 0, da0, db0, da1, db1)                           \
    WINO_KSTEP(CH, BUF, 1, da1, db1, da0, db0)                           \
    WINO_KSTEP(CH, BUF, 2, da0, db0, da1, db1)                           \
    WINO_KSTEP(CH, BUF, 3, da1, db1, da0, db0)                           \
    WINO_KSTEP(CH, BUF, 4, da0, db0, da1, db1)                           \
    WINO_KSTEP(CH, BUF, 5, da1, db1, da0, db0)                           \
    WINO_KSTEP(CH, BUF, 6, da0, db0, da1, db1)                           \
    if (NEXT) WINO_STORE_X((CH) + 1, (BUF) ^ 1)                          \
    __builtin_amdgcn_sched_barrier(0);                                   \
    WINO_KSTEP(CH, BUF, 7, da1, db1, da0, db0)                           \
    __syncthreads();                                                     \
  }

  // ---- output transform of the item at (e_b, e_r0, e_c0, e_co0).  acc[i][m][r]: tile = l31, channel =
  // (wg*WM + m)*32 + (r&3) + 8*(r>>2) + 4*hh.  Row transform in registers, column transform across the four
  // frequency-column waves through LDS: ex[j][cg][r][lane], cg = 32-channel group in the block; NW waves share
  // the 32 (cg, r) rows.  (The K loop ended on a barrier, so the halo buffers are free to alias; the last
  // barrier of the epilogue frees them again for the next item's halo.)
#define WINO_EPILOGUE                                                     \
  {                                                                      \
    float* ex = smem; /* [2 ar][4 j][2 cg][16 r][64 lanes] = 64 KB */    \
    constexpr int PPW = 32 / NW;                                         \
    const __amdgpu_buffer_rsrc_t yrsrc =                                 \
        make_rsrc(a.y + (size_t)e_ysl * a.y_split_stride + (size_t)e_b * a.Co * H * W, \
                  (unsigned long long)e_nb * a.Co * H * W * 4ull);       \
    const int row_base = e_r0 + 2 * ty, col = e_c0 + 2 * tx;             \
    _Pragma("unroll") for (int m = 0; m < WM; ++m)                       \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                     \
      ex[((wj * 2 + wg * WM + m) * 16 + r) * 64 + lane] = acc[0][m][r] + acc[1][m][r] + acc[2][m][r]; \
      ex[(((4 + wj) * 2 + wg * WM + m) * 16 + r) * 64 + lane] = acc[1][m][r] - acc[2][m][r] - acc[3][m][r]; \
    }                                                                    \
    /* byte offset of (chn, row_base, col) per (cg, r) row of this wave; invalid channels / columns / rows get */ \
    /* an out-of-range offset, so their stores are dropped (and accumulate-loads return 0) without branches */ \
    unsigned yo[PPW];                                                    \
    _Pragma("unroll") for (int rr = 0; rr < PPW; ++rr) {                 \
      const int p = wave * PPW + rr;                                     \
      const int chn = e_co0 + (p >> 4) * 32 + (p & 3) + 8 * ((p & 15) >> 2) + 4 * hh; \
      yo[rr] = (chn < a.Co && col < W && tb < e_nb) ? (unsigned)(((tb * a.Co + chn) * H + row_base) * W + col) * 4u : SIVAE_OOB; \
    }                                                                    \
    const bool bnb = a.bnb_x != nullptr;                                 \
    const __amdgpu_buffer_rsrc_t brsrc =                                 \
        make_rsrc((bnb ? a.bnb_x : a.y) + (size_t)e_b * a.Co * H * W, (unsigned long long)e_nb * a.Co * H * W * 4ull); \
    __syncthreads();                                                     \
    _Pragma("unroll") for (int rr = 0; rr < PPW; ++rr) {                 \
      const int p = wave * PPW + rr;                                     \
      float ssum = 0.f, ssq = 0.f;                                       \
      float bm = 0.f, bis = 0.f, bgs = 0.f, bbt = 0.f;                   \
      if (bnb) {                                                         \
        const int chn_ = e_co0 + (p >> 4) * 32 + (p & 3) + 8 * ((p & 15) >> 2) + 4 * hh; \
        const int cc_ = chn_ < a.Co ? chn_ : a.Co - 1;                   \
        bm = a.bnb_mean[cc_];                                            \
        bis = a.bnb_invstd[cc_];                                         \
        bgs = bis * a.bnb_gamma[cc_];                                    \
        bbt = a.bnb_beta[cc_];                                           \
      }                                                                  \
      _Pragma("unroll") for (int ar = 0; ar < 2; ++ar) {                 \
        float e[4];                                                      \
        _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) e[jj] = ex[(((ar * 4 + jj) * 2 + (p >> 4)) * 16 + (p & 15)) * 64 + lane]; \
        const bool ok = yo[rr] != SIVAE_OOB && row_base + ar < H;        \
        const unsigned off = ok ? yo[rr] + (unsigned)(ar * W) * 4u : SIVAE_OOB; \
        float y0 = e[0] + e[1] + e[2];                                   \
        float y1 = e[1] - e[2] - e[3];                                   \
        if (a.accumulate) {                                              \
          const float2 o = buf_load_f32x2(yrsrc, off, 0u);               \
          y0 += o.x;                                                     \
          y1 += o.y;                                                     \
        }                                                                \
        buf_store_f32x2(yrsrc, y0, y1, off, 0u);                         \
        if (bnb) {                                                       \
          const float2 xv = buf_load_f32x2(brsrc, off, 0u);              \
          const float g0 = ((xv.x - bm) * bgs + bbt) > 0.f ? y0 : y0 * a.bnb_slope; \
          const float g1 = ((xv.y - bm) * bgs + bbt) > 0.f ? y1 : y1 * a.bnb_slope; \
          ssum += ok ? (g0 + g1) : 0.f;                                  \
          ssq += ok ? (g0 * ((xv.x - bm) * bis) + g1 * ((xv.y - bm) * bis)) : 0.f; \
        } else {                                                         \
          ssum += ok ? (y0 + y1) : 0.f;                                  \
          ssq += ok ? (y0 * y0 + y1 * y1) : 0.f;                         \
        }                                                                \
      }                                                                  \
      if (a.stats != nullptr) {                                          \
        const int chn = e_co0 + (p >> 4) * 32 + (p & 3) + 8 * ((p & 15) >> 2) + 4 * hh; \
        const float s = half_wave_sum_hi(ssum);                          \
        const float q = half_wave_sum_hi(ssq);                           \
        if (l31 == 31 && chn < a.Co) {                                   \
          float* dst = a.stats + ((size_t)e_pt * a.Co + chn) * 2;        \
          dst[0] = s;                                                    \
          dst[1] = q;                                                    \
        }                                                                \
      }                                                                  \
    }                                                                    \
    __syncthreads();                                                     \
  }

  // One barrier per 16-channel chunk, at the end of its MFMA phase: it publishes the halo of chunk ch+1 (written
  // during the phase into the other buffer) and retires the readers of buffer (ch & 1) before chunk ch+1's
  // phase overwrites it with chunk ch+2.
  if (PRO) {
    // pro4[2p] = {mean, mean', scale, scale'}, pro4[2p + 1] = {beta, beta', 0, 0} of the channel pair (2p, 2p + 1)
    for (int idx = tid; idx < a.pro_nseg * a.Ci_pad; idx += NT) {
      const int c = idx % a.Ci_pad, so = (idx / a.Ci_pad) * a.Ci;  // (segment g's statistics start at g * Ci)
      const int c0_ = c & ~1, c1_ = c | 1;
      const bool k0 = c0_ < a.Ci, k1 = c1_ < a.Ci;
      if ((c & 1) == 0)
        pro4[idx] = make_float4(k0 ? a.pro_mean[so + c0_] : 0.f, k1 ? a.pro_mean[so + c1_] : 0.f,
                                k0 ? a.pro_invstd[so + c0_] * a.pro_gamma[c0_] : 0.f,
                                k1 ? a.pro_invstd[so + c1_] * a.pro_gamma[c1_] : 0.f);
      else
        pro4[idx] = make_float4(k0 ? a.pro_beta[c0_] : 0.f, k1 ? a.pro_beta[c1_] : 0.f, 0.f, 0.f);
    }
    __syncthreads();
  }
  WINO_SETUP(item)
  WINO_LOAD_X(kc0)
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) WINO_LOAD_A(kc0 * (CK / 2) + kk, kk)
  for (;;) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][m][r] = 0.f;
    WINO_STORE_X(kc0, 0)
    __syncthreads();
    int ch = kc0;
    for (; ch + 1 < kc1; ch += 2) {
      WINO_MMA(ch, 0, true)
      const bool more = ch + 2 < kc1;
      WINO_MMA(ch + 1, 1, more)
    }
    if (ch < kc1) WINO_MMA(ch, 0, false)

    WINO_EPI_PRIO_UP
    // coordinates of the item just accumulated; then put the next item's first loads in flight
    const int e_pt = pt, e_b = b, e_nb = nb_here, e_r0 = r0, e_c0 = c0, e_co0 = co0, e_ysl = ysl;
    const int next = item + (int)gridDim.x;
    const bool has_next = next < n_items;
    // (with accumulate the epilogue loads y; vmcnt completes in order, so the prefetch goes after it)
    const bool early = has_next && !a.accumulate && a.bnb_x == nullptr;
    if (early) {
      WINO_SETUP(next)
      WINO_LOAD_X(kc0)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) WINO_LOAD_A(kc0 * (CK / 2) + kk, kk)
    }
    WINO_EPILOGUE
    if (has_next && !early) {
      WINO_SETUP(next)
      WINO_LOAD_X(kc0)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) WINO_LOAD_A(kc0 * (CK / 2) + kk, kk)
    }
    WINO_EPI_PRIO_DOWN
    if (!has_next) break;
    item = next;
  }
#undef WINO_SETUP
#undef WINO_EPILOGUE
#undef WINO_LOAD_X
#undef WINO_LOAD_A
#undef WINO_STORE_X
#undef WINO_MMA
#undef WINO_KSTEP
#undef WINO_READ
#undef WINO_STEP
#undef xmask2

}

// ---- weight transform U = G g G^T, packed [j][ci_pad][co_pad][i]; padding entries are zero
//   mode 0 (forward): g = w[n][k]            (n = output channel, k = input channel)
//   mode 1 (dgrad):   g = flip180(w[k][n])   (k = w's output channel is the GEMM's input channel)
__device__ __forceinline__ void pack_wino_body(const float* __restrict__ w, float* __restrict__ up, int Co,
                                                        int Ci, int mode, int kdim, int ndim, int kpad, int npad, size_t idx0_, const size_t stride_) {
  const size_t total = (size_t)kpad * npad;
  for (size_t idx = idx0_; idx < total; idx += stride_) {
    const int n = (int)(idx % npad), k = (int)(idx / npad);
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
    float gg[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      gg[0][c] = g[0][c];
      gg[1][c] = 0.5f * (g[0][c] + g[1][c] + g[2][c]);
      gg[2][c] = 0.5f * (g[0][c] - g[1][c] + g[2][c]);
      gg[3][c] = g[2][c];
    }
    float u[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u[i][0] = gg[i][0];
      u[i][1] = 0.5f * (gg[i][0] + gg[i][1] + gg[i][2]);
      u[i][2] = 0.5f * (gg[i][0] - gg[i][1] + gg[i][2]);
      u[i][3] = gg[i][2];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4* dst = reinterpret_cast<float4*>(up + (((size_t)j * kpad + k) * npad + n) * 4);
      *dst = make_float4(u[0][j], u[1][j], u[2][j], u[3][j]);
    }
  }
}

__global__ void __launch_bounds__(256) pack_wino_kernel(const float* __restrict__ w, float* __restrict__ up, int Co,
                                                        int Ci, int mode, int kdim, int ndim, int kpad, int npad) {
  pack_wino_body(w, up, Co, Ci, mode, kdim, ndim, kpad, npad, (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)gridDim.x * 256);
}

__global__ void __launch_bounds__(256) pack_wino_batch_kernel(const SivaePackJob* __restrict__ jobs,
                                                               const unsigned short* __restrict__ block_job) {
  const SivaePackJob j = jobs[block_job[blockIdx.x]];
  pack_wino_body(j.w, j.dst, j.Co, j.Ci, j.mode, j.kdim, j.ndim, j.kpad, j.npad, (size_t)(blockIdx.x - j.blk0) * 256 + threadIdx.x, (size_t)j.nblk * 256);
}


static inline int wino_kpad(int k) { return ((k + WINO_CK - 1) / WINO_CK) * WINO_CK; }
static inline int wino_npad(int n) { return ((n + WINO_TCO - 1) / WINO_TCO) * WINO_TCO; }

extern "C" size_t sivae_pack_wino_weight_bytes(int Co, int Ci, int mode) {
  if (Co <= 0 || Ci <= 0 || (mode != 0 && mode != 1)) return 0;
  const int kdim = mode == 0 ? Ci : Co, ndim = mode == 0 ? Co : Ci;
  return (size_t)16 * wino_kpad(kdim) * wino_npad(ndim) * sizeof(float);
}

extern "C" int sivae_pack_wino_weight(const float* w, float* up, int Co, int Ci, int mode, hipStream_t stream) {
  if (!w || !up) return SIVAE_ERR_NULL;
  if (Co <= 0 || Ci <= 0) return SIVAE_ERR_SHAPE;
  if (mode != 0 && mode != 1) return SIVAE_ERR_MODE;
  const int kdim = mode == 0 ? Ci : Co, ndim = mode == 0 ? Co : Ci;
  const int kpad = wino_kpad(kdim), npad = wino_npad(ndim);
  int nb = cdiv((long long)kpad * npad, 256);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(pack_wino_kernel, dim3(nb), dim3(256), 0, stream, w, up, Co, Ci, mode, kdim, ndim, kpad, npad);
  return sivae_launch_status();
}

// Winograd path handles even H >= 8 and even W >= 16 (tiles are whole 2x2 blocks), plus the 8x8 and 4x4 maps with
// two / four images per tile block; other sizes stay on the direct kernel.
extern "C" int sivae_conv2d_wino_supported(int H, int W) {
  if ((H == 8 && W == 8) || (H == 4 && W == 4)) return 1;
  return (H >= 8 && W >= 16 && !(H & 1) && !(W & 1)) ? 1 : 0;
}

static inline bool wino_wide(int W) { return W >= 32; }  // 2x16 tiles (4x32 px) vs 4x8 tiles (8x16 px)

extern "C" int sivae_conv2d_wino_num_px_tiles(int B, int H, int W) {
  if (B <= 0 || !sivae_conv2d_wino_supported(H, W)) return SIVAE_ERR_SHAPE;
  if (W == 8) return cdiv(B, 2);
  if (W == 4) return cdiv(B, 4);
  const int pxh = wino_wide(W) ? 4 : 8, pxw = wino_wide(W) ? 32 : 16;
  return B * cdiv(H, pxh) * cdiv(W, pxw);
}

// persistent grid: two co-resident blocks per CU
static int wino_grid_blocks() {
  static int g = 0;
  if (g == 0) {
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    const char* e = getenv("SIVAE_WINO_GRID");
    g = e ? atoi(e) : 2 * cus;
    if (g <= 0) g = 2 * cus;
  }
  return g;
}

template <int TTH_L2, int TTW_L2, int NG, int WM>
static int wino_launch(WinoArgs& a, hipStream_t stream) {
  constexpr int PXH = 2 << TTH_L2, PXW = 2 << TTW_L2;
  constexpr int TBL = 32 >> (TTH_L2 + TTW_L2), TB = TBL > 4 ? 4 : TBL;
  constexpr int PH = (TTW_L2 == 2) ? 9 : (TTW_L2 == 1 ? 3 : (1 << TTW_L2) + (1 << TTW_L2) / 4);
  constexpr int PLANE = (TB == 2) ? 2 * 208 : (TB == 4 ? 4 * 40 : (PXH + 2) * 2 * PH);
  a.nbh = cdiv(a.H, PXH);
  a.nbw = cdiv(a.W, PXW);
  a.n_co_tiles = cdiv(a.Co, WINO_TCO);
  const long long nblk = (long long)cdiv(a.B, TB) * a.nbh * a.nbw * a.n_co_tiles;
  if (nblk > 0x7fffffffLL) return SIVAE_ERR_RANGE;
  size_t lds = (size_t)2 * WINO_CK * PLANE * sizeof(float);
  const size_t exch = (size_t)WINO_EX_FLOATS * sizeof(float);
  if (lds > exch) return SIVAE_ERR_SHAPE;  // (halo buffers alias the exchange area)
  lds = exch + (a.pro_mean ? (size_t)a.pro_nseg * a.Ci_pad * 16 : 0);
  if (lds > 80 * 1024) return SIVAE_ERR_SHAPE;
  auto kern = a.pro_mean ? conv_wino_kernel<TTH_L2, TTW_L2, true, NG, WM> : conv_wino_kernel<TTH_L2, TTW_L2, false, NG, WM>;
  {
    static size_t lds_hwm[2] = {0, 0};  // per template instantiation, per prologue variant
    const int rc_lds = sivae_ensure_lds(reinterpret_cast<const void*>(kern), lds, &lds_hwm[a.pro_mean ? 1 : 0]);
    if (rc_lds != SIVAE_OK) return rc_lds;
  }
  a.n_items_base = (int)nblk;
  if (a.chunks_per_split <= 0) {  // no split: one slice = the whole K range
    a.chunks_per_split = a.Ci_pad / WINO_CK;
    a.y_split_stride = 0;
  }
  const int k_splits = cdiv(a.Ci_pad / WINO_CK, a.chunks_per_split);
  const long long nitems = nblk * k_splits;
  if (nitems > 0x7fffffffLL) return SIVAE_ERR_RANGE;
  a.n_items = (int)nitems;
  const int grid = nitems < wino_grid_blocks() ? (int)nitems : wino_grid_blocks();
  a.xcd_group = (sivae_xcd_remap() && k_splits == 1 && a.n_co_tiles > 1 && !(grid & 7) &&
                 (size_t)a.Ci_pad * a.Co_pad * 64 <= (size_t)2 << 20) ? 1 : 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NG * 256), lds, stream, a);
  return sivae_launch_status();
}

static int wino_fwd_impl(const float* x, const float* up, float* y, const float* bias, const float* pro_mean,
                         const float* pro_invstd, const float* pro_gamma, const float* pro_beta, float pro_slope,
                         float* stats_partial, const float* bnb_x, const float* bnb_mean, const float* bnb_invstd,
                         const float* bnb_gamma, const float* bnb_beta, float bnb_slope, int B, int Ci, int Co, int H,
                         int W, int upsample, int accumulate, hipStream_t stream, int chunks_per_split = 0,
                         long long y_split_stride = 0, int pro_seg_images = 0) {
  if (!x || !up || !y) return SIVAE_ERR_NULL;
  if (bias) return SIVAE_ERR_MODE;  // none of the 3x3 convs has a bias (:56-61); sivae_conv2d_fwd handles that case
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  if (!sivae_conv2d_wino_supported(H, W)) return SIVAE_ERR_SHAPE;
  if (pro_mean && (!pro_invstd || !pro_gamma || !pro_beta)) return SIVAE_ERR_NULL;
  if (pro_mean && !(pro_slope >= 0.f && pro_slope <= 1.f)) return SIVAE_ERR_MODE;  // prologue uses max(v, v*slope)
  if (((uintptr_t)y & 7u) != 0) return SIVAE_ERR_SHAPE;  // float2 stores
  const long long hw = (long long)H * W;
  if ((long long)B * Co * hw >= 0xffffffffLL) return SIVAE_ERR_RANGE;
  if ((long long)Ci * hw * 4 >= 0x7fffffffLL || (long long)Co * hw * 4 >= 0x7fffffffLL) return SIVAE_ERR_RANGE;
  WinoArgs a;
  a.x = x;
  a.up = up;
  a.y = y;
  a.bias = bias;
  a.pro_mean = pro_mean;
  a.pro_invstd = pro_invstd;
  a.pro_gamma = pro_gamma;
  a.pro_beta = pro_beta;
  a.pro_slope = pro_slope;
  if (pro_seg_images < 0 || (pro_seg_images > 0 && B % pro_seg_images != 0)) return SIVAE_ERR_SHAPE;
  a.pro_seg_images = pro_seg_images > 0 ? pro_seg_images : B;
  a.pro_nseg = B / a.pro_seg_images;
  // (a tile block of the 8x8 / 4x4 maps holds 2 / 4 images: a segment must be a whole number of tile blocks)
  if (a.pro_nseg > 1 && (a.pro_seg_images % ((W == 8) ? 2 : (W == 4 ? 4 : 1))) != 0) return SIVAE_ERR_SHAPE;
  a.stats = stats_partial;
  a.B = B;
  a.Ci = Ci;
  a.Co = Co;
  a.H = H;
  a.W = W;
  a.Ci_pad = wino_kpad(Ci);
  a.Co_pad = wino_npad(Co);
  if (16ull * a.Ci_pad * a.Co_pad * 4ull >= 0xffffffffull) return SIVAE_ERR_RANGE;
  a.accumulate = accumulate;
  a.upsample = upsample;
  a.chunks_per_split = chunks_per_split;
  a.y_split_stride = y_split_stride;
  if (bnb_x && (!bnb_mean || !bnb_invstd || !bnb_gamma || !bnb_beta || !stats_partial)) return SIVAE_ERR_NULL;
  a.bnb_x = bnb_x;
  a.bnb_mean = bnb_mean;
  a.bnb_invstd = bnb_invstd;
  a.bnb_gamma = bnb_gamma;
  a.bnb_beta = bnb_beta;
  a.bnb_slope = bnb_slope;
  // production: 4-wave blocks, 64co x 32 tiles per block, 128 accumulator registers per wave (119 TF issued =
  // 268 TF algorithmic on 512->512 @32x32; the 8-wave <NG=2,WM=1> split measured 106 TF)
  if (W == 8) return wino_launch<2, 2, 1, 2>(a, stream);  // 8x8 maps: 2 images x 4x4 tiles per block
  if (W == 4) return wino_launch<1, 1, 1, 2>(a, stream);  // 4x4 maps: 4 images x 2x2 tiles per block
  return wino_wide(W) ? wino_launch<1, 4, 1, 2>(a, stream) : wino_launch<2, 3, 1, 2>(a, stream);
}

extern "C" int sivae_conv2d_wino_fwd(const float* x, const float* up, float* y, const float* bias,
                                     const float* pro_mean, const float* pro_invstd, const float* pro_gamma,
                                     const float* pro_beta, float pro_slope, float* stats_partial, int B, int Ci,
                                     int Co, int H, int W, int upsample, int accumulate, hipStream_t stream) {
  return wino_fwd_impl(x, up, y, bias, pro_mean, pro_invstd, pro_gamma, pro_beta, pro_slope, stats_partial, nullptr,
                       nullptr, nullptr, nullptr, nullptr, 1.f, B, Ci, Co, H, W, upsample, accumulate, stream);
}

// Segmented batch (B = nseg * seg_images): the fused BatchNorm prologue takes pro_mean / pro_invstd [nseg][Ci]; the
// statistics partials come out in image order, so rows [g*n/nseg, (g+1)*n/nseg) belong to segment g
// (sivae_bn_stats_from_conv_seg).  With 8x8 / 4x4 maps seg_images must be a multiple of 2 / 4.
extern "C" int sivae_conv2d_wino_fwd_seg(const float* x, const float* up, float* y, const float* pro_mean,
                                         const float* pro_invstd, const float* pro_gamma, const float* pro_beta,
                                         float pro_slope, float* stats_partial, int B, int Ci, int Co, int H, int W,
                                         int upsample, int accumulate, int seg_images, hipStream_t stream) {
  if (seg_images <= 0) return SIVAE_ERR_SHAPE;
  return wino_fwd_impl(x, up, y, nullptr, pro_mean, pro_invstd, pro_gamma, pro_beta, pro_slope, stats_partial, nullptr,
                       nullptr, nullptr, nullptr, nullptr, 1.f, B, Ci, Co, H, W, upsample, accumulate, stream, 0, 0,
                       seg_images);
}

// Data-gradient use feeding a BatchNorm backward: y = dL/dh with h = LeakyReLU(BatchNorm(bn_x)) (bn_x has y's shape);
// bnbwd_partial [sivae_conv2d_wino_num_px_tiles][Co][2] receives per tile {sum g, sum g * xhat}, g = y * LeakyReLU'(z)
// — the first reduction of sivae_bn_bwd (see sivae_bn_bwd_from_partials).
extern "C" int sivae_conv2d_wino_dgrad_bnbwd(const float* dy, const float* up, float* y, const float* bn_x,
                                             const float* bn_mean, const float* bn_invstd, const float* bn_gamma,
                                             const float* bn_beta, float slope, float* bnbwd_partial, int B, int Ci,
                                             int Co, int H, int W, hipStream_t stream) {
  if (!bn_x) return SIVAE_ERR_NULL;
  return wino_fwd_impl(dy, up, y, nullptr, nullptr, nullptr, nullptr, nullptr, 1.f, bnbwd_partial, bn_x, bn_mean,
                       bn_invstd, bn_gamma, bn_beta, slope, B, Ci, Co, H, W, 0, 0, stream);
}

// ---- split-K variant for launches that would otherwise leave most of the chip idle (the 16-image shard of
// config 4 runs the 512-channel 4x4 / 8x8 layers as 32..64 work items, each a serial walk over 512 input channels:
// ~110 us per launch whatever the batch).  The K range is cut into S slices (S * items ~ one block per CU), every
// (item, slice) writes its partial output tensor, and one small kernel sums the slices in a fixed order (deterministic),
// adds the old y when accumulating, and leaves per-IMAGE {sum, sumsq} rows for the consumer BatchNorm.
__global__ void __launch_bounds__(64) wino_splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ y,
                                                                float* __restrict__ stats, int S, int C, int HW,
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

// The same sum for planes of 16 / 64 / 256 pixels (the 4x4 / 8x8 / 16x16 maps this kernel family serves) over 16-byte
// vectors: G = HW / 4 lanes per plane, 256 / G planes per block, all S slice loads of a vector in flight at once (the
// one-wave-per-plane form above is 8192..16384 blocks with 16 of 64 lanes busy: 7.2 us per call, 52 calls per iteration
// of the 16-image shard).  Slices are still added in slice order; a plane's {sum, sumsq} fold over its G lanes by xor
// shuffles (fixed tree).
template <int G>
__global__ void __launch_bounds__(256) wino_splitk_reduce_vec_kernel(const float4* __restrict__ part, float4* __restrict__ y,
                                                                     float* __restrict__ stats, int S, int n_planes,
                                                                     size_t slice_stride4, int accumulate) {
  const int t = (int)blockIdx.x * 256 + (int)threadIdx.x;
  const int bc = t / G;  // b * C + c
  const bool live = bc < n_planes;
  float s = 0.f, q = 0.f;
  if (live) {
    float4 tv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < S) tv[k] = part[(size_t)k * slice_stride4 + t];
    float4 v = accumulate ? y[t] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < S) {
        v.x += tv[k].x;
        v.y += tv[k].y;
        v.z += tv[k].z;
        v.w += tv[k].w;
      }
    y[t] = v;
    s = (v.x + v.y) + (v.z + v.w);
    q = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (stats != nullptr) {
#pragma unroll
    for (int o = G / 2; o >= 1; o >>= 1) {
      s += __shfl_xor(s, o, 64);
      q += __shfl_xor(q, o, 64);
    }
    if (live && (t % G) == 0) {
      stats[(size_t)bc * 2 + 0] = s;
      stats[(size_t)bc * 2 + 1] = q;
    }
  }
}

// number of K slices sivae_conv2d_wino_fwd_splitk will use (1: it is the plain kernel)
extern "C" int sivae_conv2d_wino_splitk(int B, int Ci, int Co, int H, int W) {
  if (B <= 0 || Ci <= 0 || Co <= 0 || !sivae_conv2d_wino_supported(H, W)) return SIVAE_ERR_SHAPE;
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("SIVAE_WINO_SPLITK");
    enabled = (e && e[0] == '0') ? 0 : 1;
  }
  if (!enabled) return 1;
  const long long items = (long long)sivae_conv2d_wino_num_px_tiles(B, H, W) * cdiv(Co, WINO_TCO);
  const int nchunks = wino_kpad(Ci) / WINO_CK;
  const int cus = wino_grid_blocks() / 2;
  if (items > cus || nchunks < 8) return 1;  // every CU has a block already, or a short K loop
  int S = (int)(2 * cus / items);             // aim at the persistent grid's two blocks per CU
  if (S > nchunks / 4) S = nchunks / 4;  // >= 4 chunks (64 input channels) per slice
  if (S > 8) S = 8;
  return S < 2 ? 1 : S;
}

extern "C" size_t sivae_conv2d_wino_splitk_workspace_bytes(int B, int Ci, int Co, int H, int W) {
  const int S = sivae_conv2d_wino_splitk(B, Ci, Co, H, W);
  if (S <= 1) return 0;
  return (size_t)S * B * Co * H * W * sizeof(float);
}

// stats_partial: [B][Co][2] (one row per image) when the call splits, sivae_conv2d_wino_num_px_tiles rows otherwise —
// sivae_conv2d_wino_splitk_stats_rows gives the count
extern "C" int sivae_conv2d_wino_splitk_stats_rows(int B, int Ci, int Co, int H, int W) {
  const int S = sivae_conv2d_wino_splitk(B, Ci, Co, H, W);
  if (S < 0) return S;
  return S > 1 ? B : sivae_conv2d_wino_num_px_tiles(B, H, W);
}

static int wino_fwd_splitk_impl(const float* x, const float* up, float* y, const float* pro_mean,
                                const float* pro_invstd, const float* pro_gamma, const float* pro_beta, float pro_slope,
                                float* stats_partial, int B, int Ci, int Co, int H, int W, int upsample, int accumulate,
                                void* workspace, size_t workspace_bytes, hipStream_t stream, int seg_images) {
  const int S = sivae_conv2d_wino_splitk(B, Ci, Co, H, W);
  if (S < 0) return S;
  if (S == 1)
    return wino_fwd_impl(x, up, y, nullptr, pro_mean, pro_invstd, pro_gamma, pro_beta, pro_slope, stats_partial, nullptr,
                         nullptr, nullptr, nullptr, nullptr, 1.f, B, Ci, Co, H, W, upsample, accumulate, stream, 0, 0,
                         seg_images);
  if (!y || !workspace) return SIVAE_ERR_NULL;
  if (workspace_bytes < sivae_conv2d_wino_splitk_workspace_bytes(B, Ci, Co, H, W)) return SIVAE_ERR_WORKSPACE;
  if (((uintptr_t)workspace & 7u) != 0) return SIVAE_ERR_SHAPE;
  const int nchunks = wino_kpad(Ci) / WINO_CK;
  const int cps = cdiv(nchunks, S);
  const long long stride = (long long)B * Co * H * W;
  float* part = reinterpret_cast<float*>(workspace);
  const int rc = wino_fwd_impl(x, up, part, nullptr, pro_mean, pro_invstd, pro_gamma, pro_beta, pro_slope, nullptr,
                               nullptr, nullptr, nullptr, nullptr, nullptr, 1.f, B, Ci, Co, H, W, upsample, 0, stream,
                               cps, stride, seg_images);
  if (rc != SIVAE_OK) return rc;
  const int Sx = cdiv(nchunks, cps), HW = H * W, planes = B * Co;
  const bool vec = Sx <= 8 && (HW == 16 || HW == 64 || HW == 256) && ((uintptr_t)y & 15u) == 0 &&
                   ((uintptr_t)workspace & 15u) == 0 && (long long)planes * (HW / 4) < 0x7fffffffLL;
  if (vec) {
    const float4* p4 = reinterpret_cast<const float4*>(part);
    float4* y4 = reinterpret_cast<float4*>(y);
    const unsigned nb = (unsigned)(((long long)planes * (HW / 4) + 255) / 256);
    if (HW == 16)
      hipLaunchKernelGGL(wino_splitk_reduce_vec_kernel<4>, dim3(nb), dim3(256), 0, stream, p4, y4, stats_partial, Sx, planes,
                         (size_t)stride / 4, accumulate);
    else if (HW == 64)
      hipLaunchKernelGGL(wino_splitk_reduce_vec_kernel<16>, dim3(nb), dim3(256), 0, stream, p4, y4, stats_partial, Sx, planes,
                         (size_t)stride / 4, accumulate);
    else
      hipLaunchKernelGGL(wino_splitk_reduce_vec_kernel<64>, dim3(nb), dim3(256), 0, stream, p4, y4, stats_partial, Sx, planes,
                         (size_t)stride / 4, accumulate);
  } else {
    hipLaunchKernelGGL(wino_splitk_reduce_kernel, dim3((unsigned)planes), dim3(64), 0, stream, part, y, stats_partial, Sx, Co,
                       HW, (size_t)stride, accumulate);
  }
  return sivae_launch_status();
}

extern "C" int sivae_conv2d_wino_fwd_splitk(const float* x, const float* up, float* y, const float* pro_mean,
                                            const float* pro_invstd, const float* pro_gamma, const float* pro_beta,
                                            float pro_slope, float* stats_partial, int B, int Ci, int Co, int H, int W,
                                            int upsample, int accumulate, void* workspace, size_t workspace_bytes,
                                            hipStream_t stream) {
  return wino_fwd_splitk_impl(x, up, y, pro_mean, pro_invstd, pro_gamma, pro_beta, pro_slope, stats_partial, B, Ci, Co, H,
                              W, upsample, accumulate, workspace, workspace_bytes, stream, 0);
}

// segmented batch (see sivae_conv2d_wino_fwd_seg); statistics rows are per image when the call splits
extern "C" int sivae_conv2d_wino_fwd_splitk_seg(const float* x, const float* up, float* y, const float* pro_mean,
                                                const float* pro_invstd, const float* pro_gamma, const float* pro_beta,
                                                float pro_slope, float* stats_partial, int B, int Ci, int Co, int H,
                                                int W, int upsample, int accumulate, int seg_images, void* workspace,
                                                size_t workspace_bytes, hipStream_t stream) {
  if (seg_images <= 0) return SIVAE_ERR_SHAPE;
  return wino_fwd_splitk_impl(x, up, y, pro_mean, pro_invstd, pro_gamma, pro_beta, pro_slope, stats_partial, B, Ci, Co, H,
                              W, upsample, accumulate, workspace, workspace_bytes, stream, seg_images);
}

// ---- batched packing (pack_batch.h)
int sivae_packjob_wino(SivaePackJob* j, int Co, int Ci, int mode) {
  j->kdim = mode == 0 ? Ci : Co;
  j->ndim = mode == 0 ? Co : Ci;
  j->kpad = wino_kpad(j->kdim);
  j->npad = wino_npad(j->ndim);
  j->total = (unsigned long long)j->kpad * j->npad;
  return SIVAE_OK;
}
void sivae_packbatch_wino(const SivaePackJob* jobs, const unsigned short* block_job, int nblocks, hipStream_t s) {
  hipLaunchKernelGGL(pack_wino_batch_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, jobs, block_job);
}
