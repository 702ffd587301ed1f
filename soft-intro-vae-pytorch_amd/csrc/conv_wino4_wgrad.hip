// Winograd F(4x4,3x3) weight gradient of the 3x3 stride-1 "same" convolution (fp32 MFMA, gfx950).
//
// With Y = A^T [ (G g G^T) . (B^T d B) ] A per 4x4 output tile (conv_wino4.hip), the filter gradient is linear in the
// same transformed operands (the weight half of aten::convolution_backward of the nn.Conv2d(k=3) layers,
// soft_intro_vae/train_soft_intro_vae.py:56-61):
//
//   dU[i][j][co][ci] = sum over tiles  Mg[i][j][co][tile] * V[i][j][ci][tile]     Mg = A dY A^T  (4x4 -> 6x6)
//   dg[co][ci]       = G^T dU[.][.][co][ci] G                                     V  = B^T d  B   (6x6 -> 6x6)
//
// 36 independent GEMMs with K = number of tiles: 36 multiplies per (tile, co, ci) instead of the 144 of the direct form
// (9 taps x 16 pixels) and the 64 of the F(2x2,3x3) form (conv_wino_wgrad.hip).
//
// fp32 "MFMA" time and ordinary VALU time ADD on a gfx950 SIMD (conv_wino4.hip), so both operands are transformed ONCE
// per block into LDS and the MFMA lanes read them with plain ds_read_b32:
//   * a block = 12 waves = 64 output channels x 32 input channels x 36 frequencies; wave (j, s) owns frequency column j
//     of output-channel subtile s: 6 accumulators = 96 registers, three waves per SIMD, one block per CU;
//   * a stage = a 4 x 16 pixel strip = 4 tiles = two k-steps (a k-step is a pair of tiles).  Its raw operands — the
//     6 x 24 input halo of 32 channels (18 KB) and the 4 x 16 dY strip of 64 channels (16 KB) — arrive by LDS-direct
//     16-byte loads (three per wave and stage) into a ring of three slots, requested three stages ahead.  The slot
//     layouts are chosen for the 16-byte reads of the transform: dY [row][co][group ^ ((co >> 2) & 3)], x [row][ci][(group +
//     ((ci >> 3) & 1)) % 6] — conflict-free for the lane groups a ds_read_b128 serves per cycle, and the four / six
//     groups of a row segment still come from adjacent lanes of a request;
//   * every k-step transforms the two tiles of the NEXT k-step into the other half of the double-buffered V[36][2][32] /
//     Mg[36][2][64]: nine wave-tasks of a third of an item each (thread = tile x channel x frequency-column pair, the
//     pairing of conv_wino4.hip), spread over the SIMDs and sliced between the wave's own six MFMAs.  A long task on
//     one wave does not overlap anything: the other waves' MFMAs starve its VALU issue, then it runs alone (first form:
//     phase-serial, 0.85-1.09x the F(2x2,3x3) kernel; this form 1.35-1.55x);
//   * the MFMA schedule is rotated by one pair across the barrier, so that a k-step opens with MFMAs whose operands are
//     already in registers (its own LDS reads in flight) and closes with MFMAs under which its LDS stores drain;
//   * with the fused BatchNorm + LeakyReLU prologue, k-step A rewrites groups 0-3 and k-step B groups 4-5 of the NEXT
//     stage's raw halo in place (zero padding restored through the table select).
// Two LDS-only barriers per stage.  The three task roles (x, x pair (0,5), dY) and the task-less waves run separate copies
// of the loop.  The sum over tiles is split across blocks ("slices") of whole stage triples; every slice writes its
// partial — already row-transformed in registers, G^T dU: 18 values per (co, ci) instead of 36 —, the slices are added in a
// fixed order (wide split-K reducer from 8 slices up) and a last kernel applies the column half . G (no atomics,
// run-to-run reproducible).
#include "common.h"
#include <stdlib.h>

struct Wino4WgArgs {
  const float* x;
  const float* dy;
  float* ws;  // [n_slices][3][6][Co_pad][Ci_pad]: row-transformed partials G^T dU
  const float* pro_mean;
  const float* pro_invstd;
  const float* pro_gamma;
  const float* pro_beta;
  float pro_slope;
  int pro_seg_images, pro_nseg;  // segments (conv_wino.hip): pro_mean / pro_invstd are [pro_nseg][Ci], pro_nseg <= 2
  int B, Ci, Co, H, W;
  int Ci_pad, Co_pad;
  int nrh, nrw, nstages, sps;
  int n_co_tiles, n_ci_tiles;
  int nblk, xcd_remap;
  // Maps narrower than a stage's 16 columns (8 x 8, 4 x 4: round 6): the 4 x 16 strip is `ips` = 16 / W whole images side by
  // side (nrw = 1, so both column-border flags are set and halo groups 0 / 5 read as zeros); the row halo stays inside
  // one image (a strip is 4 rows of it).  Patch column 0 / 5 of a tile at the left / right edge of its image is zero padding:
  // 64-bit lane masks for the (0,5) transform role — lane -> tile = (lane >> 5) [+ 2 in k-step A], per k-step.
  int ips, iw_l2;
  unsigned long long seam_e0B, seam_e5B, seam_e0A, seam_e5A;
};

#define G4_NT 768
#define G4_XS 4608  // raw x slot: [6 rows][32 ci][6 groups of 4 floats]
#define G4_YS 4096  // raw dY slot: [4 rows][64 co][4 groups, swizzled by (co >> 2) & 3]
#define G4_VS 2304  // V of a k-step: [36][2 tiles][32 ci]
#define G4_MS 4608  // Mg of a k-step: [36][2 tiles][64 co]

// Timing ablations (results WRONG with any bit set): 1 no LDS-direct loads, 2 no transform phase, 4 no MFMAs, 16 no prologue
#ifndef G4_ABLATE
#define G4_ABLATE 0
#endif

// GRID: the image-strip mode of the 8 x 8 / 4 x 4 maps (its own instantiations: the large-map kernels carry none of it)
template <bool PRO, bool GRID = false>
__global__ void __launch_bounds__(G4_NT, 1) wino4_wgrad_kernel(Wino4WgArgs a) {
  // every ring slot is its own static array: the compiler orders an LDS access behind each in-flight LDS-direct load
  // it cannot prove disjoint (conv_wino4.hip)
  __shared__ __attribute__((aligned(16))) float rx0[G4_XS];
  __shared__ __attribute__((aligned(16))) float rx1[G4_XS];
  __shared__ __attribute__((aligned(16))) float rx2[G4_XS];
  __shared__ __attribute__((aligned(16))) float ry0[G4_YS];
  __shared__ __attribute__((aligned(16))) float ry1[G4_YS];
  __shared__ __attribute__((aligned(16))) float ry2[G4_YS];
  __shared__ __attribute__((aligned(16))) float vs0[G4_VS];  // V / Mg of a k-step (two tiles), double-buffered
  __shared__ __attribute__((aligned(16))) float vs1[G4_VS];
  __shared__ __attribute__((aligned(16))) float ms0[G4_MS];
  __shared__ __attribute__((aligned(16))) float ms1[G4_MS];
  // {scale, shift} = {invstd*gamma, beta - mean*scale} per (segment, channel of the tile): x' = lrelu(x*scale + shift)
  __shared__ float2 pro2[PRO ? 64 : 1];
#define G4_RX(K) ((K) == 0 ? rx0 : ((K) == 1 ? rx1 : rx2))
#define G4_RY(K) ((K) == 0 ? ry0 : ((K) == 1 ? ry1 : ry2))
#define G4_VB(HB) ((HB) ? vs1 : vs0)
#define G4_MB(HB) ((HB) ? ms1 : ms0)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int wj = wave % 6, wsb = wave / 6;
  const int H = a.H, W = a.W, HW = H * W;

  const int ntiles = a.n_co_tiles * a.n_ci_tiles;
  // XCD-aware block order (conv_wino_wgrad.hip): the (co, ci) tiles of one slice read the same stages; put them on one XCD
  int lb = (int)blockIdx.x;
  if (a.xcd_remap) {
    lb = (lb & 7) * ((int)gridDim.x >> 3) + (lb >> 3);
    if (lb >= a.nblk) return;
  }
  const int tile = lb % ntiles, slice = lb / ntiles;
  const int ci0 = (tile % a.n_ci_tiles) * 32, co0 = (tile / a.n_ci_tiles) * 64;
  const int s_begin = slice * a.sps;
  const int s_end = (s_begin + a.sps < a.nstages) ? (s_begin + a.sps) : a.nstages;

  // ---- request role: wave w issues pieces 3w .. 3w+2 of a stage; pieces 0..17 are the x slot (waves 0-5), 18..33 the dY
  // slot (waves 6-11; the last two repeat piece 33).  A piece = 64 lanes x 16 bytes = 1 KB of consecutive LDS.
  const bool req_x = wave < 6;
  // (an x wave's three pieces are exactly halo row `wave`: the row bits are scalar; per lane only "group 0" / "group 5")
  const unsigned rq_sbits = 16u | (wave == 0 ? 1u : 0u) | (wave == 5 ? 2u : 0u);
  unsigned rq_off[3], rq_lbits = 0;  // rq_lbits: bits 4 / 8 of piece i at byte i
  int rq_lds[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int d = wave * 3 + i;
    if (d < 18) {
      const int pos = d * 64 + lane, row = pos / 192, rem = pos - row * 192, ci = rem / 6;
      const int p = (rem - ci * 6 + 6 - ((ci >> 3) & 1)) % 6;  // (the slot holds group p rotated by r(ci), below)
      // (GRID: group p = 1..4 is column 4 (p - 1) of the strip -> image (4 (p - 1)) >> iw_l2 of the strip at its own column;
      // the stage base points 4 columns left of the strip, hence the + 4; groups 0 / 5 are always masked: nrw = 1)
      const int gq = (p >= 1 && p <= 4) ? 4 * (p - 1) : 0;
      const int gcol = GRID ? (gq >> a.iw_l2) * a.Ci * HW + 4 + (gq & (W - 1)) : 4 * p;
      rq_off[i] = (ci0 + ci < a.Ci) ? (unsigned)((ci0 + ci) * HW + row * W + gcol) * 4u : SIVAE_OOB;
      rq_lbits |= ((p == 0 ? 4u : 0u) | (p == 5 ? 8u : 0u)) << (8 * i);
      rq_lds[i] = d * 256;
    } else {
      const int dd = d - 18 < 15 ? d - 18 : 15;
      const int pos = dd * 64 + lane, row = pos >> 8, co = (pos & 255) >> 2, p = (pos & 3) ^ ((co >> 2) & 3);
      const int gcol = GRID ? ((4 * p) >> a.iw_l2) * a.Co * HW + ((4 * p) & (W - 1)) : 4 * p;
      rq_off[i] = (co0 + co < a.Co) ? (unsigned)((co0 + co) * HW + row * W + gcol) * 4u : SIVAE_OOB;
      rq_lds[i] = dd * 256;
    }
  }
  // ---- prologue role: k-step A of stage s rewrites the whole raw x slot of stage s+1 in place — 16-byte group tid (rows 0-3;
  // a wave covers 64 consecutive groups of ONE row: row = wave / 3) and, for the first six waves, group 768 + tid (rows 4, 5:
  // the same channel / group-of-the-row, so the table entry and the left / right test are shared).  Per lane one register:
  // channel << 3 | (true group == 0) | (true group == 5) << 1; the row part of everything is wave-uniform.
  unsigned fx_lc;
  {
    const int rem = (wave % 3) * 64 + lane, ci = rem / 6, pp = rem - ci * 6, p = (pp + 6 - ((ci >> 3) & 1)) % 6;
    fx_lc = ((unsigned)ci << 3) | (p == 0 ? 1u : 0u) | (p == 5 ? 2u : 0u);
  }
  const int fx_s0 = ((wave / 3) * 192 + (wave % 3) * 64) * 4;  // floats; the second group: + 768 * 4
  const unsigned fx_sb0 = 16u | (wave < 3 ? 1u : 0u);           // row 0 <-> top border
  const unsigned fx_sb1 = 16u | (wave >= 3 ? 2u : 0u);          // (waves 0-5) row 4 + wave / 3 == 5 <-> bottom border
  // ---- transform roles.  Every k-step transforms the two tiles of the NEXT k-step (k-step A, MFMAs on tiles 0,1: tiles
  // 2,3 of the same stage; k-step B, MFMAs on tiles 2,3: tiles 0,1 of the next stage) in nine wave-tasks, each a THIRD of an
  // item — the frequency columns come in pairs that share their partial sums ((1,2), (3,4); (0,5) stand alone):
  //   x  -> V : thread (tile = hh, ci = l31, pair): waves 0, 1, 2                     (48 VALU, 12 ds_write_b32 per thread)
  //   dY -> Mg: thread (tile tl, co = lane, pair): waves 4,5 (1,2); 3,7 (3,4); 6,10 (0,5)   (32 / 32 / 16 VALU, 12 writes)
  // spread so that the three waves of a SIMD (w, w+4, w+8) carry ~80 VALU together; a wave slices its task between its
  // own six MFMAs (a long task on one wave would run alone after the other waves' MFMAs: the first form of this loop).
  const bool r_x = wave < 3;
  const bool r_y = wave == 4 || wave == 5 || wave == 3 || wave == 7 || wave == 6 || wave == 10;
  const int tp = r_x ? wave : ((wave == 4 || wave == 5) ? 0 : ((wave == 3 || wave == 7) ? 1 : 2));
  const int ytl = (wave == 5 || wave == 7 || wave == 10) ? 1 : 0;
  const int jA = tp == 0 ? 1 : (tp == 1 ? 3 : 0), jB = tp == 0 ? 2 : (tp == 1 ? 4 : 5);
  // x: patch columns 1..4 = 16-byte group tile + 1 (k-step B: tiles 0,1; k-step A: tiles 2,3 = + 8 floats)
  // The 16-byte groups of a (row, channel) are stored ROTATED by r(ci) = (ci >> 3) & 1 (slot ci*6 + (p + r) % 6): with the
  // plain stride of 6 slots the 16 lanes a ds_read_b128 serves per cycle hit 8 bank groups twice each
  const int x_rot = (l31 >> 3) & 1;
  const int tx_rd4 = l31 * 6 + hh + 1 + x_rot;
  // patch columns 0 / 5 of tile t: last float of group t, first float of group t + 2 (k-step B: t = hh; A: t = hh + 2)
  const int tx_e0B = l31 * 6 + hh + x_rot, tx_e5B = l31 * 6 + (hh + 2 + x_rot) % 6;
  const int tx_e0A = l31 * 6 + hh + 2 + x_rot, tx_e5A = l31 * 6 + (hh + 4 + x_rot) % 6;
  const float t_al = tp == 0 ? -4.f : -1.f;  // a = d4 + al*d2
  const float t_be = tp == 0 ? 1.f : 2.f;    // b = be*d3 + ga*d1
  const float t_ga = tp == 0 ? -4.f : -2.f;
  // dY: group (tile ^ swizzle) of channel row lane (k-step A: tile + 2 = the group index ^ 2 = + / - 8 floats)
  const int ty_rd4 = lane * 4 + (ytl ^ ((lane >> 2) & 3));
  const float y_al = tp == 0 ? 1.f : 4.f;  // e = c0 + al*c2, o = c1 + al*c3; outputs e +- be*o
  const float y_be = tp == 0 ? 1.f : 2.f;
  const int t_wr = r_x ? lane : ytl * 64 + lane;
  // ---- MFMA role: A = Mg[(i*6 + wj)][hh][wsb*32 + l31], B = V[(i*6 + wj)][hh][l31] of the k-step's half buffers
  const int m_rd = wj * 128 + hh * 64 + wsb * 32 + l31;
  const int v_rd = wj * 64 + lane;

  // ---- stage to request next (scalar): image db, strip row dry, strip column drx
  int sd = s_begin;
  const int ips = GRID ? a.ips : 1;  // images per strip
  int db = sd / (a.nrh * a.nrw);     // strip set -> (below) its first image
  int dry, drx;
  {
    const int rem = sd - db * (a.nrh * a.nrw);
    dry = rem / a.nrw;
    drx = rem - dry * a.nrw;
  }
  db *= ips;
  // this wave's operand (x for the first six waves, dY for the others) at image db
  const long long rq_img_step = (long long)(req_x ? a.Ci : a.Co) * HW * ips;
  const float* rq_img = (req_x ? a.x : a.dy) + (long long)db * ((long long)(req_x ? a.Ci : a.Co) * HW);
  unsigned inf0 = 0, inf1 = 0, inf2 = 0;  // per ring slot: border / tail bits and the segment's table offset << 8
#define G4_INF(K) ((K) == 0 ? inf0 : ((K) == 1 ? inf1 : inf2))
#define G4_SETINF(K, V) { if ((K) == 0) inf0 = (V); else if ((K) == 1) inf1 = (V); else inf2 = (V); }

  // request stage sd into ring slot K (three unconditional LDS-direct loads per wave; invalid groups read as zeros)
#define G4_REQ(K)                                                                                   \
  {                                                                                                 \
    const unsigned fl_ = (dry == 0 ? 1u : 0u) | (dry == a.nrh - 1 ? 2u : 0u) | (drx == 0 ? 4u : 0u) |  \
                         (drx == a.nrw - 1 ? 8u : 0u) | (sd >= s_end ? 16u : 0u);                   \
    const float* base_ = rq_img + (req_x ? (dry * 4 - 1) * W + (drx * 16 - 4) : dry * 4 * W + drx * 16); \
    const __amdgpu_buffer_rsrc_t rs_ = make_rsrc(base_, 0xFFFFFFFEull);                             \
    float* lds_ = req_x ? G4_RX(K) : G4_RY(K);                                                      \
    unsigned o0_ = rq_off[0], o1_ = rq_off[1], o2_ = rq_off[2];                                     \
    if (fl_) {                                                                                      \
      if ((req_x ? rq_sbits : 16u) & fl_) {                                                         \
        o0_ = o1_ = o2_ = SIVAE_OOB;                                                                \
      } else if (req_x) {                                                                           \
        const unsigned lm_ = fl_ & 12u;                                                             \
        o0_ = (rq_lbits & lm_) ? SIVAE_OOB : o0_;                                                   \
        o1_ = (rq_lbits & (lm_ << 8)) ? SIVAE_OOB : o1_;                                            \
        o2_ = (rq_lbits & (lm_ << 16)) ? SIVAE_OOB : o2_;                                           \
      }                                                                                             \
    }                                                                                               \
    if (!(G4_ABLATE & 1)) {                                                                         \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (float __attribute__((address_space(3)))*)(lds_ + rq_lds[0]), 16, o0_, 0, 0, 0); \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (float __attribute__((address_space(3)))*)(lds_ + rq_lds[1]), 16, o1_, 0, 0, 0); \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (float __attribute__((address_space(3)))*)(lds_ + rq_lds[2]), 16, o2_, 0, 0, 0); \
    }                                                                                               \
    if (PRO) G4_SETINF(K, fl_ | ((db >= a.pro_seg_images ? 32u : 0u) << 8))                         \
    ++sd;                                                                                           \
    if (++drx == a.nrw) {                                                                           \
      drx = 0;                                                                                      \
      if (++dry == a.nrh) {                                                                         \
        dry = 0;                                                                                    \
        db += ips;                                                                                  \
        rq_img += rq_img_step;                                                                      \
      }                                                                                             \
    }                                                                                               \
  }
  // fused BatchNorm + LeakyReLU prologue, in place on the raw x slot K: x' = max(v, slope v), v = (x - mean) scale + beta;
  // groups outside the image (and channels beyond Ci: zero table rows) stay zero
  // x' = max(v, slope v), v = x scale + shift (two pixels per packed-fp32 instruction); groups outside the image and
  // channels beyond Ci (zero table rows) stay zero
#define G4_FIX_ONE(QP, SC, SH)                                                                      \
  {                                                                                                 \
    const float4 v_ = *(QP);                                                                        \
    const f32x2 sc_ = {SC, SC}, sh_ = {SH, SH}, sl_ = {a.pro_slope, a.pro_slope};                   \
    f32x2 lo_ = {v_.x, v_.y}, hi_ = {v_.z, v_.w};                                                   \
    lo_ = __builtin_elementwise_fma(lo_, sc_, sh_);                                                 \
    hi_ = __builtin_elementwise_fma(hi_, sc_, sh_);                                                 \
    const f32x2 ls_ = lo_ * sl_, hs_ = hi_ * sl_;                                                   \
    *(QP) = make_float4(fmaxf(lo_[0], ls_[0]), fmaxf(lo_[1], ls_[1]), fmaxf(hi_[0], hs_[0]), fmaxf(hi_[1], hs_[1])); \
  }
  // (the lane offset is recomputed from a LAUNDERED lane index at every use: hoisted out of the loop such constants are
  // spilled, and a spill reload inside the loop carries an s_waitcnt vmcnt(0) — it would wait out the requests in flight)
#define G4_FIX(K)                                                                                   \
  if (!(G4_ABLATE & 16)) {                                                                          \
    const unsigned in_ = G4_INF(K);                                                                 \
    int ln_ = lane;                                                                                 \
    asm volatile("" : "+v"(ln_));                                                                   \
    const float2 p_ = pro2[(in_ >> 8) + (fx_lc >> 3)];                                              \
    const bool lout_ = (fx_lc & (in_ >> 2) & 3u) != 0u;                                             \
    float4* q_ = reinterpret_cast<float4*>(G4_RX(K) + fx_s0) + ln_;                                 \
    {                                                                                               \
      const bool out_ = lout_ || (fx_sb0 & in_ & 31u) != 0u;                                        \
      G4_FIX_ONE(q_, out_ ? 0.f : p_.x, out_ ? 0.f : p_.y)                                          \
    }                                                                                               \
    if (wave < 6) {                                                                                 \
      const bool out_ = lout_ || (fx_sb1 & in_ & 31u) != 0u;                                        \
      G4_FIX_ONE(q_ + 768, out_ ? 0.f : p_.x, out_ ? 0.f : p_.y)                                    \
    }                                                                                               \
  }
#define G4_FENCE __builtin_amdgcn_sched_barrier(0);
  // workgroup barrier that orders LDS traffic only (a __syncthreads() with LDS-direct loads in flight waits vmcnt(0))
#define G4_LDS_BARRIER asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // A (4 -> 6) of one vector: {d0, e+o, e-o, e'+2o', e'-2o', d3}, e = d0+d2, o = d1+d3, e' = d0+4d2, o' = d1+4d3
#define G4_A4(D0, D1, D2, D3, M)                                                                    \
  {                                                                                                 \
    const float ae_ = (D0) + (D2), ao_ = (D1) + (D3);                                               \
    const float ae2_ = fmaf(4.f, (D2), (D0)), ao2_ = fmaf(4.f, (D3), (D1));                         \
    M[0] = (D0);                                                                                    \
    M[1] = ae_ + ao_;                                                                               \
    M[2] = ae_ - ao_;                                                                               \
    M[3] = fmaf(2.f, ao2_, ae2_);                                                                   \
    M[4] = fmaf(-2.f, ao2_, ae2_);                                                                  \
    M[5] = (D3);                                                                                    \
  }
  // ---- task slices; ROLE (compile-time): 0 no task, 1 x -> V, 2 dY -> Mg.  The three roles run three COPIES of the whole
  // loop (branching once per wave): with role branches inside a k-step the compiler carries the other roles' undefined
  // registers through every join and spills hundreds of them.
  // Registers: rd_ raw rows, ex_ patch columns 0 / 5 (x), tA_ / tB_ the pair's column-direction results per row.
  // slice R: raw reads of this wave's task out of ring slot K (XOFF: + 8 floats, YXOR: ^ 8 floats for tiles 2,3)
#define G4_T_READ(ROLE, K, XOFF, YXOR, HALF, RD, EX)                                                \
  if (!(G4_ABLATE & 2)) {                                                                           \
    if ((ROLE) == 1 || (ROLE) == 3) { /* three patch rows per call; 16-byte reads (float4-typed: ds_read_b128) */ \
      const float4* p4_ = reinterpret_cast<const float4*>(G4_RX(K)) + (tx_rd4 + (XOFF) / 4 + (HALF) * 3 * 192); \
      _Pragma("unroll") for (int r = 0; r < 3; ++r) RD[r] = p4_[r * 192];                           \
      if ((ROLE) == 3) { /* whole neighbour groups (conflict-free 16-byte reads; a dword read here is 8-way) */ \
        const float4* e0_ = reinterpret_cast<const float4*>(G4_RX(K)) + ((XOFF) ? tx_e0A : tx_e0B) + (HALF) * 3 * 192; \
        const float4* e5_ = reinterpret_cast<const float4*>(G4_RX(K)) + ((XOFF) ? tx_e5A : tx_e5B) + (HALF) * 3 * 192; \
        _Pragma("unroll") for (int r = 0; r < 3; ++r) {                                             \
          EX[r] = e0_[r * 192].w;                                                                   \
          EX[3 + r] = e5_[r * 192].x;                                                               \
          if (GRID) { /* the neighbour column belongs to the next image of the strip: zero padding */ \
            asm volatile("v_cndmask_b32_e64 %0, %0, 0, %1" : "+v"(EX[r]) : "s"((XOFF) ? a.seam_e0A : a.seam_e0B));     \
            asm volatile("v_cndmask_b32_e64 %0, %0, 0, %1" : "+v"(EX[3 + r]) : "s"((XOFF) ? a.seam_e5A : a.seam_e5B)); \
          }                                                                                         \
        }                                                                                           \
      }                                                                                             \
    } else if ((ROLE) == 2 && (HALF) == 0) {                                                        \
      const float4* p4_ = reinterpret_cast<const float4*>(G4_RY(K)) + (ty_rd4 ^ ((YXOR) / 4));      \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) RD[r] = p4_[r * 256];                           \
    }                                                                                               \
  }
  // slice 0: the column direction (per raw row, the pair's two results); x: rows 3*HALF .. 3*HALF+2
#define G4_T_COL(ROLE, HALF, RD, EX)                                                                \
  if (!(G4_ABLATE & 2)) {                                                                           \
    if ((ROLE) == 3) {                                                                              \
      _Pragma("unroll") for (int r = 0; r < 3; ++r) {                                               \
        tA_[3 * (HALF) + r] = fmaf(4.f, EX[r], fmaf(-5.f, RD[r].y, RD[r].w));                       \
        tB_[3 * (HALF) + r] = fmaf(4.f, RD[r].x, fmaf(-5.f, RD[r].z, EX[3 + r]));                   \
      }                                                                                             \
    } else if ((ROLE) == 1) {                                                                       \
      _Pragma("unroll") for (int r = 0; r < 3; ++r) {                                               \
        const float a_ = fmaf(t_al, RD[r].y, RD[r].w);                                              \
        const float b_ = fmaf(t_ga, RD[r].x, t_be * RD[r].z);                                       \
        tA_[3 * (HALF) + r] = a_ + b_;                                                              \
        tB_[3 * (HALF) + r] = a_ - b_;                                                              \
      }                                                                                             \
    } else if ((ROLE) == 2 && (HALF) == 0) {                                                        \
      if (tp == 2) {                                                                                \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                             \
          tA_[r] = RD[r].x;                                                                         \
          tB_[r] = RD[r].w;                                                                         \
        }                                                                                           \
      } else {                                                                                      \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                             \
          const float e_ = fmaf(y_al, RD[r].z, RD[r].x), o_ = fmaf(y_al, RD[r].w, RD[r].y);         \
          tA_[r] = fmaf(y_be, o_, e_);                                                              \
          tB_[r] = fmaf(-y_be, o_, e_);                                                             \
        }                                                                                           \
      }                                                                                             \
    }                                                                                               \
  }
  // slices 1, 2: the row direction of one column of the pair -> half buffer HB
#define G4_T_ROW(ROLE, HB, T, J)                                                                    \
  if (!(G4_ABLATE & 2)) {                                                                           \
    if ((ROLE) == 1 || (ROLE) == 3) {                                                               \
      const float A_ = fmaf(-4.f, T[2], T[4]), B_ = fmaf(-4.f, T[1], T[3]);                         \
      const float C_ = T[4] - T[2], D_ = T[3] - T[1];                                               \
      float* q_ = G4_VB(HB) + (J) * 64 + t_wr;                                                      \
      q_[0 * 384] = fmaf(4.f, T[0], fmaf(-5.f, T[2], T[4]));                                        \
      q_[1 * 384] = A_ + B_;                                                                        \
      q_[2 * 384] = A_ - B_;                                                                        \
      q_[3 * 384] = fmaf(2.f, D_, C_);                                                              \
      q_[4 * 384] = fmaf(-2.f, D_, C_);                                                             \
      q_[5 * 384] = fmaf(4.f, T[1], fmaf(-5.f, T[3], T[5]));                                        \
    } else if ((ROLE) == 2) {                                                                       \
      float o_[6];                                                                                  \
      G4_A4(T[0], T[1], T[2], T[3], o_)                                                             \
      float* q_ = G4_MB(HB) + (J) * 128 + t_wr;                                                     \
      _Pragma("unroll") for (int i = 0; i < 6; ++i) q_[i * 768] = o_[i];                            \
    }                                                                                               \
  }
#define G4_READ2(HB, I0, AV, BV)                                                                    \
  {                                                                                                 \
    const float* pa_ = G4_MB(HB) + m_rd;                                                            \
    const float* pb_ = G4_VB(HB) + v_rd;                                                            \
    AV[0] = pa_[(I0) * 768];                                                                        \
    AV[1] = pa_[((I0) + 1) * 768];                                                                  \
    BV[0] = pb_[(I0) * 384];                                                                        \
    BV[1] = pb_[((I0) + 1) * 384];                                                                  \
  }
#define G4_MMA2(I0, AV, BV)                                                                         \
  if (!(G4_ABLATE & 4)) {                                                                           \
    acc[I0] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[0], BV[0], acc[I0], 0, 0, 0);                 \
    acc[(I0) + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[1], BV[1], acc[(I0) + 1], 0, 0, 0);     \
  }
  // One k-step on half buffer HB; this wave's task (raw slot TK, tile offsets XOFF / YXOR) fills half buffer HB ^ 1.
  // The MFMA schedule is ROTATED by one pair: the k-step opens — right behind the barrier, while its own LDS reads are in
  // flight — with the previous k-step's last pair (operands a2_/b2_ carried in registers), and ends with its pair (2,3),
  // under which the task's LDS stores drain before the closing barrier.
#define G4_KSTEP(ROLE, HB, TK, XOFF, YXOR, REQCODE, FIXCODE)                                        \
  {                                                                                                 \
    float4 rd_[4], rd2_[3];                                                                         \
    float ex_[6], ex2_[6], tA_[6], tB_[6];                                                          \
    float a0_[2], b0_[2], a1_[2], b1_[2];                                                           \
    G4_T_READ(ROLE, TK, XOFF, YXOR, 0, rd_, ex_)                                                    \
    G4_T_READ(ROLE, TK, XOFF, YXOR, 1, rd2_, ex2_)                                                  \
    G4_READ2(HB, 0, a0_, b0_)                                                                       \
    G4_FENCE                                                                                        \
    G4_MMA2(4, a2_, b2_)                                                                            \
    G4_FENCE                                                                                        \
    REQCODE /* (scalar address arithmetic: under the two MFMAs just issued, not in front of them) */ \
    G4_READ2(HB, 2, a1_, b1_)                                                                       \
    G4_T_COL(ROLE, 0, rd_, ex_)                                                                     \
    G4_FENCE                                                                                        \
    G4_MMA2(0, a0_, b0_)                                                                            \
    G4_FENCE                                                                                        \
    G4_READ2(HB, 4, a2_, b2_)                                                                       \
    G4_T_COL(ROLE, 1, rd2_, ex2_)                                                                   \
    G4_T_ROW(ROLE, (HB) ^ 1, tA_, jA)                                                               \
    G4_T_ROW(ROLE, (HB) ^ 1, tB_, jB)                                                               \
    FIXCODE                                                                                         \
    G4_FENCE                                                                                        \
    G4_MMA2(2, a1_, b1_)                                                                            \
    G4_FENCE                                                                                        \
  }
  // One stage s in ring slot K = two k-steps.  On entry: half buffer 0 holds the transformed tiles 0,1 of stage s; slot K
  // has landed and is already rewritten by the prologue; slot K+1 (stage s+1) has landed; stage s+2 is in flight.
  //   k-step A: MFMAs on half 0  ||  tiles 2,3 of slot K -> half 1  ||  prologue on the whole slot K+1
  //   k-step B: request stage s+3 into slot K  ||  MFMAs on half 1  ||  tiles 0,1 of slot K+1 -> half 0
#define G4_STAGE(ROLE, K)                                                                           \
  {                                                                                                 \
    G4_KSTEP(ROLE, 0, K, 8, 8, , if (PRO) G4_FIX(((K) + 1) % 3))                                    \
    G4_LDS_BARRIER                                                                                  \
    G4_FENCE                                                                                        \
    G4_KSTEP(ROLE, 1, ((K) + 1) % 3, 0, 0, G4_REQ(K), )                                             \
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                                                \
    G4_LDS_BARRIER                                                                                  \
    G4_FENCE                                                                                        \
  }
  // the whole K loop of one role (every copy executes the same barriers)
#define G4_LOOP(ROLE)                                                                               \
  {                                                                                                 \
    { /* tiles 0,1 of the first stage -> half buffer 0 */                                           \
      float4 rd_[4], rd2_[3];                                                                       \
      float ex_[6], ex2_[6], tA_[6], tB_[6];                                                        \
      G4_T_READ(ROLE, 0, 0, 0, 0, rd_, ex_)                                                         \
      G4_T_READ(ROLE, 0, 0, 0, 1, rd2_, ex2_)                                                       \
      G4_T_COL(ROLE, 0, rd_, ex_)                                                                   \
      G4_T_COL(ROLE, 1, rd2_, ex2_)                                                                 \
      G4_T_ROW(ROLE, 0, tA_, jA)                                                                    \
      G4_T_ROW(ROLE, 0, tB_, jB)                                                                    \
    }                                                                                               \
    __syncthreads();                                                                                \
    float a2_[2] = {0.f, 0.f}, b2_[2] = {0.f, 0.f}; /* the rotated pair of the k-step before */      \
    for (int s = s_begin; s < s_end; s += 3) { /* whole triples: stages beyond s_end are all-zero operands */ \
      G4_STAGE(ROLE, 0)                                                                             \
      G4_STAGE(ROLE, 1)                                                                             \
      G4_STAGE(ROLE, 2)                                                                             \
    }                                                                                               \
    G4_MMA2(4, a2_, b2_)                                                                            \
  }

  f32x16 acc[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  if (PRO) {
    for (int idx = tid; idx < a.pro_nseg * 32; idx += G4_NT) {
      const int c = ci0 + (idx & 31), so = (idx >> 5) * a.Ci;
      float sc = 0.f, sh = 0.f;
      if (c < a.Ci) {
        sc = a.pro_invstd[so + c] * a.pro_gamma[c];
        sh = fmaf(-a.pro_mean[so + c], sc, a.pro_beta[c]);
      }
      pro2[idx] = make_float2(sc, sh);
    }
  }
  if (s_begin < s_end) {
    G4_REQ(0)
    G4_REQ(1)
    G4_REQ(2)
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    __syncthreads();
    if (PRO) {
      G4_FIX(0)
      __syncthreads();
    }
    if (r_x && tp == 2) G4_LOOP(3)
    else if (r_x) G4_LOOP(1)
    else if (r_y) G4_LOOP(2)
    else G4_LOOP(0)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- partial of this slice.  acc[i][r] = dU of frequency (i, wj), co = co0 + wsb*32 + row(r, hh), ci = ci0 + l31: a
  // wave owns a whole frequency COLUMN, so the row half of G^T dU G — T[a][wj] = sum_i G[i][a] dU[i][wj], 6 -> 3 values —
  // is done here in registers (linear: it commutes with the sum over slices); the partials and everything the reducers
  // read are half the size ([n_slices][3][6][Co_pad][Ci_pad]).  The column half runs in wino4_wgrad_reduce_kernel.
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float u0 = acc[0][r], u1 = acc[1][r], u2 = acc[2][r], u3 = acc[3][r], u4 = acc[4][r], u5 = acc[5][r];
    const float s12 = u1 + u2, d12 = u2 - u1, s34 = u3 + u4, d34 = u3 - u4;
    acc[0][r] = 0.25f * u0 - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
    acc[1][r] = (1.f / 6.f) * d12 + (1.f / 12.f) * d34;
    acc[2][r] = -(1.f / 6.f) * s12 + (1.f / 6.f) * s34 + u5;
  }
#pragma unroll
  for (int ar = 0; ar < 3; ++ar) {
    float* base = a.ws + ((size_t)(slice * 18 + ar * 6 + wj) * a.Co_pad + co0 + wsb * 32) * a.Ci_pad + ci0 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
      base[(size_t)row * a.Ci_pad] = acc[ar][r];
    }
  }
}

// dW[co][ci] = (sum over slices T[.][.][co][ci]) G, T = G^T dU the row-transformed partials [3][6] of the kernel above.
// Block = one co x 64 ci x 4 slice phases.
__global__ void __launch_bounds__(256) wino4_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                                 int Co, int Ci, int Co_pad, int Ci_pad, int n_slices) {
  __shared__ float red[3][18][64];
  const int cil = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int n_cic = (Ci + 63) / 64;
  const int co = blockIdx.x / n_cic, ci = (blockIdx.x % n_cic) * 64 + cil;
  float t[18];
#pragma unroll
  for (int f = 0; f < 18; ++f) t[f] = 0.f;
  if (ci < Ci) {
    for (int s = ph; s < n_slices; s += 4) {
      const float* p = ws + ((size_t)(s * 18) * Co_pad + co) * Ci_pad + ci;
#pragma unroll
      for (int f = 0; f < 18; ++f) t[f] += p[(size_t)f * Co_pad * Ci_pad];
    }
  }
  if (ph > 0) {
#pragma unroll
    for (int f = 0; f < 18; ++f) red[ph - 1][f][cil] = t[f];
  }
  __syncthreads();
  if (ph == 0 && ci < Ci) {
#pragma unroll
    for (int f = 0; f < 18; ++f) t[f] = ((t[f] + red[0][f][cil]) + red[1][f][cil]) + red[2][f][cil];
    // dW[r][.] = T[r][.] G;  G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]
    float* dst = dw + ((size_t)co * Ci + ci) * 9;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float* tr = t + r * 6;
      const float s12 = tr[1] + tr[2], d12 = tr[2] - tr[1], s34 = tr[3] + tr[4], d34 = tr[3] - tr[4];
      dst[r * 3 + 0] = 0.25f * tr[0] - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
      dst[r * 3 + 1] = (1.f / 6.f) * d12 + (1.f / 12.f) * d34;
      dst[r * 3 + 2] = -(1.f / 6.f) * s12 + (1.f / 6.f) * s34 + tr[5];
    }
  }
}

namespace {
inline int wino4_wg_ips(int W) { return W >= 16 ? 1 : 16 / W; }  // images per 16-column strip
struct Wino4WgPlan {
  int Ci_pad, Co_pad, n_ci_tiles, n_co_tiles, nrh, nrw, nstages, sps, n_slices;
};

int wino4_wg_plan(int B, int Ci, int Co, int H, int W, Wino4WgPlan* p) {
  p->n_ci_tiles = cdiv(Ci, 32);
  p->n_co_tiles = cdiv(Co, 64);
  p->Ci_pad = p->n_ci_tiles * 32;
  p->Co_pad = p->n_co_tiles * 64;
  p->nrh = H / 4;
  p->nrw = W >= 16 ? W / 16 : 1;
  const long long ns = (long long)(B / wino4_wg_ips(W)) * p->nrh * p->nrw;
  if (ns > 0x3fffffffLL) return SIVAE_ERR_RANGE;
  p->nstages = (int)ns;
  const int ntiles = p->n_ci_tiles * p->n_co_tiles;
  // one block per CU is resident (12 waves, ~158 KB LDS): aim at SIVAE_WG4_SLOTS blocks (default: one round of the CUs),
  // at least 24 stages per slice so that the 144 KB partial write-out of a block stays small next to its MFMA work
  static int slots = 0;
  if (slots == 0) {
    const char* e = getenv("SIVAE_WG4_SLOTS");
    slots = e ? atoi(e) : sivae_num_cus();
    if (slots <= 0) slots = 256;
  }
  int n_slices = cdiv(slots, ntiles);
  const int max_slices = p->nstages / 24 > 0 ? p->nstages / 24 : 1;
  if (n_slices > max_slices) n_slices = max_slices;
  p->sps = cdiv(cdiv(p->nstages, n_slices), 3) * 3;  // whole stage triples
  p->n_slices = cdiv(p->nstages, p->sps);
  return SIVAE_OK;
}
}  // namespace

// maps the F(4x4,3x3) weight gradient takes (its stage is a 4 x 16 pixel strip of whole tiles)
// (round 6: also the 8 x 8 and 4 x 4 maps, whose strip is 2 / 4 whole images side by side — the batch and the images per
// segment must then be multiples of sivae_conv2d_wino4_wgrad_images_per_stage)
extern "C" int sivae_conv2d_wino4_wgrad_supported(int H, int W) {
  if ((H == 8 && W == 8) || (H == 4 && W == 4)) return 1;
  return (H >= 4 && W >= 16 && (H % 4) == 0 && (W % 16) == 0) ? 1 : 0;
}
extern "C" int sivae_conv2d_wino4_wgrad_images_per_stage(int H, int W) {
  return sivae_conv2d_wino4_wgrad_supported(H, W) ? wino4_wg_ips(W) : 0;
}

// does it beat the F(2x2,3x3) weight gradient for this launch?  One block per CU is resident: it needs (co, ci) tiles x
// slices of >= 24 stages for every CU
extern "C" int sivae_conv2d_wino4_wgrad_pays(int B, int Ci, int Co, int H, int W) {
  if (B <= 0 || Ci < 16 || Co < 16 || !sivae_conv2d_wino4_wgrad_supported(H, W) || B % wino4_wg_ips(W)) return 0;
  const long long stages = (long long)(B / wino4_wg_ips(W)) * (H / 4) * (W >= 16 ? W / 16 : 1);
  const long long tiles = (long long)cdiv(Ci, 32) * cdiv(Co, 64);
  if (tiles * (stages / 24) >= sivae_num_cus()) return 1;
  if (wino4_wg_ips(W) > 1) {
    // 8 x 8 / 4 x 4 maps: the alternative is the direct weight gradient (conv_wgrad.hip): ~45 us + 1.5e-7 us per (image x
    // pixel x co x ci), never under ~60 us; this kernel costs ~15 us + 2.3 us per stage of a slice (fitted to
    // profiles/r6_wino4_small_maps_vs_f23.txt: 1.3-2.1x at the small batches of a per-GPU shard too).  Taken with a 10 % margin.
    Wino4WgPlan p;
    if (wino4_wg_plan(B, Ci, Co, H, W, &p) != SIVAE_OK) return 0;
    const double t4 = 15.0 + 2.3 * p.sps;
    double td = 45.0 + 1.5e-7 * (double)B * H * W * (double)Ci * Co;
    if (td < 60.0) td = 60.0;
    return 1.10 * t4 < td ? 1 : 0;
  }
  return 0;
}

extern "C" size_t sivae_conv2d_wino4_wgrad_workspace_bytes(int B, int Ci, int Co, int H, int W) {
  Wino4WgPlan p;
  if (B <= 0 || Ci <= 0 || Co <= 0 || !sivae_conv2d_wino4_wgrad_supported(H, W) || B % wino4_wg_ips(W)) return 0;
  if (wino4_wg_plan(B, Ci, Co, H, W, &p) != SIVAE_OK) return 0;
  // (+ one slice-sized slot: from 8 slices up the slices are summed by the wide split-K reducer first)
  return (size_t)(p.n_slices + 1) * 18 * p.Co_pad * p.Ci_pad * sizeof(float);
}

// dw[Co][Ci][3][3] = weight gradient of y = conv3x3(x', w) given dy, x' = x or (pro_mean != NULL) LeakyReLU(BatchNorm(x))
// with per-segment statistics when seg_images > 0 (B = nseg * seg_images, nseg <= 2; sivae_conv2d_wino_wgrad_seg)
extern "C" int sivae_conv2d_wino4_wgrad(const float* x, const float* dy, float* dw, const float* pro_mean,
                                        const float* pro_invstd, const float* pro_gamma, const float* pro_beta,
                                        float pro_slope, int B, int Ci, int Co, int H, int W, int seg_images,
                                        void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!x || !dy || !dw || !workspace) return SIVAE_ERR_NULL;
  if (seg_images < 0 || (seg_images > 0 && B % seg_images != 0)) return SIVAE_ERR_SHAPE;
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  if (!sivae_conv2d_wino4_wgrad_supported(H, W)) return SIVAE_ERR_SHAPE;
  const int ips = wino4_wg_ips(W);
  if ((B % ips) || (seg_images % ips)) return SIVAE_ERR_SHAPE;  // whole image strips, each inside one segment
  if (pro_mean && (!pro_invstd || !pro_gamma || !pro_beta)) return SIVAE_ERR_NULL;
  if (pro_mean && !(pro_slope >= 0.f && pro_slope <= 1.f)) return SIVAE_ERR_MODE;  // prologue uses max(v, v*slope)
  if ((((uintptr_t)x) & 15u) != 0 || (((uintptr_t)dy) & 15u) != 0) return SIVAE_ERR_SHAPE;  // 16-byte LDS-direct loads
  const long long hw = (long long)H * W;
  if ((long long)Ci * hw * 4 >= 0x7fffffffLL || (long long)Co * hw * 4 >= 0x7fffffffLL) return SIVAE_ERR_RANGE;
  Wino4WgPlan p;
  int rc = wino4_wg_plan(B, Ci, Co, H, W, &p);
  if (rc != SIVAE_OK) return rc;
  const size_t slice_elems = (size_t)18 * p.Co_pad * p.Ci_pad;  // (row-transformed partials: [3][6] per (co, ci))
  const size_t need = (size_t)(p.n_slices + 1) * slice_elems * sizeof(float);
  if (workspace_bytes < need) return SIVAE_ERR_WORKSPACE;
  Wino4WgArgs a;
  a.x = x;
  a.dy = dy;
  a.ws = static_cast<float*>(workspace);
  a.pro_mean = pro_mean;
  a.pro_invstd = pro_invstd;
  a.pro_gamma = pro_gamma;
  a.pro_beta = pro_beta;
  a.pro_slope = pro_slope;
  a.pro_seg_images = seg_images > 0 ? seg_images : B;
  a.pro_nseg = B / a.pro_seg_images;
  if (pro_mean && a.pro_nseg > 2) return SIVAE_ERR_SHAPE;
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
  const long long nblk = (long long)p.n_ci_tiles * p.n_co_tiles * p.n_slices;
  if (nblk > 0x7fffffffLL) return SIVAE_ERR_RANGE;
  a.nblk = (int)nblk;
  a.xcd_remap = sivae_xcd_remap();
  a.ips = ips;
  a.iw_l2 = ips > 1 ? ilog2_exact(W) : 0;
  a.seam_e0B = a.seam_e5B = a.seam_e0A = a.seam_e5A = 0ull;
  if (ips > 1)
    for (int l = 0; l < 64; ++l)
      for (int ka = 0; ka < 2; ++ka) {  // k-step B transforms tiles 0, 1 (tile = lane >> 5), k-step A tiles 2, 3
        const int px = 4 * ((l >> 5) + 2 * ka);
        if ((px & (W - 1)) == 0) (ka ? a.seam_e0A : a.seam_e0B) |= 1ull << l;
        if (((px + 4) & (W - 1)) == 0) (ka ? a.seam_e5A : a.seam_e5B) |= 1ull << l;
      }
  auto kern = ips > 1 ? (pro_mean ? wino4_wgrad_kernel<true, true> : wino4_wgrad_kernel<false, true>)
                      : (pro_mean ? wino4_wgrad_kernel<true, false> : wino4_wgrad_kernel<false, false>);
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.xcd_remap ? (nblk + 7) / 8 * 8 : nblk)), dim3(G4_NT), 0, stream, a);
  rc = sivae_launch_status();
  if (rc != SIVAE_OK) return rc;
  // G^T . G runs on Co x Ci / 64 blocks: with many slices (few channel tiles) that is a handful of blocks each walking
  // hundreds of strided rows — sum the slices with the wide reducer first (2 304+ blocks, fixed order: deterministic)
  const float* red_src = static_cast<const float*>(workspace);
  int red_slices = p.n_slices;
  if (p.n_slices >= 8) {
    float* sum = static_cast<float*>(workspace) + (size_t)p.n_slices * slice_elems;
    sivae_launch_slice_reduce(red_src, sum, p.n_slices, slice_elems, stream);
    rc = sivae_launch_status();
    if (rc != SIVAE_OK) return rc;
    red_src = sum;
    red_slices = 1;
  }
  const int n_cic = (Ci + 63) / 64;
  hipLaunchKernelGGL(wino4_wgrad_reduce_kernel, dim3((unsigned)(Co * n_cic)), dim3(256), 0, stream, red_src, dw, Co, Ci,
                     p.Co_pad, p.Ci_pad, red_slices);
  return sivae_launch_status();
}
