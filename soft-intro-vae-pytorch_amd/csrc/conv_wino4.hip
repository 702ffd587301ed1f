// Winograd F(4x4, 3x3) stride-1 "same" convolution for gfx950 on the exact-fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32): 36 multiplies per 4x4 output tile = 2.25 per output pixel instead of the 4 of
// F(2x2,3x3) (conv_wino.hip) and the 9 of the direct form — 1.78x fewer MFMA passes than conv_wino.hip for the same
// nn.Conv2d(k=3, s=1, p=1) (reference: soft_intro_vae/train_soft_intro_vae.py:56-61).  Forward and — fed the flipped /
// transposed weight transform — the data gradient, for the large-map layers (H % 16 == 0, W % 32 == 0); the deep
// 4x4 ... 16x16 layers stay on F(2x2,3x3).
//
//   V = B^T d B   (6x6 input patch d of a tile, per input channel)          B^T, G, A^T: Lavin & Gray's F(4x4,3x3)
//   U = G g G^T   (3x3 filter g -> 6x6, per (co, ci); once per optimizer step by sivae_pack_wino4_weight)
//   M[i][j] = sum_ci U[i][j][co][ci] * V[i][j][ci][tile]      <- 36 independent GEMMs, K = Ci
//   Y = A^T M A   (4x4 outputs of the tile)
// fp32 cost of the larger transforms (coefficients up to 8 and 1/24): 1.2e-5 relative on a 512-channel layer, 3.3e-5 on
// the reconstruction of the six-level 256x256 network end to end (profiles/r1_wino_numerics.txt) — inside the 1e-4 gate.
//
// Work split: a block is 12 waves = 64 output channels x 32 tiles (8 x 4 tiles = 32 x 16 pixels) x 36 frequencies.
// Wave (j, s) owns frequency COLUMN j (i = 0..5) of the 32-channel subtile s: 6 accumulators of 32x32 = 96 registers, so
// three waves fit a SIMD (168 registers) and one block fills a CU.  As in conv_wino.hip nothing the MFMA loop consumes
// is shared between waves except the raw zero-padded input halo in LDS:
//   * B operand: a lane (tile t = lane & 31, channel k = lane >> 5) reads six rows of its tile's patch — one aligned
//     16-byte ds_read per row for patch columns 1..4 (the LDS column origin is skewed by 3 so that column 1 of every
//     tile is 16-byte aligned; row stride 40: the four 16-lane groups of a ds_read_b128 hit 64 distinct banks) plus one
//     dword for column 0 / 5 —, forms t[r] = (row r of d) . (column j of B) with wave-uniform coefficients and
//     V[0..5] = B^T t in 12 more VALU ops.  The transformed tile is never stored.
//   * A operand: U is packed [j][ci][co][i], 24 bytes per (co, ci): one 16-byte + one 8-byte buffer load per k-step,
//     two k-steps ahead, straight into registers.
//   * halo: 34 x 18 raw pixels x 16 channels per chunk, HBM -> LDS by LDS-direct buffer loads (no staging registers:
//     the 96 accumulators leave none), double-buffered, one barrier per chunk; the first chunk of the NEXT work item is
//     requested during the last chunk of the current one.
// After the K loop: the row transform A^T M in registers (6 -> 4 values), then four rounds (one per output row of the
// tiles) of a 48 KB exchange through LDS (aliasing the halo buffer the item finished on) for the column transform
// across the six frequency-column waves; outputs leave as 16-byte stores (four consecutive pixels of one channel per
// lane) with the usual fused epilogues (accumulate, BatchNorm sum / sumsq partials).
#include "common.h"
#include "pack_batch.h"
#include <stdlib.h>

struct Wino4Args {
  const float* x;
  const float* up;  // packed U [6(j)][Ci_pad][Co_pad][6(i)]
  float* y;
  float* stats;  // [n_px_tiles][Co][2] or null
  // fused producer BatchNorm + LeakyReLU on the input (conv2 of a ResidualBlock reads LeakyReLU(BN1(conv1)) that is never
  // written, train_soft_intro_vae.py:58-61): x' = max(v, slope*v), v = (x - mean[c]) * invstd[c]*gamma[c] + beta[c];
  // segments: images [g*pro_seg_images, ...) use the statistics row g of pro_mean / pro_invstd ([pro_nseg][Ci])
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
  // Maps smaller than a work item's 32 x 16 pixels: the item is a GRID of whole images — 16 x 16: 2 x 1 images, 8 x 8:
  // 4 x 2, 4 x 4: 8 x 4 (two = 1; iw_l2 / ih_l2 = log2 of the image width / height, 0 otherwise).  Every seam between two
  // images is zero padding for both: the halo rows 0 / 17 and groups 0 / 9 are out of range for the loader, and the
  // transform role zeroes patch column 0 / 5 and patch row 0 / 5 of the tiles that touch a seam through the four 64-bit
  // lane masks below (lane -> tile = lane & 31, tile column = tile & 7, tile row = tile >> 3; computed by the launch).
  int two;
  int iw_l2, ih_l2;
  unsigned long long seam_c0, seam_c5, seam_r0, seam_r5;
  // split-K (launches with fewer work items than CUs): an item = (pixel tile, K slice, co tile); slice ks walks the cps
  // chunks from ks * cps on and writes its partial outputs to y + ks * slice_stride (summed by wino4_splitk_reduce_kernel)
  int ksl, cps;
  long long slice_stride;
};

#define W4_CK 8
#define W4_RS 40
#define W4_LH 18
#define W4_PLANE 768  // 18 rows x 40 = 720, padded to 3 x 64 groups of 16 bytes: a wave's LDS-direct load fills 64 groups
#define W4_XBUF (W4_CK * W4_PLANE)  // raw halo of one chunk: 6144 floats (24 KB)
#define W4_VBUF (36 * W4_CK * 32)   // transformed chunk V[freq][ch][tile]: 9216 floats (36 KB)
#define W4_EXF 12288                // output-transform exchange: 48 KB = the second V buffer + 12 KB behind it
#define W4_NT 768
#define W4_TCO 64
#define W4_PXH 16
#define W4_PXW 32

// cache policy of the output stores (auxiliary bits of the buffer instruction: 2 = nt)
#ifndef W4_STORE_AUX
#define W4_STORE_AUX 0
#endif
// out-of-range marker of 16-byte stores: beyond every window (Co*H*W*4 < 2^31 is checked at launch), and neither the
// store's own 16 bytes nor a row offset added on top wraps around 2^32 (a dwordx4 store at 0xFFFFFFFF drops only its first
// dword: the other three wrap into the window)
#define W4_OOB16 0x80000000u
__device__ __forceinline__ void buf_store_f32x4(__amdgpu_buffer_rsrc_t r, float4 v, unsigned voff, unsigned soff = 0u) {
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  f32x4 f = {v.x, v.y, v.z, v.w};
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f), r, (int)voff, (int)soff, W4_STORE_AUX);
}

#define W4_PRO_MAX 1024  // prologue table entries (segments x padded input channels)

// Timing ablations (compile with -DW4_ABLATE=<bits>; results are WRONG with any bit set — tools/w4_timing.py):
//   1 no halo loads in the K loop, 2 no U refills, 8 no chunk barriers, 16 no output stores
#ifndef W4_ABLATE
#define W4_ABLATE 0
#endif

// Structure (round 3, second form).  On gfx950 the fp32 "matrix" instruction runs at the fp32 VECTOR rate and — measured
// here: a k-step costs 18 MFMA x 64 + (all other VALU of the SIMD's three waves) x 4 cycles — its time and the time of
// ordinary VALU instructions on the same SIMD ADD.  The first form of this kernel let every wave transform its own B
// operand from the raw halo (7.5 VALU per MFMA: +47 %).  Now the input transform is done ONCE per block and chunk:
//   * transform role: thread = (tile, channel, column pair) — the frequency columns come in pairs that share their
//     partial sums ((1,2): (d4-4d2) +- (d3-4d1); (3,4): (d4-d2) +- 2(d3-d1); (0,5) stand alone) — reads six patch rows
//     (one aligned 16-byte ds_read each, + two dwords for the (0,5) pair), and writes its 12 transformed values to
//     V[freq][channel][tile] in LDS: ~54 VALU per thread and chunk = 2.25 per MFMA;
//   * MFMA role: wave (j, s) = frequency column j x 32-channel subtile s as before (6 accumulators = 96 registers, three
//     waves per SIMD, one block per CU); its B operand is ONE conflict-free ds_read_b32 per MFMA out of V, its A operand
//     (U, packed [j][ci][co][i]) a 16-byte + an 8-byte buffer load per k-step, two k-steps ahead.
// Pipeline per 8-channel chunk c (one barrier): MFMAs on V[c & 1]  ||  transform of the raw halo of chunk c+1 (slices
// between the MFMAs) into V[(c+1) & 1]  ||  LDS-direct 16-byte loads of the halo of chunk c+2 (two per wave) into the raw
// buffer chunk c's transform freed; across work items the chunk numbering simply continues (the first two chunks of the
// NEXT item are requested / transformed during the last two chunks of the current one).  With the fused BatchNorm +
// LeakyReLU prologue every thread rewrites the two groups it requested before the barrier publishes them.
// Phase stamps of the first items of every block (tools/w4_timing.py; a -DW4_TIMING build only)
#ifdef W4_TIMING
__device__ long long w4_dbg[256 * 8 * 16];
#define W4_STAMP(SLOT)                                                                       \
  if (tid == 0 && it_n < 8) w4_dbg[((int)blockIdx.x * 8 + it_n) * 16 + (SLOT)] = ((SLOT) == 6 || (SLOT) == 7) ? clock64() : wall_clock64();
extern "C" int sivae_debug_w4_read(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(w4_dbg), sizeof(long long) * 256 * 8 * 16);
}
#define W4_STAMPK(CH) \
  if (tid == 0 && it_n < 8 && (CH) < 16) w4_dbg[((int)blockIdx.x * 8 + it_n) * 16 + 8 + ((CH) >> 1)] = wall_clock64();
#else
#define W4_STAMP(SLOT)
#define W4_STAMPK(CH)
#endif

// GRID: the image-grid mode (maps up to 16 x 16) as its own instantiation — the large-map kernels carry neither its seam
// masks (eight scalar registers) nor its selects
template <bool PRO, bool GRID = false>
__global__ void __launch_bounds__(W4_NT, 1) conv_wino4_kernel(Wino4Args a) {
  constexpr int CK = W4_CK, RS = W4_RS, PLANE = W4_PLANE, XBUF = W4_XBUF, VBUF = W4_VBUF;
  // the two raw-halo buffers are SEPARATE static arrays: the compiler orders a ds_read behind every in-flight LDS-direct
  // load it cannot prove disjoint (vmcnt(0) before the read); the transform reads one buffer while the loads fill the other
  __shared__ __attribute__((aligned(16))) float raw0[XBUF];
  __shared__ __attribute__((aligned(16))) float raw1[XBUF];
  __shared__ __attribute__((aligned(16))) float vx[VBUF + W4_EXF];  // V0 | V1 + spare (= the epilogue's exchange area)
#define RAWB(BUF) ((BUF) ? raw1 : raw0)
  // {mean, invstd*gamma, beta, -} per (segment, input channel); padded channels carry zeros (-> x' = 0)
  __shared__ float4 pro4[PRO ? W4_PRO_MAX : 1];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int wj = wave % 6, ws = wave / 6;
  const int H = a.H, W = a.W, HW = H * W;
  const int tx = l31 & 7, ty = l31 >> 3;

  // ---- transform role: column pair tp, item (tile tt, channel tc) of the chunk
  const int tp = wave >> 2;                       // 0: columns (1,2)   1: columns (3,4)   2: columns (0,5)
  const int ti = (wave & 3) * 64 + lane, tt = ti & 31, tc = ti >> 5;
  const int trb = tc * PLANE + 4 * (tt >> 3) * RS + 4 * (tt & 7) + 4;  // patch column 1 at LDS column 4*tx + 4
  const int jA = tp == 0 ? 1 : (tp == 1 ? 3 : 0), jB = tp == 0 ? 2 : (tp == 1 ? 4 : 5);
  const int tvb = tc * 32 + tt;
  const float t_al = tp == 0 ? -4.f : -1.f;  // a = d4 + al*d2
  const float t_be = tp == 0 ? 1.f : 2.f;    // b = be*d3 + ga*d1
  const float t_ga = tp == 0 ? -4.f : -2.f;
  const bool pair05 = tp == 2;
  // pair mode: lanes whose tile column is 4 (patch column 0 = the other image's last column) / 3 (patch column 5 = the
  // other image's first column) — as 64-bit lane masks for v_cndmask (tile = lane & 31, column = tile & 7)
  const unsigned long long seam_m0 = a.seam_c0, seam_m5 = a.seam_c5, seam_mr0 = a.seam_r0, seam_mr5 = a.seam_r5;
  // image grid of an item (branch-free: one image -> bounds H x W, masks ~0, image stride 0)
  const int two_w = GRID ? W4_PXW : a.W, two_h = GRID ? W4_PXH : a.H;
  const int two_mask = GRID ? a.W - 1 : -1, two_hmask = GRID ? a.H - 1 : -1;
  const int two_img = GRID ? a.Ci * a.H * a.W : 0, two_nx = W4_PXW >> a.iw_l2, two_ipi = GRID ? two_nx * (W4_PXH >> a.ih_l2) : 1;
  // ---- MFMA role: B operand V[(i*6 + wj)][2*kk + hh][l31]
  const int vrb = (wj * CK + hh) * 32 + l31;

  // ---- halo role: a channel plane is 192 groups of four floats (18 rows x 10 groups + padding), three wave-instructions;
  // wave w fills third w % 3 of the planes w / 3 + 4n (n = 0, 1).  Group k of a row = image columns c0 - 4 + 4k .. + 3:
  // entirely inside or entirely outside the image (W % 32 == 0).
  const int dsub = wave % 3, dpl0 = wave / 3;
  const int pg = dsub * 64 + lane, prow = pg / 10, pk = pg - prow * 10;
  const bool pvalid = pg < 180;

  const int n_items = a.n_items;
  const int nchunks = a.cps;  // chunks per work item: Ci_pad / CK (a multiple of 4), or an even share >= 4 of it per K slice
  const int ksteps = nchunks * (CK / 2);
  const __amdgpu_buffer_rsrc_t ursrc = make_rsrc(a.up, 36ull * a.Ci_pad * a.Co_pad * 4ull);
  const unsigned va0 = (unsigned)(hh * a.Co_pad + ws * 32 + l31) * 24u;
  const unsigned ua_step = (unsigned)a.Co_pad * 24u;  // bytes per input channel

  // Consecutive blockIdx go round-robin to the 8 XCDs.  With xcd_group the block on XCD x, slot j starts at item
  // x * (grid / 8) + j: the co-tiles of one pixel tile (consecutive items) run on ONE XCD at the same time and share the
  // halo in its L2 (the halo stream is 2.4 KB per input channel and item: 2.6 TB/s at full matrix rate with one reader)
  int item = blockIdx.x;
  if (a.xcd_group) item = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
  int b, r0, c0, co0, pt;
  int cbase = 0, kslice = 0;  // first input channel / index of the item's K slice (split-K)
  __amdgpu_buffer_rsrc_t xrsrc;
  unsigned xo, ua_base;
  int pseg = 0;  // table offset of the segment of the item whose halo is being requested
#define W4_SETUP(ITEM)                                                   \
  {                                                                      \
    const int co_tile = (ITEM) % a.n_co_tiles;                           \
    const int iq_ = (ITEM) / a.n_co_tiles;                               \
    kslice = iq_ % a.ksl;                                                \
    pt = iq_ / a.ksl;                                                    \
    cbase = kslice * a.cps * CK;                                         \
    const int tbx = pt % a.nbw;                                          \
    const int t2 = pt / a.nbw;                                           \
    const int tby = t2 % a.nbh;                                          \
    b = t2 / a.nbh;                                                      \
    r0 = tby * W4_PXH;                                                   \
    c0 = tbx * W4_PXW;                                                   \
    co0 = co_tile * W4_TCO;                                              \
    /* image-grid mode (maps up to 16 x 16): images ipi * pt ... in row-major order of the grid — halo row r / group  */ \
    /* column c belong to image (r >> ih_l2) * nx + (c >> iw_l2) at (r & (H-1), c & (W-1)); rows -1 / 16 and groups   */ \
    /* -4 / 32 are zero padding (branch-free: bounds 16 x 32, masks H-1 / W-1, stride Ci*HW; otherwise H x W, ~0, 0)   */ \
    b = GRID ? two_ipi * pt : b;                                        \
    xrsrc = make_rsrc(a.x + (size_t)b * a.Ci * HW, (unsigned long long)two_ipi * a.Ci * HW * 4ull); \
    const int r = r0 - 1 + prow, c = c0 - 4 + 4 * pk;                    \
    xo = (pvalid && r >= 0 && r < two_h && c >= 0 && c < two_w)          \
             ? (unsigned)(((r >> a.ih_l2) * two_nx + (c >> a.iw_l2)) * two_img + (r & two_hmask) * W + (c & two_mask)) * 4u \
             : SIVAE_OOB;                                                \
    ua_base = (unsigned)((wj * a.Ci_pad + cbase) * a.Co_pad + co0) * 24u; \
    if (PRO) pseg = (b / a.pro_seg_images) * a.Ci_pad + cbase;           \
  }
  // piece N (0, 1) of halo chunk CH -> raw buffer BUF (out-of-image / padding groups receive 0; channels beyond Ci
  // re-read the last one: their U is zero)
#define W4_DMA1(CH, BUF, N)                                              \
  if (!((W4_ABLATE & 1) && item >= 0)) {                                 \
    const int ck = dpl0 + 4 * (N);                                       \
    const int ci = cbase + (CH)*CK + ck;                                 \
    const int cic = ci < a.Ci ? ci : a.Ci - 1;                           \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(                            \
        xrsrc, (float __attribute__((address_space(3)))*)(RAWB(BUF) + ck * PLANE + dsub * 256), 16, xo, \
        (unsigned)cic * (unsigned)HW * 4u, 0, 0);                        \
  }
  // The fused BatchNorm + LeakyReLU prologue: the halo arrives RAW in LDS (LDS-direct loads bypass the registers), so
  // every thread rewrites the two 16-byte groups IT requested (no other thread touches them before the chunk's barrier):
  // x' = max(v, slope * v) * inside-the-image, v = (x - mean) * scale + beta.
#define W4_FIXUP(CH, BUF, XO, PSEG)                                      \
  {                                                                      \
    const float msk_ = (XO) != SIVAE_OOB ? 1.f : 0.f;                    \
    _Pragma("unroll") for (int n_ = 0; n_ < 2; ++n_) {                   \
      const int ck = dpl0 + 4 * n_;                                      \
      float4* q_ = reinterpret_cast<float4*>(RAWB(BUF) + ck * PLANE + dsub * 256 + lane * 4); \
      const float4 p_ = pro4[(PSEG) + (CH)*CK + ck];                     \
      float4 v_ = *q_;                                                   \
      v_.x = fmaf(v_.x - p_.x, p_.y, p_.z);                              \
      v_.y = fmaf(v_.y - p_.x, p_.y, p_.z);                              \
      v_.z = fmaf(v_.z - p_.x, p_.y, p_.z);                              \
      v_.w = fmaf(v_.w - p_.x, p_.y, p_.z);                              \
      v_.x = fmaxf(v_.x, v_.x * a.pro_slope) * msk_;                     \
      v_.y = fmaxf(v_.y, v_.y * a.pro_slope) * msk_;                     \
      v_.z = fmaxf(v_.z, v_.z * a.pro_slope) * msk_;                     \
      v_.w = fmaxf(v_.w, v_.w * a.pro_slope) * msk_;                     \
      *q_ = v_;                                                          \
    }                                                                    \
  }

  f32x16 acc[6];
  float4 U4[2];
  float2 U2[2];
#if W4_ABLATE & 2
  U4[0] = U4[1] = make_float4(1.f, 0.5f, 0.25f, 2.f);
  U2[0] = U2[1] = make_float2(1.f, 0.5f);
#endif
  // (the U refill is UNCONDITIONAL with a selected base / clamped index: a load inside an `if` makes hipcc assume no
  // younger load is outstanding at the join and turns every later wait into vmcnt(0))
#define W4_LOAD_A(UBASE, KS_ABS, SLOT)                                   \
  if (!(W4_ABLATE & 2)) {                                                \
    const unsigned so = (UBASE) + (unsigned)(2 * (KS_ABS)) * ua_step;    \
    U4[SLOT] = buf_load_f32x4(ursrc, va0, so);                           \
    U2[SLOT] = buf_load_f32x2(ursrc, va0 + 16u, so);                     \
  }
#define W4_REFILL(CH, KK)                                                \
  {                                                                      \
    const int ks2 = (CH) * (CK / 2) + (KK) + 2;                          \
    const bool in_item = ks2 < ksteps;                                   \
    const unsigned ub_ = (in_item || !has_next) ? ua_cur : ua_base;      \
    const int kq_ = in_item ? ks2 : (has_next ? ks2 - ksteps : ksteps - 1); \
    W4_LOAD_A(ub_, kq_, (KK)&1)                                          \
  }
#define W4_FENCE __builtin_amdgcn_sched_barrier(0);
  // Workgroup barrier that orders LDS traffic only.  __syncthreads() is a full fence: with LDS-direct loads in flight
  // (they write LDS, so the compiler counts them) it emits s_waitcnt vmcnt(0) in front of the barrier — i.e. every chunk
  // would wait out the HBM latency of the halo pieces requested a moment earlier.  Those loads are ordered by hand (their
  // consumer sits two chunks later, behind U-operand waits that complete after them), so the barriers inside the K loop
  // wait for this wave's LDS reads / writes only.
#define W4_LDS_BARRIER asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // B operands of k-step KK of the transformed chunk in V buffer VB
#define W4_READB(VB, KK, BV)                                             \
  {                                                                      \
    const float* p_ = vx + (VB)*VBUF + 2 * (KK)*32 + vrb;                \
    _Pragma("unroll") for (int i = 0; i < 6; ++i) BV[i] = p_[i * 6 * CK * 32]; \
  }
  // ---- transform slices.  Rows R0..R0+2 of this thread's patch -> the pair's column dot products tA[r], tB[r]
#define W4_TREAD(RB, R0, P05)                                               \
  {                                                                      \
    const float* p_ = RAWB(RB) + trb + (R0)*RS;                          \
    _Pragma("unroll") for (int r = 0; r < 3; ++r) {                      \
      td_[r] = *reinterpret_cast<const float4*>(p_ + r * RS);            \
      if (P05) {                                                         \
        te0_[r] = p_[r * RS - 1];                                        \
        te5_[r] = p_[r * RS + 4];                                        \
        if (GRID) { /* a seam between two images is zero padding for both (16 x 16 maps: tile columns 3 | 4) */   \
          asm volatile("v_cndmask_b32_e64 %0, %0, 0, %1" : "+v"(te0_[r]) : "s"(seam_m0));                \
          asm volatile("v_cndmask_b32_e64 %0, %0, 0, %1" : "+v"(te5_[r]) : "s"(seam_m5));                \
        }                                                                \
      }                                                                  \
    }                                                                    \
  }
#define W4_TDOT(R0, P05)                                                    \
  {                                                                      \
    _Pragma("unroll") for (int r = 0; r < 3; ++r) {                      \
      if (P05) {                                                         \
        tA_[(R0) + r] = fmaf(4.f, te0_[r], fmaf(-5.f, td_[r].y, td_[r].w)); \
        tB_[(R0) + r] = fmaf(4.f, td_[r].x, fmaf(-5.f, td_[r].z, te5_[r])); \
      } else {                                                           \
        const float a_ = fmaf(t_al, td_[r].y, td_[r].w);                 \
        const float b_ = fmaf(t_ga, td_[r].x, t_be * td_[r].z);          \
        tA_[(R0) + r] = a_ + b_;                                         \
        tB_[(R0) + r] = a_ - b_;                                         \
      }                                                                  \
    }                                                                    \
    if (GRID) { /* patch row 0 / 5 of a tile at the top / bottom edge of its image is zero padding */ \
      constexpr int re_ = (R0) == 0 ? 0 : 5;                             \
      const unsigned long long rm_ = (R0) == 0 ? seam_mr0 : seam_mr5;    \
      asm volatile("v_cndmask_b32_e64 %0, %0, 0, %1" : "+v"(tA_[re_]) : "s"(rm_)); \
      asm volatile("v_cndmask_b32_e64 %0, %0, 0, %1" : "+v"(tB_[re_]) : "s"(rm_)); \
    }                                                                    \
  }
  // V[.][J] = B^T t (the row direction) for one column of the pair -> V buffer VN
#define W4_TCOL(VN, T, J)                                                \
  {                                                                      \
    const float A_ = fmaf(-4.f, T[2], T[4]), B_ = fmaf(-4.f, T[1], T[3]); \
    const float C_ = T[4] - T[2], D_ = T[3] - T[1];                      \
    float* q_ = vx + (VN)*VBUF + (J)*CK * 32 + tvb;                      \
    q_[0 * 6 * CK * 32] = fmaf(4.f, T[0], fmaf(-5.f, T[2], T[4]));       \
    q_[1 * 6 * CK * 32] = A_ + B_;                                       \
    q_[2 * 6 * CK * 32] = A_ - B_;                                       \
    q_[3 * 6 * CK * 32] = fmaf(2.f, D_, C_);                             \
    q_[4 * 6 * CK * 32] = fmaf(-2.f, D_, C_);                            \
    q_[5 * 6 * CK * 32] = fmaf(4.f, T[1], fmaf(-5.f, T[3], T[5]));       \
  }
#define W4_MF(I, UV, BV) acc[I] = __builtin_amdgcn_mfma_f32_32x32x2f32(UV, BV[I], acc[I], 0, 0, 0);
  // One chunk c: four k-steps of six MFMAs on V buffer VB; between them the transform of raw buffer VB^1 (chunk c+1)
  // into V buffer VB^1 — all six patch rows are requested in k-step 0, then a MID barrier: raw buffer VB^1 is free from
  // there on and receives the halo of chunk c+3 (DCH; both pieces in k-step 3, behind that k-step's U refill: loads
  // complete in order) — a lead of five k-steps before chunk c+2's transform needs it, against HBM latency under load (the
  // first form requested chunk c+2 here: three k-steps, and the 128x128 / 256x256 layers, whose inputs stream from HBM,
  // paid ~25 % more per k-step than the 32x32 ones).  With the prologue the pieces of chunk c+2 (FCH, raw buffer VB,
  // requested during chunk c-1 with the offsets / segment XOF / PSF of that time) are fixed up in k-step 2.
#define W4_CHUNK(CH, VB, DCH, FCH, P05)                                  \
  {                                                                      \
    float4 td_[3];                                                       \
    float te0_[3], te5_[3], tA_[6], tB_[6];                              \
    float b0_[6], b1_[6];                                                \
    W4_READB(VB, 0, b0_)                                                 \
    W4_TREAD((VB) ^ 1, 0, P05)                                           \
    W4_FENCE                                                             \
    /* k-step 0 */                                                       \
    W4_READB(VB, 1, b1_)                                                 \
    W4_MF(0, U4[0].x, b0_) W4_MF(1, U4[0].y, b0_) W4_FENCE               \
    W4_TDOT(0, P05)                                                      \
    W4_FENCE                                                             \
    W4_MF(2, U4[0].z, b0_) W4_MF(3, U4[0].w, b0_) W4_FENCE               \
    W4_TREAD((VB) ^ 1, 3, P05)                                           \
    W4_FENCE                                                             \
    W4_MF(4, U2[0].x, b0_) W4_MF(5, U2[0].y, b0_) W4_FENCE               \
    W4_REFILL(CH, 0)                                                     \
    W4_FENCE                                                             \
    /* every wave has read its rows of raw buffer VB^1 (the reads have returned: lgkmcnt(0)) */ \
    if (!(W4_ABLATE & 8)) W4_LDS_BARRIER                                 \
    W4_FENCE                                                             \
    /* k-step 1 */                                                       \
    W4_READB(VB, 2, b0_)                                                 \
    W4_MF(0, U4[1].x, b1_) W4_MF(1, U4[1].y, b1_) W4_FENCE               \
    W4_TDOT(3, P05)                                                      \
    W4_FENCE                                                             \
    W4_MF(2, U4[1].z, b1_) W4_MF(3, U4[1].w, b1_) W4_MF(4, U2[1].x, b1_) W4_MF(5, U2[1].y, b1_) W4_FENCE \
    W4_REFILL(CH, 1)                                                     \
    W4_FENCE                                                             \
    /* k-step 2 */                                                       \
    W4_READB(VB, 3, b1_)                                                 \
    W4_MF(0, U4[0].x, b0_) W4_MF(1, U4[0].y, b0_) W4_MF(2, U4[0].z, b0_) W4_FENCE \
    W4_TCOL((VB) ^ 1, tA_, jA)                                           \
    W4_FENCE                                                             \
    W4_MF(3, U4[0].w, b0_) W4_MF(4, U2[0].x, b0_) W4_MF(5, U2[0].y, b0_) W4_FENCE \
    if (PRO) {                                                           \
      W4_FIXUP(FCH, VB, xo_f, pseg_f)                                    \
      W4_FENCE                                                           \
    }                                                                    \
    W4_REFILL(CH, 2)                                                     \
    W4_FENCE                                                             \
    /* k-step 3 */                                                       \
    W4_MF(0, U4[1].x, b1_) W4_MF(1, U4[1].y, b1_) W4_MF(2, U4[1].z, b1_) W4_FENCE \
    W4_TCOL((VB) ^ 1, tB_, jB)                                           \
    W4_FENCE                                                             \
    W4_MF(3, U4[1].w, b1_) W4_MF(4, U2[1].x, b1_) W4_MF(5, U2[1].y, b1_) W4_FENCE \
    W4_REFILL(CH, 3)                                                     \
    W4_DMA1(DCH, (VB) ^ 1, 0)                                            \
    W4_DMA1(DCH, (VB) ^ 1, 1)                                            \
    xo_f = xo;                                                           \
    pseg_f = pseg;                                                       \
    W4_FENCE                                                             \
    /* (the halo requested during the PREVIOUS chunk has landed: the U operands of this k-step, younger, were waited for) */ \
    if (!(W4_ABLATE & 8)) W4_LDS_BARRIER                                 \
  }
  // the whole transform of raw buffer RB into V buffer VN in one go (the very first chunk of a block)
#define W4_TRANSFORM_ALL(RB, VN, P05)                                    \
  {                                                                      \
    float4 td_[3];                                                       \
    float te0_[3], te5_[3], tA_[6], tB_[6];                              \
    W4_TREAD(RB, 0, P05)                                                 \
    W4_TDOT(0, P05)                                                      \
    W4_TREAD(RB, 3, P05)                                                 \
    W4_TDOT(3, P05)                                                      \
    W4_TCOL(VN, tA_, jA)                                                 \
    W4_TCOL(VN, tB_, jB)                                                 \
  }
  // two chunks (V buffers 0 then 1).  Chunk c requests the halo of chunk c+3 and fixes up / will transform chunk c+2:
  // from the item's third-last chunk on these belong to the NEXT item's chunks 0, 1, 2 (the chunk numbering simply runs
  // on; the last item of a block re-requests its own: unconditional loads)
#define W4_PAIR(CH, P05)                                                 \
  {                                                                      \
    const int d0_ = (CH) + 3 < nchunks ? (CH) + 3 : (CH) + 3 - nchunks;  \
    const int f0_ = (CH) + 2 < nchunks ? (CH) + 2 : (CH) + 2 - nchunks;  \
    W4_CHUNK(CH, 0, d0_, f0_, P05)                                       \
    if ((CH) + 4 == nchunks && has_next) W4_SETUP(next)                  \
    const int d1_ = (CH) + 4 < nchunks ? (CH) + 4 : (CH) + 4 - nchunks;  \
    const int f1_ = (CH) + 3 < nchunks ? (CH) + 3 : (CH) + 3 - nchunks;  \
    W4_CHUNK((CH) + 1, 1, d1_, f1_, P05)                                 \
  }

  if (PRO) {
    for (int idx = tid; idx < a.pro_nseg * a.Ci_pad; idx += W4_NT) {
      const int c = idx % a.Ci_pad, so = (idx / a.Ci_pad) * a.Ci;  // (segment g's statistics start at g * Ci)
      pro4[idx] = c < a.Ci ? make_float4(a.pro_mean[so + c], a.pro_invstd[so + c] * a.pro_gamma[c], a.pro_beta[c], 0.f)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
  }
  W4_SETUP(item)
  W4_DMA1(0, 0, 0)
  W4_DMA1(0, 0, 1)
  W4_DMA1(1, 1, 0)
  W4_DMA1(1, 1, 1)
  unsigned ua_cur = ua_base;
  W4_LOAD_A(ua_cur, 0, 0)
  W4_LOAD_A(ua_cur, 1, 1)
  __builtin_amdgcn_s_waitcnt(0x0F70);
  if (PRO) {
    W4_FIXUP(0, 0, xo, pseg)
    W4_FIXUP(1, 1, xo, pseg)
  }
  __syncthreads();
  if (pair05) W4_TRANSFORM_ALL(0, 0, true) else W4_TRANSFORM_ALL(0, 0, false)
  __syncthreads();
  W4_DMA1(2, 0, 0)  // chunk 2 into the raw buffer the first transform freed
  W4_DMA1(2, 0, 1)
  unsigned xo_f = xo;  // offsets / segment the halo pieces awaiting their fix-up were requested with
  int pseg_f = pseg;
#ifdef W4_TIMING
  int it_n = 0;
#endif
  for (;;) {
    W4_STAMP(0)
    W4_STAMP(6)
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // coordinates of the item being accumulated (W4_SETUP overwrites b, r0, ... for the next one in the last pair)
    const int e_pt = pt, e_b = b, e_r0 = r0, e_c0 = c0, e_co0 = co0, e_ks = kslice;
    const int next = item + (int)gridDim.x;
    const bool has_next = next < n_items;
    if (pair05) {
      for (int ch = 0; ch < nchunks; ch += 2) {
        W4_PAIR(ch, true)
        W4_STAMPK(ch)
      }
    } else {
      for (int ch = 0; ch < nchunks; ch += 2) {
        W4_PAIR(ch, false)
        W4_STAMPK(ch)
      }
    }
    W4_STAMP(1)

    // ---- output transform.  acc[i][r]: frequency (i, wj), tile = l31, channel = ws*32 + (r&3) + 8*(r>>2) + 4*hh.
    // Round a = output row a of every tile: Z[a][j] = sum_i A^T[a][i] M[i][j] in registers -> ex[j][s][r][lane] (48 KB:
    // the V buffer the item finished on + the spare behind it; V buffer 0 already holds the next item's first chunk);
    // then pair q = (s, r) of this wave: Y[a][0..3] = Z[a][.] A, one 16-byte store.
    {
      float* ex = vx + VBUF;
      const __amdgpu_buffer_rsrc_t yrsrc =
          make_rsrc(a.y + (size_t)e_ks * a.slice_stride + (size_t)e_b * a.Co * HW,
                    (unsigned long long)two_ipi * a.Co * HW * 4ull);
      float ssum[3] = {0.f, 0.f, 0.f}, ssq[3] = {0.f, 0.f, 0.f};
      // The lane index is laundered once per item: otherwise hipcc hoists the lane-dependent LDS / global offsets of this
      // epilogue out of the persistent item loop, spills them, and reloads each one behind an s_waitcnt vmcnt(0) — which
      // also waits for every output store issued so far (stores share vmcnt on gfx9): ~25 K cycles per item.
      // (and it is RECOMPUTED, not copied: `lane` itself is spilled by then, and its reload — `s_waitcnt vmcnt(1)` at the
      // epilogue's entry, in order behind the next item's halo requests — waited out an HBM latency per item)
      int lane_;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_));
      const int hh_ = lane_ >> 5;
      // image-grid mode: the tile at (row 4 ty, column 4 tx) of the item lies in image e_b + (4 ty >> ih_l2) * nx + (4 tx >> iw_l2)
      const int ty4 = ((lane_ >> 3) & 3) * 4, tx4 = (lane_ & 7) * 4;
      const int ty_ = (ty4 & two_hmask) >> 2, tx_ = (tx4 & two_mask) >> 2;
      const unsigned img_off =
          GRID ? (unsigned)((ty4 >> a.ih_l2) * two_nx + (tx4 >> a.iw_l2)) * (unsigned)(a.Co * HW) * 4u : 0u;
      // row transform Z = A^T M in place (acc[a][r] <- Z[a][wj] of channel slot r): the partial sums m1 +- m2, m3 +- m4
      // are shared by the four output rows (10 VALU per slot instead of 14)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float m0 = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r], m4 = acc[4][r], m5 = acc[5][r];
        const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
        acc[0][r] = m0 + s12 + s34;
        acc[1][r] = d12 + 2.f * d34;
        acc[2][r] = s12 + 4.f * s34;
        acc[3][r] = d12 + 8.f * d34 + m5;
      }
      // byte offset of output row 0 of this lane's tile for each of the wave's (s, r) pairs; the row of round a is added
      // through the scalar offset of the store (not range-checked: the marker of a padded channel stays out of range)
      unsigned off0[3];
#pragma unroll
      for (int qi = 0; qi < 3; ++qi) {
        const int q = wave + 12 * qi;
        const int chn = e_co0 + (q >> 4) * 32 + (q & 3) + 8 * ((q & 15) >> 2) + 4 * hh_;
        off0[qi] = chn < a.Co ? (unsigned)((chn * H + e_r0 + 4 * ty_) * W + e_c0 + 4 * tx_) * 4u + img_off : W4_OOB16;
      }
      // Store-data lifetime (see bn_fused.hip::BF_KEEP): the registers a 16-byte store reads stay pinned until the NEXT
      // store of the wave has been issued (six LDS reads, their wait and ~20 VALU later; the last one of a round until
      // the barrier behind it) — nothing recycles them while the store may still be reading its data under
      // back-pressure.  (Pinning a whole round — 12 registers — spilled 50-80 registers of this 168-register kernel.)
      float4 held[3];
      held[0] = held[1] = held[2] = make_float4(0.f, 0.f, 0.f, 0.f);
#define W4_KEEP(V) asm volatile("" ::"v"((V).x), "v"((V).y), "v"((V).z), "v"((V).w));
#pragma unroll
      for (int ar = 0; ar < 4; ++ar) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ex[((wj * 2 + ws) * 16 + r) * 64 + lane_] = acc[ar][r];
        __syncthreads();
        if (ar == 0) { W4_STAMP(2) }
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
            if (!((W4_ABLATE & 16) && item >= 0)) buf_store_f32x4(yrsrc, held[qi], off0[qi], row_off);
            if (qi > 0) W4_KEEP(held[qi - 1])
            ssum[qi] += (o.x + o.y) + (o.z + o.w);
            ssq[qi] += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
          }
        }
        __syncthreads();
        W4_KEEP(held[1]) W4_KEEP(held[2])
        if (ar == 0) { W4_STAMP(3) }
      }
      W4_STAMP(4)
#undef W4_KEEP
      if (a.stats != nullptr) {
#pragma unroll
        for (int qi = 0; qi < 3; ++qi) {
          const int q = wave + 12 * qi;
          const float s_ = half_wave_sum_hi(ssum[qi]);
          const float q_ = half_wave_sum_hi(ssq[qi]);
          if (q < 32) {
            const int chn = e_co0 + (q >> 4) * 32 + (q & 3) + 8 * ((q & 15) >> 2) + 4 * hh_;
            if ((lane_ & 31) == 31 && chn < a.Co) {  // (lane_ / hh_: the recomputed lane index — no spill reload here)
              float* dst = a.stats + ((size_t)e_pt * a.Co + chn) * 2;
              dst[0] = s_;
              dst[1] = q_;
            }
          }
        }
      }
    }
    W4_STAMP(5)
    W4_STAMP(7)
#ifdef W4_TIMING
    ++it_n;
#endif
    if (!has_next) break;
    item = next;
    ua_cur = ua_base;
  }
#undef W4_SETUP
#undef W4_FIXUP
#undef W4_DMA1
#undef W4_LOAD_A
#undef W4_REFILL
#undef W4_READB
#undef W4_TREAD
#undef W4_TDOT
#undef W4_TCOL
#undef W4_MF
#undef W4_FENCE
#undef W4_LDS_BARRIER
#undef W4_CHUNK
#undef W4_TRANSFORM_ALL
#undef W4_PAIR
#undef RAWB
}

// ---- weight transform U = G g G^T (6x6), packed [j][ci_pad][co_pad][i]; padding entries are zero
//   mode 0 (forward): g = w[n][k]            (n = output channel, k = input channel)
//   mode 1 (dgrad):   g = flip180(w[k][n])   (k = w's output channel is the GEMM's input channel)
__device__ __forceinline__ void pack_wino4_body(const float* __restrict__ w, float* __restrict__ up, int Co,
                                                         int Ci, int mode, int kdim, int ndim, int kpad, int npad, size_t idx0_, const size_t stride_) {
  const float G[6][3] = {{0.25f, 0.f, 0.f},
                         {-1.f / 6.f, -1.f / 6.f, -1.f / 6.f},
                         {-1.f / 6.f, 1.f / 6.f, -1.f / 6.f},
                         {1.f / 24.f, 1.f / 12.f, 1.f / 6.f},
                         {1.f / 24.f, -1.f / 12.f, 1.f / 6.f},
                         {0.f, 0.f, 1.f}};
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
    float gg[6][3];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int c = 0; c < 3; ++c) gg[i][c] = G[i][0] * g[0][c] + G[i][1] * g[1][c] + G[i][2] * g[2][c];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float* dst = up + (((size_t)j * kpad + k) * npad + n) * 6;
#pragma unroll
      for (int i = 0; i < 6; ++i) dst[i] = gg[i][0] * G[j][0] + gg[i][1] * G[j][1] + gg[i][2] * G[j][2];
    }
  }
}

__global__ void __launch_bounds__(256) pack_wino4_kernel(const float* __restrict__ w, float* __restrict__ up, int Co,
                                                         int Ci, int mode, int kdim, int ndim, int kpad, int npad) {
  pack_wino4_body(w, up, Co, Ci, mode, kdim, ndim, kpad, npad, (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)gridDim.x * 256);
}

__global__ void __launch_bounds__(256) pack_wino4_batch_kernel(const SivaePackJob* __restrict__ jobs,
                                                                const unsigned short* __restrict__ block_job) {
  const SivaePackJob j = jobs[block_job[blockIdx.x]];
  pack_wino4_body(j.w, j.dst, j.Co, j.Ci, j.mode, j.kdim, j.ndim, j.kpad, j.npad, (size_t)(blockIdx.x - j.blk0) * 256 + threadIdx.x, (size_t)j.nblk * 256);
}


static inline int w4_kpad(int k) { return ((k + 31) / 32) * 32; }  // an even number of 16-channel chunks
static inline int w4_npad(int n) { return ((n + W4_TCO - 1) / W4_TCO) * W4_TCO; }

extern "C" size_t sivae_pack_wino4_weight_bytes(int Co, int Ci, int mode) {
  if (Co <= 0 || Ci <= 0 || (mode != 0 && mode != 1)) return 0;
  const int kdim = mode == 0 ? Ci : Co, ndim = mode == 0 ? Co : Ci;
  return (size_t)36 * w4_kpad(kdim) * w4_npad(ndim) * sizeof(float);
}

extern "C" int sivae_pack_wino4_weight(const float* w, float* up, int Co, int Ci, int mode, hipStream_t stream) {
  if (!w || !up) return SIVAE_ERR_NULL;
  if (Co <= 0 || Ci <= 0) return SIVAE_ERR_SHAPE;
  if (mode != 0 && mode != 1) return SIVAE_ERR_MODE;
  const int kdim = mode == 0 ? Ci : Co, ndim = mode == 0 ? Co : Ci;
  const int kpad = w4_kpad(kdim), npad = w4_npad(ndim);
  int nb = cdiv((long long)kpad * npad, 256);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(pack_wino4_kernel, dim3(nb), dim3(256), 0, stream, w, up, Co, Ci, mode, kdim, ndim, kpad, npad);
  return sivae_launch_status();
}

// maps the F(4x4,3x3) kernel takes.  1: whole 32 x 16 pixel tile blocks (H % 16 == 0, W % 32 == 0); otherwise a work item
// is a grid of whole images and the batch (with segments: the images per segment) must be a multiple of
// sivae_conv2d_wino4_images_per_item — 2: 16 x 16 maps (2 images side by side), 3: 8 x 8 maps (4 x 2 images),
// 4: 4 x 4 maps (8 x 4 images).  0: not supported.
extern "C" int sivae_conv2d_wino4_supported(int H, int W) {
  if (H == 16 && W == 16) return 2;
  if (H == 8 && W == 8) return 3;
  if (H == 4 && W == 4) return 4;
  return (H >= 16 && W >= 32 && (H % W4_PXH) == 0 && (W % W4_PXW) == 0) ? 1 : 0;
}
// images per work item: 1 (mode 1), 2 / 8 / 32 (modes 2 / 3 / 4), 0 (unsupported map)
extern "C" int sivae_conv2d_wino4_images_per_item(int H, int W) {
  const int sup = sivae_conv2d_wino4_supported(H, W);
  return sup == 0 ? 0 : (sup == 1 ? 1 : (W4_PXW / W) * (W4_PXH / H));
}
static inline long long w4_px_tiles(int B, int H, int W) {
  const int ipi = sivae_conv2d_wino4_images_per_item(H, W);
  return ipi > 1 ? B / ipi : (long long)B * (H / W4_PXH) * (W / W4_PXW);
}

// does the F(4x4,3x3) kernel beat F(2x2,3x3) for this launch?  Its work item is 64 channels x 512 pixels and a CU holds
// ONE block: below one item per CU (the 512-channel 32x32 layers of an 8-image shard: 128 items) half the chip idles and
// the F(2x2,3x3) kernel with its 4x smaller items and split-K wins (measured 0.78x); from one item per CU up it is
// 1.27-1.57x (256x256 shard sizes 8 / 16 / 32 / 128 images).
extern "C" int sivae_conv2d_wino4_pays(int B, int Ci, int Co, int H, int W) {
  const int sup = sivae_conv2d_wino4_supported(H, W);
  if (B <= 0 || Ci < 16 || Co <= 0 || !sup || B % sivae_conv2d_wino4_images_per_item(H, W)) return 0;
  const long long items = w4_px_tiles(B, H, W) * ((Co + W4_TCO - 1) / W4_TCO);
  return items >= sivae_num_cus() ? 1 : 0;
}

extern "C" int sivae_conv2d_wino4_num_px_tiles(int B, int H, int W) {
  const int sup = sivae_conv2d_wino4_supported(H, W);
  if (B <= 0 || !sup || B % sivae_conv2d_wino4_images_per_item(H, W)) return SIVAE_ERR_SHAPE;
  return (int)w4_px_tiles(B, H, W);
}

// y[B][Co][H][W] (+)= conv3x3(x, U);  stats_partial (optional): [sivae_conv2d_wino4_num_px_tiles][Co][2] per-tile {sum, sumsq}
// of y in image order (sivae_bn_stats_from_conv / _seg).  The data gradient is this function on dy with the mode-1 pack.
static int wino4_impl(const float* x, const float* up, float* y, const float* pro_mean, const float* pro_invstd,
                      const float* pro_gamma, const float* pro_beta, float pro_slope, float* stats_partial, int B, int Ci,
                      int Co, int H, int W, int accumulate, int seg_images, hipStream_t stream, int ksl = 1) {
  if (!x || !up || !y) return SIVAE_ERR_NULL;
  if (pro_mean && (!pro_invstd || !pro_gamma || !pro_beta)) return SIVAE_ERR_NULL;
  if (pro_mean && !(pro_slope >= 0.f && pro_slope <= 1.f)) return SIVAE_ERR_MODE;  // prologue uses max(v, v*slope)
  if (seg_images < 0 || (seg_images > 0 && B % seg_images != 0)) return SIVAE_ERR_SHAPE;
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  const int sup = sivae_conv2d_wino4_supported(H, W);
  if (!sup) return SIVAE_ERR_SHAPE;
  const int ipi = sivae_conv2d_wino4_images_per_item(H, W);
  if ((B % ipi) || (seg_images % ipi)) return SIVAE_ERR_SHAPE;  // whole image grids, each inside one segment
  if (((uintptr_t)y & 15u) != 0) return SIVAE_ERR_SHAPE;  // 16-byte stores
  const long long hw = (long long)H * W;
  if ((long long)Ci * hw * 4 >= 0x7fffffffLL || (long long)Co * hw * 4 >= 0x7fffffffLL) return SIVAE_ERR_RANGE;
  Wino4Args a;
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
  a.Ci_pad = w4_kpad(Ci);
  a.Co_pad = w4_npad(Co);
  if (pro_mean && a.pro_nseg * a.Ci_pad > W4_PRO_MAX) return SIVAE_ERR_SHAPE;
  if (36ull * a.Ci_pad * a.Co_pad * 4ull >= 0xffffffffull) return SIVAE_ERR_RANGE;
  a.two = sup >= 2 ? 1 : 0;
  a.iw_l2 = a.two ? ilog2_exact(W) : 0;
  a.ih_l2 = a.two ? ilog2_exact(H) : 0;
  a.seam_c0 = a.seam_c5 = a.seam_r0 = a.seam_r5 = 0ull;
  if (a.two)
    for (int l = 0; l < 64; ++l) {  // lane -> tile = lane & 31 at pixel (4 * (tile >> 3), 4 * (tile & 7)) of the item
      const int px = 4 * (l & 7), py = 4 * ((l >> 3) & 3);
      if ((px & (W - 1)) == 0) a.seam_c0 |= 1ull << l;        // patch column 0 lies left of the tile's image
      if (((px + 4) & (W - 1)) == 0) a.seam_c5 |= 1ull << l;  // patch column 5 lies right of it
      if ((py & (H - 1)) == 0) a.seam_r0 |= 1ull << l;        // patch row 0 lies above it
      if (((py + 4) & (H - 1)) == 0) a.seam_r5 |= 1ull << l;  // patch row 5 lies below it
    }
  a.nbh = a.two ? 1 : H / W4_PXH;
  a.nbw = a.two ? 1 : W / W4_PXW;
  a.n_co_tiles = a.Co_pad / W4_TCO;
  a.accumulate = accumulate;
  const int nchunks_all = a.Ci_pad / W4_CK;
  if (ksl < 1 || nchunks_all % ksl != 0 || (ksl > 1 && ((nchunks_all / ksl) < 4 || ((nchunks_all / ksl) & 1))))
    return SIVAE_ERR_SHAPE;
  a.ksl = ksl;
  a.cps = nchunks_all / ksl;
  a.slice_stride = ksl > 1 ? (long long)B * Co * hw : 0;  // (y is then the [ksl][B][Co][H][W] partial-sum workspace)
  const long long nitems = w4_px_tiles(B, H, W) * a.n_co_tiles * ksl;
  if (nitems > 0x7fffffffLL) return SIVAE_ERR_RANGE;
  a.n_items = (int)nitems;
  // (96 KB of static LDS: one block per CU)
  const int cus = sivae_num_cus();
  const int grid = nitems < cus ? (int)nitems : cus;
  a.xcd_group = (sivae_xcd_remap() && a.n_co_tiles > 1 && !(grid & 7)) ? 1 : 0;
  if (a.two) {
    if (pro_mean)
      hipLaunchKernelGGL((conv_wino4_kernel<true, true>), dim3((unsigned)grid), dim3(W4_NT), 0, stream, a);
    else
      hipLaunchKernelGGL((conv_wino4_kernel<false, true>), dim3((unsigned)grid), dim3(W4_NT), 0, stream, a);
  } else if (pro_mean) {
    hipLaunchKernelGGL((conv_wino4_kernel<true, false>), dim3((unsigned)grid), dim3(W4_NT), 0, stream, a);
  } else {
    hipLaunchKernelGGL((conv_wino4_kernel<false, false>), dim3((unsigned)grid), dim3(W4_NT), 0, stream, a);
  }
  return sivae_launch_status();
}

extern "C" int sivae_conv2d_wino4_fwd(const float* x, const float* up, float* y, float* stats_partial, int B, int Ci,
                                      int Co, int H, int W, int accumulate, hipStream_t stream) {
  return wino4_impl(x, up, y, nullptr, nullptr, nullptr, nullptr, 1.f, stats_partial, B, Ci, Co, H, W, accumulate, 0,
                    stream);
}

// with the producer BatchNorm + LeakyReLU fused into the input (pro_mean != NULL); seg_images > 0: segmented batch,
// pro_mean / pro_invstd are [B / seg_images][Ci].  Ci_pad * segments <= 1024.
extern "C" int sivae_conv2d_wino4_fwd_pro(const float* x, const float* up, float* y, const float* pro_mean,
                                          const float* pro_invstd, const float* pro_gamma, const float* pro_beta,
                                          float pro_slope, float* stats_partial, int B, int Ci, int Co, int H, int W,
                                          int accumulate, int seg_images, hipStream_t stream) {
  return wino4_impl(x, up, y, pro_mean, pro_invstd, pro_gamma, pro_beta, pro_slope, stats_partial, B, Ci, Co, H, W,
                    accumulate, seg_images, stream);
}

// ---- split-K form for launches that would leave most of the chip idle (the 16-image shard of config 4 runs the
// 512-channel 16x16 / 32x32 layers as 64..128 work items of one block per CU).  The K range is cut into S slices (S x
// items ~ one block per CU), every (item, slice) writes its partial output tensor, and one small kernel sums the slices in
// a fixed order (deterministic), adds the old y when accumulating, and leaves per-IMAGE {sum, sumsq} rows for the
// consumer BatchNorm (as sivae_conv2d_wino_fwd_splitk does for the F(2x2,3x3) kernel).
__global__ void __launch_bounds__(64) wino4_splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ y,
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

// The same sum over 16-byte vectors with all S slice loads of a vector in flight at once (the scalar form above took
// 12-16 us per call on the 16- and 8-image shards: HW / 64 dependent rounds of S + 1 dword loads per lane — 25 / 38 calls
// per iteration).  NT = 64 for planes up to 16x16, 256 above; the slices are still added in slice order.
template <int NT>
__global__ void __launch_bounds__(256) wino4_splitk_reduce_vec_kernel(const float4* __restrict__ part, float4* __restrict__ y,
                                                                      float* __restrict__ stats, int S, int HW4,
                                                                      size_t slice_stride4, int accumulate, int n_planes) {
  // NT = 256: a block per plane; NT = 64: a wave per plane, four planes per block (8192 one-wave blocks took 9 us to launch)
  const int bc = NT == 64 ? (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6) : (int)blockIdx.x;  // b * C + c
  if (bc >= n_planes) return;
  const int t = NT == 64 ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
  const size_t base = (size_t)bc * HW4;
  float s = 0.f, q = 0.f;
  for (int p = t; p < HW4; p += NT) {
    float4 t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < S) t[k] = part[(size_t)k * slice_stride4 + base + p];
    float4 v = accumulate ? y[base + p] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < S) {
        v.x += t[k].x;
        v.y += t[k].y;
        v.z += t[k].z;
        v.w += t[k].w;
      }
    y[base + p] = v;
    s += (v.x + v.y) + (v.z + v.w);
    q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (stats != nullptr) {
    s = wave_sum(s);
    q = wave_sum(q);
    if (NT == 64) {
      if (t == 0) {
        stats[(size_t)bc * 2 + 0] = s;
        stats[(size_t)bc * 2 + 1] = q;
      }
    } else {
      __shared__ float red[2 * (NT / 64)];
      const int wave = threadIdx.x >> 6;
      if ((threadIdx.x & 63) == 0) {
        red[2 * wave] = s;
        red[2 * wave + 1] = q;
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        float ss = 0.f, qq = 0.f;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) {
          ss += red[2 * w];
          qq += red[2 * w + 1];
        }
        stats[(size_t)bc * 2 + 0] = ss;
        stats[(size_t)bc * 2 + 1] = qq;
      }
    }
  }
}

// number of K slices sivae_conv2d_wino4_fwd_splitk will use (1: the plain kernel; its statistics rows are then per pixel
// tile — sivae_conv2d_wino4_num_px_tiles —, otherwise per image: B rows)
extern "C" int sivae_conv2d_wino4_splitk(int B, int Ci, int Co, int H, int W) {
  const int sup = sivae_conv2d_wino4_supported(H, W);
  if (B <= 0 || Ci <= 0 || Co <= 0 || !sup || B % sivae_conv2d_wino4_images_per_item(H, W)) return SIVAE_ERR_SHAPE;
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("SIVAE_WINO4_SPLITK");
    enabled = (e && e[0] == '0') ? 0 : 1;
  }
  if (!enabled || Ci < 16) return 1;
  const long long items = w4_px_tiles(B, H, W) * ((Co + W4_TCO - 1) / W4_TCO);
  const int cus = sivae_num_cus();
  if (items >= cus) return 1;
  const int nchunks = w4_kpad(Ci) / W4_CK;
  int S = 1;
  // powers of two up to 8: S x items <= one block per CU, at least 8 chunks (64 input channels) per slice and an EVEN
  // number of chunks per slice (the kernel's K loop is unrolled by chunk pairs; wino4_impl rejects odd counts — padded
  // channel counts of 288 / 352 / 416 / 480 would otherwise pick S = 4 with 9 / 11 / 13 / 15 chunks per slice)
  while (S < 8 && (long long)(2 * S) * items <= cus && nchunks % (2 * S) == 0 && nchunks / (2 * S) >= 8 &&
         ((nchunks / (2 * S)) & 1) == 0)
    S *= 2;
  return S;
}

// Does the image-grid form of the 8 x 8 / 4 x 4 maps (modes 3, 4) beat F(2x2,3x3) for this launch?  Measured per (batch,
// channels) against conv_wino.hip's kernel (tools/bench_wino4_small.py, profiles/r6_wino4_small_maps_vs_f23.txt): its work
// item is 64 output channels x 32 tiles with ~25 us of fixed cost around 2.6 us per 8-channel chunk, so it wins — 1.1-1.75x
// — where (slices x items) fill the chip AND a K slice is long enough; below that F(2x2,3x3) with its 4x smaller items is
// up to 2x faster.  8 x 8: >= 256 channels per slice, or an unsplit launch with >= 128; 4 x 4 (where F(2x2,3x3) wastes
// half of every tile column it transforms): >= 64 channels per slice.
extern "C" int sivae_conv2d_wino4_small_pays(int B, int Ci, int Co, int H, int W) {
  const int sup = sivae_conv2d_wino4_supported(H, W);
  if (sup < 3 || B <= 0 || Ci < 16 || Co <= 0 || B % sivae_conv2d_wino4_images_per_item(H, W)) return 0;
  const int S = sivae_conv2d_wino4_splitk(B, Ci, Co, H, W);
  if (S < 1) return 0;
  const long long items = w4_px_tiles(B, H, W) * ((Co + W4_TCO - 1) / W4_TCO);
  if (items * S < sivae_num_cus()) return 0;
  const int per_slice = w4_kpad(Ci) / S;
  if (sup == 3) return (per_slice >= 256 || (S == 1 && per_slice >= 128)) ? 1 : 0;
  return per_slice >= 64 ? 1 : 0;
}

extern "C" size_t sivae_conv2d_wino4_splitk_workspace_bytes(int B, int Ci, int Co, int H, int W) {
  const int S = sivae_conv2d_wino4_splitk(B, Ci, Co, H, W);
  if (S <= 1) return 0;
  return (size_t)S * B * Co * H * W * sizeof(float);
}

// y (+)= conv3x3(x', U) as sivae_conv2d_wino4_fwd_pro (pro_mean may be NULL: no prologue), split over K when
// sivae_conv2d_wino4_splitk(...) > 1: stats_partial is then [B][Co][2] (per image)
extern "C" int sivae_conv2d_wino4_fwd_splitk(const float* x, const float* up, float* y, const float* pro_mean,
                                             const float* pro_invstd, const float* pro_gamma, const float* pro_beta,
                                             float pro_slope, float* stats_partial, int B, int Ci, int Co, int H, int W,
                                             int accumulate, int seg_images, void* workspace, size_t workspace_bytes,
                                             hipStream_t stream) {
  const int S = sivae_conv2d_wino4_splitk(B, Ci, Co, H, W);
  if (S < 0) return S;
  if (S == 1)
    return wino4_impl(x, up, y, pro_mean, pro_invstd, pro_gamma, pro_beta, pro_slope, stats_partial, B, Ci, Co, H, W,
                      accumulate, seg_images, stream);
  if (!y || !workspace) return SIVAE_ERR_NULL;
  if (workspace_bytes < (size_t)S * B * Co * H * W * sizeof(float)) return SIVAE_ERR_WORKSPACE;
  if (((uintptr_t)workspace & 15u) != 0) return SIVAE_ERR_SHAPE;
  float* part = reinterpret_cast<float*>(workspace);
  const int rc = wino4_impl(x, up, part, pro_mean, pro_invstd, pro_gamma, pro_beta, pro_slope, nullptr, B, Ci, Co, H, W, 0,
                            seg_images, stream, S);
  if (rc != SIVAE_OK) return rc;
  const int HW = H * W;
  const size_t slice = (size_t)B * Co * HW;
  if ((HW & 3) == 0 && ((uintptr_t)y & 15u) == 0 && S <= 8) {
    const float4* p4 = reinterpret_cast<const float4*>(part);
    float4* y4 = reinterpret_cast<float4*>(y);
    if (HW <= 256)
      hipLaunchKernelGGL(wino4_splitk_reduce_vec_kernel<64>, dim3((unsigned)((B * Co + 3) / 4)), dim3(256), 0, stream, p4, y4,
                         stats_partial, S, HW / 4, slice / 4, accumulate, B * Co);
    else
      hipLaunchKernelGGL(wino4_splitk_reduce_vec_kernel<256>, dim3((unsigned)(B * Co)), dim3(256), 0, stream, p4, y4,
                         stats_partial, S, HW / 4, slice / 4, accumulate, B * Co);
  } else {
    hipLaunchKernelGGL(wino4_splitk_reduce_kernel, dim3((unsigned)(B * Co)), dim3(64), 0, stream, part, y, stats_partial, S,
                       HW, slice, accumulate);
  }
  return sivae_launch_status();
}

// ---- batched packing (pack_batch.h)
int sivae_packjob_wino4(SivaePackJob* j, int Co, int Ci, int mode) {
  j->kdim = mode == 0 ? Ci : Co;
  j->ndim = mode == 0 ? Co : Ci;
  j->kpad = w4_kpad(j->kdim);
  j->npad = w4_npad(j->ndim);
  j->total = (unsigned long long)j->kpad * j->npad;
  return SIVAE_OK;
}
void sivae_packbatch_wino4(const SivaePackJob* jobs, const unsigned short* block_job, int nblocks, hipStream_t s) {
  hipLaunchKernelGGL(pack_wino4_batch_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, jobs, block_job);
}
