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
  // split-K (launches with fewer work items than CUs): an item = (pixel tile, K slice, co tile); slice ks walks the cps
  // chunks from ks * cps on and writes its partial outputs to y + ks * slice_stride (summed by wino4_splitk_reduce_kernel)
  int ksl, cps;
  long long slice_stride;
  // image-grid form (conv_wino4_grid_kernel; behind the round-5 fields so that their offsets stay what they were)
  int iw_l2, ih_l2;
  unsigned long long seam_c0, seam_c5, seam_r0, seam_r5;
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

#define W4_POOL 0
#define W4_GRID 0
#define W4_KERNEL_NAME conv_wino4_kernel
#define W4_TWO a.two
#define W4_IPI (a.two ? 2 : 1)
#include "conv_wino4_kernel.inc"
#undef W4_GRID
#undef W4_KERNEL_NAME
#undef W4_TWO
#undef W4_IPI
#define W4_GRID 1
#define W4_KERNEL_NAME conv_wino4_grid_kernel
#define W4_TWO true
#define W4_IPI two_ipi
#include "conv_wino4_kernel.inc"
#undef W4_GRID
#undef W4_KERNEL_NAME
#undef W4_TWO
#undef W4_IPI
#undef W4_POOL
#define W4_POOL 1
#define W4_GRID 0
#define W4_KERNEL_NAME conv_wino4_pool_kernel
#define W4_TWO false
#define W4_IPI 1
#include "conv_wino4_kernel.inc"
#undef W4_GRID
#undef W4_KERNEL_NAME
#undef W4_TWO
#undef W4_IPI
#undef W4_POOL

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
                      int Co, int H, int W, int accumulate, int seg_images, hipStream_t stream, int ksl = 1,
                      bool pool = false) {
  if (!x || !up || !y) return SIVAE_ERR_NULL;
  if (pro_mean && (!pro_invstd || !pro_gamma || !pro_beta)) return SIVAE_ERR_NULL;
  if (pro_mean && !(pro_slope >= 0.f && pro_slope <= 1.f)) return SIVAE_ERR_MODE;  // prologue uses max(v, v*slope)
  if (seg_images < 0 || (seg_images > 0 && B % seg_images != 0)) return SIVAE_ERR_SHAPE;
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  const int sup = sivae_conv2d_wino4_supported(H, W);
  if (!sup) return SIVAE_ERR_SHAPE;
  const int ipi = sivae_conv2d_wino4_images_per_item(H, W);
  if ((B % ipi) || (seg_images % ipi)) return SIVAE_ERR_SHAPE;  // whole image grids, each inside one segment
  if (!pool && ((uintptr_t)y & 15u) != 0) return SIVAE_ERR_SHAPE;  // 16-byte stores
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
  // One co tile (Co <= 64) too: a halo row of an item is 160 bytes from 16 bytes in front of a 128-byte line, i.e. THREE
  // lines of the L2 — with consecutive items (horizontal neighbours) round-robin over the XCDs every L2 fetches all three
  // (3 x 18/16 = 3.4 reads per input byte: the 64-channel 256 x 256 layers moved 4.7 TB/s through the fabric); grouped, the 32
  // blocks of an XCD hold a 64-row band of one image at the same time and the edge lines are fetched once.
  // (SIVAE_W4_XCD_SINGLE=0: the round-5 order, A/B switch)
  static const int single_ok = [] {
    const char* e = getenv("SIVAE_W4_XCD_SINGLE");
    return (e && e[0] == '0') ? 0 : 1;
  }();
  a.xcd_group = (sivae_xcd_remap() && (a.n_co_tiles > 1 || single_ok) && !(grid & 7)) ? 1 : 0;
  if (pool) {  // (whole-tile maps only, no prologue / statistics / split-K: checked by the caller)
    hipLaunchKernelGGL(conv_wino4_pool_kernel<false>, dim3((unsigned)grid), dim3(W4_NT), 0, stream, a);
  } else if (sup >= 3) {  // (the 16 x 16 pairs, mode 2, stay on the round-5 kernel: a.two)
    if (pro_mean)
      hipLaunchKernelGGL(conv_wino4_grid_kernel<true>, dim3((unsigned)grid), dim3(W4_NT), 0, stream, a);
    else
      hipLaunchKernelGGL(conv_wino4_grid_kernel<false>, dim3((unsigned)grid), dim3(W4_NT), 0, stream, a);
  } else if (pro_mean) {
    hipLaunchKernelGGL(conv_wino4_kernel<true>, dim3((unsigned)grid), dim3(W4_NT), 0, stream, a);
  } else {
    hipLaunchKernelGGL(conv_wino4_kernel<false>, dim3((unsigned)grid), dim3(W4_NT), 0, stream, a);
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

// ---- data gradient of conv3x3(Upsample2(x)) with respect to the LOW-resolution x (round 6; conv1 behind nn.Upsample,
// train_soft_intro_vae.py:155,56): dx[B][N][H/2][W/2] (+)= blocksum2x2( conv3x3^T(dy[B][C][H][W]) ) in one F(4x4,3x3) pass with
// the block sum folded into the output transform (conv_wino4_pool_kernel).  `up`: sivae_pack_wino4_weight(w, mode 1) of the
// conv's weight w[C][N][3][3].  Maps: H % 16 == 0, W % 32 == 0 (mode 1 of sivae_conv2d_wino4_supported).
// `_pays`: the launches where it beats the phase-folded F(2x2,2x2) kernel (conv_wino_up_dgrad.hip) — that kernel needs 128
// output channels per block and splits K over wave pairs below (N <= 64: 0.47 matrix-pipe busy against this kernel's 0.6),
// so: N <= 64 and at least one work item per CU.
extern "C" int sivae_conv2d_wino4_dgrad_pool_pays(int B, int C, int N, int H, int W) {
  if (B <= 0 || C < 16 || N <= 0 || N > 64 || sivae_conv2d_wino4_supported(H, W) != 1) return 0;
  return w4_px_tiles(B, H, W) * ((N + W4_TCO - 1) / W4_TCO) >= sivae_num_cus() ? 1 : 0;
}
extern "C" int sivae_conv2d_wino4_dgrad_pool(const float* dy, const float* up, float* dx, int B, int C, int N, int H,
                                             int W, int accumulate, hipStream_t stream) {
  if (!dy || !up || !dx) return SIVAE_ERR_NULL;
  if (sivae_conv2d_wino4_supported(H, W) != 1) return SIVAE_ERR_SHAPE;
  if (((uintptr_t)dx & 7u) != 0) return SIVAE_ERR_SHAPE;  // 8-byte stores
  return wino4_impl(dy, up, dx, nullptr, nullptr, nullptr, nullptr, 1.f, nullptr, B, C, N, H, W, accumulate, 0, stream, 1,
                    true);
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

// The same for the 8 x 8 / 4 x 4 maps of the image-grid form (round 6): a plane is only LPP = 16 / 4 vectors, so a wave of the
// kernel above would run with 16 / 4 active lanes (21 launches of 25 us per cifar10 iteration).  Here consecutive threads
// take consecutive vectors — 64 / LPP planes per wave — and the per-plane {sum, sumsq} are folded inside aligned groups of
// LPP lanes.
template <int LPP>
__global__ void __launch_bounds__(256) wino4_splitk_reduce_small_kernel(const float4* __restrict__ part, float4* __restrict__ y,
                                                                        float* __restrict__ stats, int S, size_t slice_stride4,
                                                                        int accumulate, int n_planes) {
  const size_t gt = (size_t)blockIdx.x * 256 + threadIdx.x;  // vector index = plane * LPP + vector of the plane
  if (gt >= (size_t)n_planes * LPP) return;                   // (whole LPP-groups leave together: LPP divides 256)
  float4 t[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (k < S) t[k] = part[(size_t)k * slice_stride4 + gt];
  float4 v = accumulate ? y[gt] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (k < S) {
      v.x += t[k].x;
      v.y += t[k].y;
      v.z += t[k].z;
      v.w += t[k].w;
    }
  y[gt] = v;
  if (stats != nullptr) {
    float s = (v.x + v.y) + (v.z + v.w);
    float q = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
#pragma unroll
    for (int m = 1; m < LPP; m <<= 1) {
      s += __shfl_xor(s, m);
      q += __shfl_xor(q, m);
    }
    if ((gt & (LPP - 1)) == 0) {
      stats[(gt / LPP) * 2 + 0] = s;
      stats[(gt / LPP) * 2 + 1] = q;
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
    if (HW == 16 || HW == 64) {
      const size_t nvec = (size_t)B * Co * (HW / 4);
      if (HW == 16)
        hipLaunchKernelGGL(wino4_splitk_reduce_small_kernel<4>, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, stream, p4, y4,
                           stats_partial, S, slice / 4, accumulate, B * Co);
      else
        hipLaunchKernelGGL(wino4_splitk_reduce_small_kernel<16>, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, stream, p4, y4,
                           stats_partial, S, slice / 4, accumulate, B * Co);
    } else if (HW <= 256)
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
