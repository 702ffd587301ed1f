// Shared device/host helpers for libsivae_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

// ---- error codes (negative = argument errors defined by this library, positive = hipError_t) ----
#define SIVAE_OK 0
#define SIVAE_ERR_NULL -1       // a required pointer is null
#define SIVAE_ERR_SHAPE -2      // unsupported / inconsistent shape
#define SIVAE_ERR_KSIZE -3      // kernel size not in {1,3,5}
#define SIVAE_ERR_WORKSPACE -4  // workspace too small
#define SIVAE_ERR_RANGE -5      // tensor too large for 32-bit element indexing
#define SIVAE_ERR_MODE -6       // unknown mode / flag value

#define SIVAE_ABI_VERSION 1

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int sivae_launch_status() { return (int)hipGetLastError(); }

static inline int ilog2_exact(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
static inline int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Tile geometry shared by the implicit-GEMM conv kernels: a pixel tile is TB images x TH rows x TW
// cols (all powers of two, TB > 1 only when one tile covers a whole image).
struct TileGeom {
  int tb_log2, th_log2, tw_log2;
  int ntb, nth, ntw;  // number of tiles along batch / rows / cols
};

static inline TileGeom make_tile_geom(int B, int H, int W, int tpx /* pixels per tile, pow2 */) {
  TileGeom g;
  int tw = next_pow2(W);
  if (tw > 32) tw = 32;
  if (tw > tpx) tw = tpx;
  int th = next_pow2(H);
  if (th > tpx / tw) th = tpx / tw;
  int tb = tpx / (tw * th);
  g.tw_log2 = ilog2_exact(tw);
  g.th_log2 = ilog2_exact(th);
  g.tb_log2 = ilog2_exact(tb);
  g.ntw = cdiv(W, tw);
  g.nth = cdiv(H, th);
  g.ntb = cdiv(B, tb);
  return g;
}

// compute units of the current device (cached; 256 on an MI355X) — the persistent kernels size their grids by it
static inline int sivae_num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
  }
  return cus;
}

#ifdef __HIPCC__
// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E) — the index is usable in
// `constexpr` / `if constexpr`, which keeps register arrays statically indexed in unrolled pipelines
#include <type_traits>
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// ---- wave64 reductions (butterfly over all 64 lanes; every lane ends with the total) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over the 32 lanes that share (lane >> 5)
__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// The same sum in five DPP-modified VALU adds (no ds_bpermute traffic, no lgkmcnt waits): quad butterflies, the two
// row mirrors, then row_bcast15 folds row 0 into row 1 (and row 2 into row 3).  The total is valid ONLY in the
// upper 16 lanes of each half wave (lane & 16) — the conv epilogues let lane 31 of the half write it.
// Must be called with the whole wave active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float half_wave_sum_hi(float v) {
  v += dpp_mov_f32<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov_f32<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov_f32<0x141, 0xf>(v);  // row_half_mirror
  v += dpp_mov_f32<0x140, 0xf>(v);  // row_mirror
  v += dpp_mov_f32<0x142, 0xa>(v);  // row_bcast15 into rows 1 and 3
  return v;
}

// Transposing sum of N (a multiple of 32) per-lane values over the 32 lanes that share (lane >> 5): on return v[i], i < N / 32,
// is the total over those lanes of the caller's v[i * 32 + (lane & 31)] — N - N / 32 adds per lane instead of the 5 N of N
// separate butterflies.  Step k pairs lane l with l ^ (1 << k): the lane whose bit k is clear keeps the even value of
// every pair and sends the odd one (quad permutes for bits 0 / 1, ds_swizzle for bit 2 and 4, row_ror:8 for bit 3).
// Must be called with the whole wave active.
template <int N>
__device__ __forceinline__ void lanes32_transpose_sum(float (&v)[N], int lane) {
  static_assert(N % 32 == 0, "lanes32_transpose_sum: N must be a multiple of 32");
#define SIVAE_T32_STEP(CNT, BIT, RECV)                                       \
  {                                                                          \
    const bool sel_ = (lane >> (BIT)) & 1;                                   \
    _Pragma("unroll") for (int i = 0; i < (CNT) / 2; ++i) {                  \
      const float keep_ = sel_ ? v[2 * i + 1] : v[2 * i];                    \
      const float send_ = sel_ ? v[2 * i] : v[2 * i + 1];                    \
      v[i] = keep_ + RECV(send_);                                            \
    }                                                                        \
  }
#define SIVAE_T32_Q1(X) dpp_mov_f32<0xB1, 0xf>(X)
#define SIVAE_T32_Q2(X) dpp_mov_f32<0x4E, 0xf>(X)
#define SIVAE_T32_X4(X) __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, X), 0x101f))
#define SIVAE_T32_R8(X) dpp_mov_f32<0x128, 0xf>(X)
#define SIVAE_T32_X16(X) __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, X), 0x401f))
  SIVAE_T32_STEP(N, 0, SIVAE_T32_Q1)
  SIVAE_T32_STEP(N / 2, 1, SIVAE_T32_Q2)
  SIVAE_T32_STEP(N / 4, 2, SIVAE_T32_X4)
  SIVAE_T32_STEP(N / 8, 3, SIVAE_T32_R8)
  SIVAE_T32_STEP(N / 16, 4, SIVAE_T32_X16)
#undef SIVAE_T32_STEP
#undef SIVAE_T32_Q1
#undef SIVAE_T32_Q2
#undef SIVAE_T32_X4
#undef SIVAE_T32_R8
#undef SIVAE_T32_X16
}

// 8 per-lane values summed over the whole wave: on return v[0] of lane l is the total of the caller's v[l & 7] (the same
// transposing steps over lane bits 0-2, then the eight 8-lane groups are added): 7 select-adds + 3 adds instead of 8 six-step
// ds_bpermute butterflies.  Must be called with the whole wave active.
__device__ __forceinline__ void wave_transpose_sum8(float (&v)[8], int lane) {
#define SIVAE_T8_STEP(CNT, BIT, RECV)                                        \
  {                                                                          \
    const bool sel_ = (lane >> (BIT)) & 1;                                   \
    _Pragma("unroll") for (int i = 0; i < (CNT) / 2; ++i) {                  \
      const float keep_ = sel_ ? v[2 * i + 1] : v[2 * i];                    \
      const float send_ = sel_ ? v[2 * i] : v[2 * i + 1];                    \
      v[i] = keep_ + RECV(send_);                                            \
    }                                                                        \
  }
#define SIVAE_T8_Q1(X) dpp_mov_f32<0xB1, 0xf>(X)
#define SIVAE_T8_Q2(X) dpp_mov_f32<0x4E, 0xf>(X)
#define SIVAE_T8_X4(X) __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, X), 0x101f))
  SIVAE_T8_STEP(8, 0, SIVAE_T8_Q1)
  SIVAE_T8_STEP(4, 1, SIVAE_T8_Q2)
  SIVAE_T8_STEP(2, 2, SIVAE_T8_X4)
  v[0] += dpp_mov_f32<0x128, 0xf>(v[0]);                                                                   // lane ^ 8 (row_ror:8)
  v[0] += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v[0]), 0x401f));  // lane ^ 16
  v[0] += __shfl_xor(v[0], 32, 64);
#undef SIVAE_T8_STEP
#undef SIVAE_T8_Q1
#undef SIVAE_T8_Q2
#undef SIVAE_T8_X4
}

// 16 per-lane values summed over each 16-lane ROW of the wave: on return v[0] of lane l is the total, over the row of l, of
// the caller's v[l & 15] (the transposing steps over lane bits 0-3).  Must be called with the whole wave active.
__device__ __forceinline__ void row16_transpose_sum16(float (&v)[16], int lane) {
#define SIVAE_R16_STEP(CNT, BIT, RECV)                                       \
  {                                                                          \
    const bool sel_ = (lane >> (BIT)) & 1;                                   \
    _Pragma("unroll") for (int i = 0; i < (CNT) / 2; ++i) {                  \
      const float keep_ = sel_ ? v[2 * i + 1] : v[2 * i];                    \
      const float send_ = sel_ ? v[2 * i] : v[2 * i + 1];                    \
      v[i] = keep_ + RECV(send_);                                            \
    }                                                                        \
  }
#define SIVAE_R16_Q1(X) dpp_mov_f32<0xB1, 0xf>(X)
#define SIVAE_R16_Q2(X) dpp_mov_f32<0x4E, 0xf>(X)
#define SIVAE_R16_X4(X) __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, X), 0x101f))
#define SIVAE_R16_R8(X) dpp_mov_f32<0x128, 0xf>(X)
  SIVAE_R16_STEP(16, 0, SIVAE_R16_Q1)
  SIVAE_R16_STEP(8, 1, SIVAE_R16_Q2)
  SIVAE_R16_STEP(4, 2, SIVAE_R16_X4)
  SIVAE_R16_STEP(2, 3, SIVAE_R16_R8)
#undef SIVAE_R16_STEP
#undef SIVAE_R16_Q1
#undef SIVAE_R16_Q2
#undef SIVAE_R16_X4
#undef SIVAE_R16_R8
}

// Block-wide sum of doubles for blocks of NT threads (NT multiple of 64, <= 1024).
// `red` must hold NT/64 doubles of LDS. Result valid in every thread.
template <int NT>
__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) t += red[i];
  return t;
}

// Two block-wide sums at once: the same per-value order as two block_sum calls (butterfly inside a wave, waves in index
// order) — bit-identical results —, but the cross-lane steps of the two values overlap and one barrier pair serves both.
// `red` must hold 2 * NT/64 doubles.
template <int NT>
__device__ __forceinline__ void block_sum2(double& a, double& b, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ta = __shfl_xor(a, o, 64), tb = __shfl_xor(b, o, 64);
    a += ta;
    b += tb;
  }
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    red[wave] = a;
    red[NT / 64 + wave] = b;
  }
  __syncthreads();
  double ua = 0.0, ub = 0.0;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) {
    ua += red[i];
    ub += red[NT / 64 + i];
  }
  a = ua;
  b = ub;
}

// ---- out[e] = sum_s part[s][e] for split-K style workspaces ([n_slices][numel] floats) ----
// A block owns 64 consecutive elements; its YL wave-rows each walk every YL-th slice with four independent
// accumulators (so 4*YL loads per element are in flight instead of one), then the rows are folded through LDS in
// row order.  The summation order depends only on (n_slices, YL) -> deterministic.
template <int YL>
__global__ void __launch_bounds__(64 * YL) slice_reduce_rows_kernel(const float* __restrict__ part,
                                                                    float* __restrict__ out, int n_slices,
                                                                    size_t numel) {
  __shared__ float red[YL][64];
  const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
  const size_t i = (size_t)blockIdx.x * 64 + x;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < numel) {
    const float* p = part + i;
    int k = y;
    for (; k + 3 * YL < n_slices; k += 4 * YL) {
      s0 += p[(size_t)k * numel];
      s1 += p[(size_t)(k + YL) * numel];
      s2 += p[(size_t)(k + 2 * YL) * numel];
      s3 += p[(size_t)(k + 3 * YL) * numel];
    }
    for (; k < n_slices; k += YL) s0 += p[(size_t)k * numel];
  }
  red[y][x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (y == 0 && i < numel) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < YL; ++j) s += red[j][x];
    out[i] = s;
  }
}

static inline void sivae_launch_slice_reduce(const float* part, float* out, int n_slices, size_t numel,
                                             hipStream_t stream) {
  const unsigned nb = (unsigned)((numel + 63) / 64);
  if (n_slices >= 64)
    hipLaunchKernelGGL((slice_reduce_rows_kernel<16>), dim3(nb), dim3(1024), 0, stream, part, out, n_slices, numel);
  else if (n_slices >= 8)
    hipLaunchKernelGGL((slice_reduce_rows_kernel<4>), dim3(nb), dim3(256), 0, stream, part, out, n_slices, numel);
  else
    hipLaunchKernelGGL((slice_reduce_rows_kernel<1>), dim3(nb), dim3(64), 0, stream, part, out, n_slices, numel);
}

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }
// same function for 0 <= slope <= 1 in two VALU ops (v_mul + v_max) instead of compare / multiply / select
__device__ __forceinline__ float lrelu01(float v, float slope) { return fmaxf(v, v * slope); }

// ---- raw buffer loads: wave-uniform 128-bit descriptor + 32-bit per-lane BYTE offset + uniform
// SGPR offset.  A per-lane offset >= num_records returns 0 without touching memory, which is how
// every halo / out-of-image / out-of-batch element of a conv tile becomes a zero for free
// (no per-element branches, no exec masking).  The SGPR offset is NOT range-checked.
#define SIVAE_OOB 0xFFFFFFFFu
// out-of-range marker for 16-BYTE buffer accesses: at 0xFFFFFFFF only the first dword of a dwordx4 store is dropped — the
// offsets of dwords 1..3 wrap around 2^32 into the window (measured on gfx950, round 4).  At this offset none of the 16
// bytes wraps; every window it is used with is checked to be smaller.
#define SIVAE_OOB16 0xFFFFFFF0u
// Store-data lifetime of 16-byte buffer stores (round 5).  Observed once on gfx950 with the memory pipe saturated
// (bn_fused.hip, round 4): a buffer_store_dwordx4 whose data registers were rewritten a few issue slots later stored the NEW
// values in the last quad of each 16-lane row.  Kernels that store from short-lived temporaries pin the data of store n
// (an empty asm that "reads" it) behind the issue of store n + 1, so the register allocator cannot hand those registers
// to anything in between; kernels with a dead payload register form the data in place (bn_fused.hip::BF_KEEP).
#define SIVAE_PIN4(V) asm volatile("" ::"v"((V)[0]), "v"((V)[1]), "v"((V)[2]), "v"((V)[3]));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned long long bytes) {
  const unsigned n = bytes > 0xFFFFFFFEull ? 0xFFFFFFFEu : (unsigned)bytes;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)n, 0x00020000);
}
__device__ __forceinline__ float buf_load_f32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ float4 buf_load_f32x4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  // NOTE (hipcc / ROCm 7.2): take the result as a u32x4 and bit-cast the WHOLE vector. Per-element
  // `__builtin_bit_cast(float, q[i])` (or an `auto` result) makes the compiler emit a ONE-dword load
  // and leaves the other three lanes of the result undefined — verify `buffer_load_dwordx4` in the .s.
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  const f32x4 f = __builtin_bit_cast(f32x4, q);
  return make_float4(f[0], f[1], f[2], f[3]);
}

// raw buffer stores / 8-byte loads with the same addressing rules (an out-of-range per-lane offset drops the
// store / returns 0) — lets epilogues mask invalid lanes through the offset instead of exec-mask branches
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void buf_store_f32x2(__amdgpu_buffer_rsrc_t r, float a, float b, unsigned voff, unsigned soff) {
  f32x2 v;
  v[0] = a;
  v[1] = b;
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ float2 buf_load_f32x2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const u32x2_t q = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
  const f32x2 f = __builtin_bit_cast(f32x2, q);
  return make_float2(f[0], f[1]);
}

// exact floor(e / d) for e < 2^22, d < 2^10 with magic = floor(2^32 / d) + 1
// magic == 0 encodes d == 1
__device__ __forceinline__ unsigned fastdiv(unsigned e, unsigned magic) { return magic ? __umulhi(e, magic) : e; }
#endif

// SIVAE_XCD_REMAP=0 switches the XCD-aware block order of the weight-gradient kernels off (A/B switch)
static inline int sivae_xcd_remap() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("SIVAE_XCD_REMAP");
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v;
}

// Raise a kernel's dynamic-LDS limit once per high-water mark instead of on every launch: keeps the attribute call
// out of the launch path and, after an eager warm-up has seen the kernel, out of HIP-graph stream captures.
static inline int sivae_ensure_lds(const void* kern, size_t lds, size_t* high_water) {
  if (lds <= 64 * 1024 || lds <= *high_water) return SIVAE_OK;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  *high_water = lds;
  return SIVAE_OK;
}

static inline unsigned make_magic(unsigned d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull / d) + 1ull); }
