// Helpers shared by the bf16-storage kernels (config 3's build-defined mixed-precision mode, gfx950 only).
//
// Activation layout of the bf16 mode ("blocked NCHW", NC8HW8):   x[b][c / 8][h][w][c % 8]   bf16
// i.e. every pixel of a plane carries 8 consecutive channels in one 16-byte vector.  This is the layout the bf16
// matrix pipe wants: v_mfma_f32_32x32x16_bf16 takes 8 consecutive K values per lane, K = input channels for the
// forward / data-gradient GEMMs, so an operand is ONE ds_read_b128, and every filter tap is a 16-byte-aligned shift.
// The channel count is padded to a multiple of 16 (one MFMA k-step = two 8-channel blocks); padded channels hold 0.
// Planes stay contiguous per 8-channel block, so the per-channel BatchNorm passes stream whole planes with 16-byte
// accesses exactly like NCHW.
#pragma once
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

static inline int bf16_cblocks(int C) { return ((C + 15) / 16) * 2; }  // 8-channel blocks in storage (even)

#ifdef __HIPCC__
// two floats -> packed bf16 pair (round to nearest even: v_cvt_pk_bf16_f32), low half = a
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  bf16x2_t v;
  v[0] = (__bf16)a;
  v[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ float bf16_round(float v) { return bf16_lo(pack_bf16(v, 0.f)); }

// 16-byte vector of 8 bf16 <-> 8 floats
__device__ __forceinline__ void unpack8(const u32x4_t q, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = bf16_lo(q[i]);
    f[2 * i + 1] = bf16_hi(q[i]);
  }
}
__device__ __forceinline__ u32x4_t pack8(const float* f) {
  u32x4_t q;
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = pack_bf16(f[2 * i], f[2 * i + 1]);
  return q;
}
__device__ __forceinline__ u32x4_t buf_load_u32x4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
}
#endif
