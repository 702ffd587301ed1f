// The two nn.Linear layers (Encoder.fc train_soft_intro_vae.py:109,121 / Decoder.fc :146,166) as small-M GEMMs on the
// exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32).  M = batch shard (<= 256 rows), so the work is WEIGHT-STREAMING
// bound (8192 x 1024 fp32 = 33.5 MB read once) and parallelism has to come from the N / K dimensions: the 1x1-conv
// kernel these layers used before gave them cdiv(N,128) = 4..8 blocks walking K = 8192 serially (0.43-0.95 ms per call,
// 9 % of a 16-image iteration); here every wave owns one 32-wide output column strip for ALL batch rows and a slice of
// the contraction, ~1000 waves per call, partial sums reduced in a fixed order (deterministic, no atomics).
//
//   forward   y[b][n]  = sum_k x[b][k] W[n][k] + bias[n]   (optionally ReLU)      contraction K, split S ways
//   dgrad     dx[b][k] = sum_n dy[b][n] W[n][k]                                   contraction N, split S ways
//   wgrad     dW[n][k] = sum_b dy[b][n] x[b][k]                                   contraction B (short): no split
//
// Operand fetch: both operands of the forward GEMM are k-contiguous, so a lane (row/col i = lane & 31, half h = lane >> 5)
// loads ONE 16-byte vector per operand per 8 k values (k = kb + 4h .. 4h + 3) and issues four MFMAs, the e-th using
// element e of each vector — the MFMA sums over k in whatever order the two operands share.  Rows past the batch /
// columns past N carry an out-of-range buffer offset and read 0.
#include "common.h"

namespace {

constexpr int MAX_RB = 8;  // batch rows <= 256

struct LinArgs {
  const float* a;     // forward: x [B][K];  dgrad: dy [B][N];  wgrad: dy [B][N]
  const float* w;     // forward / dgrad: W [N][K];  wgrad: x [B][K]
  float* out;         // partial [S][B][cols] or the final tensor (S == 1, and wgrad)
  const float* bias;  // forward, S == 1 only
  int B, K, N;
  int S, slice_len;   // contraction slice per wave (multiple of 8)
  int relu;
};

// ---- forward: out[s][b][n] (or y when S == 1)
template <int RB>
__global__ void __launch_bounds__(64) linear_fwd_kernel(LinArgs a) {
  const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
  const int n_tiles = (a.N + 31) / 32;
  const int nt = blockIdx.x % n_tiles, s = blockIdx.x / n_tiles;
  const int n0 = nt * 32;
  const int k_begin = s * a.slice_len;
  int k_end = k_begin + a.slice_len;
  if (k_end > a.K) k_end = a.K;
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(a.a, (unsigned long long)a.B * a.K * 4ull);
  const __amdgpu_buffer_rsrc_t wr = make_rsrc(a.w, (unsigned long long)a.N * a.K * 4ull);
  unsigned xo[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) xo[r] = (r * 32 + i < a.B) ? (unsigned)((r * 32 + i) * a.K + 4 * h) * 4u : SIVAE_OOB;
  const unsigned wo = (n0 + i < a.N) ? (unsigned)((n0 + i) * a.K + 4 * h) * 4u : SIVAE_OOB;
  f32x16 acc[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
  // two 8-wide k groups per step, software-pipelined (round 6): group 1 of step k is requested before the MFMAs of group 0,
  // group 0 of step k + 16 before the MFMAs of group 1 — a launch is about one wave per SIMD, so nothing else covered the
  // memory round trip of a step.  Same registers, same order of the sum over k.
  // (K % 4 == 0, so a lane's 4-wide vector is entirely inside or entirely past the slice end)
  float4 w0, w1, x0[RB], x1[RB];
#define LIN_LOAD(KK, W, X)                                                     \
  {                                                                            \
    const bool v_ = (KK) + 4 * h < k_end;                                      \
    W = buf_load_f32x4(wr, v_ ? wo : SIVAE_OOB, (unsigned)(KK) * 4u);          \
    _Pragma("unroll") for (int r = 0; r < RB; ++r) X[r] = buf_load_f32x4(xr, v_ ? xo[r] : SIVAE_OOB, (unsigned)(KK) * 4u); \
  }
#define LIN_MMA(W, X)                                                          \
  _Pragma("unroll") for (int r = 0; r < RB; ++r) {                             \
    acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(X[r].x, W.x, acc[r], 0, 0, 0); \
    acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(X[r].y, W.y, acc[r], 0, 0, 0); \
    acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(X[r].z, W.z, acc[r], 0, 0, 0); \
    acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(X[r].w, W.w, acc[r], 0, 0, 0); \
  }
  LIN_LOAD(k_begin, w0, x0)
  for (int k = k_begin; k < k_end; k += 16) {
    LIN_LOAD(k + 8, w1, x1)
    __builtin_amdgcn_sched_barrier(0);
    LIN_MMA(w0, x0)
    __builtin_amdgcn_sched_barrier(0);
    LIN_LOAD(k + 16, w0, x0)  // (past the slice end: out-of-range offsets, zeros, never used)
    __builtin_amdgcn_sched_barrier(0);
    LIN_MMA(w1, x1)
    __builtin_amdgcn_sched_barrier(0);
  }
#undef LIN_LOAD
#undef LIN_MMA
  // accumulator e of a lane: row (e&3) + 8*(e>>2) + 4*h (batch row within the tile), column i (= n0 + i)
  const int n = n0 + i;
  if (n >= a.N) return;
  float* dst = a.out + (size_t)s * a.B * a.N;
  const float bias = (a.S == 1 && a.bias) ? a.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int b = r * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (b < a.B) {
        float v = acc[r][e] + bias;
        if (a.S == 1 && a.relu) v = fmaxf(v, 0.f);
        dst[(size_t)b * a.N + n] = v;
      }
    }
}

// ---- dgrad: out[s][b][k]; A = dy (n-contiguous vectors), B = W rows (one dword per n, lanes along k)
template <int RB>
__global__ void __launch_bounds__(64) linear_dgrad_kernel(LinArgs a) {
  const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
  const int k_tiles = (a.K + 31) / 32;
  const int kt = blockIdx.x % k_tiles, s = blockIdx.x / k_tiles;
  const int k0 = kt * 32;
  const int n_begin = s * a.slice_len;
  int n_end = n_begin + a.slice_len;
  if (n_end > a.N) n_end = a.N;
  const __amdgpu_buffer_rsrc_t dr = make_rsrc(a.a, (unsigned long long)a.B * a.N * 4ull);
  const __amdgpu_buffer_rsrc_t wr = make_rsrc(a.w, (unsigned long long)a.N * a.K * 4ull);
  unsigned d_o[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) d_o[r] = (r * 32 + i < a.B) ? (unsigned)((r * 32 + i) * a.N + 4 * h) * 4u : SIVAE_OOB;
  const unsigned wo = (k0 + i < a.K) ? (unsigned)(4 * h * a.K + k0 + i) * 4u : SIVAE_OOB;
  const unsigned wrow = (unsigned)a.K * 4u;
  f32x16 acc[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
  // software-pipelined like the forward: the next 8-wide n group is requested before the MFMAs of the current one
  // (N % 4 == 0: the lane's four n values are all inside or all past the end)
  float4 dA[RB], dB[RB];
  float wA[4], wB[4];
#define LIN_LOAD(NN, D, W)                                                     \
  {                                                                            \
    const bool vn_ = (NN) + 4 * h < n_end;                                     \
    _Pragma("unroll") for (int r = 0; r < RB; ++r) D[r] = buf_load_f32x4(dr, vn_ ? d_o[r] : SIVAE_OOB, (unsigned)(NN) * 4u); \
    const unsigned wso_ = (unsigned)(NN) * wrow;                               \
    const unsigned wov_ = vn_ ? wo : SIVAE_OOB;                                \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) W[q] = buf_load_f32(wr, wov_, wso_ + (unsigned)q * wrow); \
  }
#define LIN_MMA(D, W)                                                          \
  _Pragma("unroll") for (int r = 0; r < RB; ++r) {                             \
    acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(D[r].x, W[0], acc[r], 0, 0, 0); \
    acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(D[r].y, W[1], acc[r], 0, 0, 0); \
    acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(D[r].z, W[2], acc[r], 0, 0, 0); \
    acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(D[r].w, W[3], acc[r], 0, 0, 0); \
  }
  LIN_LOAD(n_begin, dA, wA)
  for (int n = n_begin; n < n_end; n += 16) {
    LIN_LOAD(n + 8, dB, wB)
    __builtin_amdgcn_sched_barrier(0);
    LIN_MMA(dA, wA)
    __builtin_amdgcn_sched_barrier(0);
    LIN_LOAD(n + 16, dA, wA)
    __builtin_amdgcn_sched_barrier(0);
    LIN_MMA(dB, wB)
    __builtin_amdgcn_sched_barrier(0);
  }
#undef LIN_LOAD
#undef LIN_MMA
  const int k = k0 + i;
  if (k >= a.K) return;
  float* dst = a.out + (size_t)s * a.B * a.K;
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int b = r * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (b < a.B) dst[(size_t)b * a.K + k] = acc[r][e];
    }
}

// ---- wgrad: dW[n][k] = sum_b dy[b][n] x[b][k]; a wave owns 32 n x 128 k (the dy operand is reused by 4 MFMAs)
__global__ void __launch_bounds__(64) linear_wgrad_kernel(LinArgs a) {
  const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
  const int k_strips = (a.K + 127) / 128;
  const int kt = blockIdx.x % k_strips, nt = blockIdx.x / k_strips;
  const int n0 = nt * 32, k0 = kt * 128;
  const __amdgpu_buffer_rsrc_t dr = make_rsrc(a.a, (unsigned long long)a.B * a.N * 4ull);
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(a.w, (unsigned long long)a.B * a.K * 4ull);
  const unsigned d_o = (n0 + i < a.N) ? (unsigned)(h * a.N + n0 + i) * 4u : SIVAE_OOB;
  unsigned xo[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) xo[t] = (k0 + t * 32 + i < a.K) ? (unsigned)(h * a.K + k0 + t * 32 + i) * 4u : SIVAE_OOB;
  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  const unsigned drow = (unsigned)a.N * 4u, xrow = (unsigned)a.K * 4u;
  // UB row pairs per step: all their loads are issued before the first MFMA (round 6: one pair per step made every step
  // wait out a memory round trip — 117 us for the 512-row batches of config 2, where the launch is one wave per CU);
  // the order of the sum over b is unchanged
  constexpr int UB = 8;
  for (int b = 0; b < a.B; b += 2 * UB) {
    float dv[UB], xv[UB][4];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      // rows b + 2u + h; rows past the batch read through an out-of-range offset -> 0
      const bool ok = b + 2 * u + h < a.B;
      dv[u] = buf_load_f32(dr, ok ? d_o : SIVAE_OOB, (unsigned)(b + 2 * u) * drow);
#pragma unroll
      for (int t = 0; t < 4; ++t) xv[u][t] = buf_load_f32(xr, ok ? xo[t] : SIVAE_OOB, (unsigned)(b + 2 * u) * xrow);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(dv[u], xv[u][t], acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int k = k0 + t * 32 + i;
    if (k < a.K) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (n < a.N) a.out[(size_t)n * a.K + k] = acc[t][e];
      }
    }
  }
}

// y[b][c] = sum_s part[s][b][c] (+ bias[c]) (ReLU)
__global__ void linear_reduce_kernel(const float* __restrict__ part, float* __restrict__ y,
                                     const float* __restrict__ bias, int S, size_t numel, int cols, int relu) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= numel) return;
  float s0 = 0.f, s1 = 0.f;
  int s = 0;
  for (; s + 1 < S; s += 2) {
    s0 += part[(size_t)s * numel + e];
    s1 += part[(size_t)(s + 1) * numel + e];
  }
  if (s < S) s0 += part[(size_t)s * numel + e];
  float v = s0 + s1;
  if (bias) v += bias[e % cols];
  if (relu) v = fmaxf(v, 0.f);
  y[e] = v;
}

// contraction split: ~1024 waves per call, slices of >= 64 (multiple of 16)
int split_for(int tiles, int contraction, int* slice_len) {
  int S = (1024 + tiles - 1) / tiles;
  const int max_s = contraction / 64 > 0 ? contraction / 64 : 1;
  if (S > max_s) S = max_s;
  if (S > 64) S = 64;
  if (S < 1) S = 1;
  int len = (contraction + S - 1) / S;
  len = ((len + 15) / 16) * 16;
  S = (contraction + len - 1) / len;
  *slice_len = len;
  return S;
}

template <typename F>
int dispatch_rb(int B, F&& f) {
  const int rb = (B + 31) / 32;
  switch (rb) {
    case 1: return f(std::integral_constant<int, 1>{});
    case 2: return f(std::integral_constant<int, 2>{});
    case 3:
    case 4: return f(std::integral_constant<int, 4>{});
    default: return f(std::integral_constant<int, 8>{});
  }
}

bool lin_ok(int B, int K, int N) {
  // (batches above 32 * MAX_RB = 256 rows run as row chunks of 256: forward / dgrad rows are independent, wgrad walks
  // the whole batch anyway — the paired 2 x 256-image passes of the 32x32 configuration used to fall back to the 1x1-conv
  // path: 187 us per call instead of ~20)
  return B > 0 && K > 0 && N > 0 && B <= 16384 && (K % 4) == 0 && (N % 4) == 0 &&
         (long long)B * K * 4 < 0x7fffffffLL && (long long)N * K * 4 < 0xffffffffLL && (long long)B * N * 4 < 0x7fffffffLL;
}

}  // namespace

extern "C" int sivae_linear_supported(int B, int K, int N) { return lin_ok(B, K, N) ? 1 : 0; }

extern "C" size_t sivae_linear_workspace_bytes(int B, int K, int N) {
  if (!lin_ok(B, K, N)) return 0;
  int l1, l2;
  const int s_f = split_for((N + 31) / 32, K, &l1);
  const int s_d = split_for((K + 31) / 32, N, &l2);
  const size_t f = s_f > 1 ? (size_t)s_f * B * N : 0, d = s_d > 1 ? (size_t)s_d * B * K : 0;
  return (f > d ? f : d) * sizeof(float) + 16;
}

extern "C" int sivae_linear_fwd(const float* x, const float* w, const float* bias, float* y, int relu, int B, int K,
                                int N, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!x || !w || !y) return SIVAE_ERR_NULL;
  if (!lin_ok(B, K, N)) return SIVAE_ERR_SHAPE;
  if (B > 32 * MAX_RB) {
    for (int r0 = 0; r0 < B; r0 += 32 * MAX_RB) {
      const int nb = B - r0 < 32 * MAX_RB ? B - r0 : 32 * MAX_RB;
      const int rc = sivae_linear_fwd(x + (size_t)r0 * K, w, bias, y + (size_t)r0 * N, relu, nb, K, N, workspace,
                                      workspace_bytes, stream);
      if (rc != SIVAE_OK) return rc;
    }
    return SIVAE_OK;
  }
  LinArgs a;
  a.a = x;
  a.w = w;
  a.bias = bias;
  a.B = B;
  a.K = K;
  a.N = N;
  a.relu = relu;
  const int tiles = (N + 31) / 32;
  a.S = split_for(tiles, K, &a.slice_len);
  if (a.S > 1) {
    if (!workspace || workspace_bytes < (size_t)a.S * B * N * sizeof(float)) return SIVAE_ERR_WORKSPACE;
    a.out = reinterpret_cast<float*>(workspace);
  } else {
    a.out = y;
  }
  const int rc = dispatch_rb(B, [&](auto rb) {
    hipLaunchKernelGGL((linear_fwd_kernel<decltype(rb)::value>), dim3(tiles * a.S), dim3(64), 0, stream, a);
    return sivae_launch_status();
  });
  if (rc != SIVAE_OK || a.S == 1) return rc;
  const size_t numel = (size_t)B * N;
  hipLaunchKernelGGL(linear_reduce_kernel, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, stream, a.out, y, bias,
                     a.S, numel, N, relu);
  return sivae_launch_status();
}

extern "C" int sivae_linear_dgrad(const float* dy, const float* w, float* dx, int B, int K, int N, void* workspace,
                                  size_t workspace_bytes, hipStream_t stream) {
  if (!dy || !w || !dx) return SIVAE_ERR_NULL;
  if (!lin_ok(B, K, N)) return SIVAE_ERR_SHAPE;
  if (B > 32 * MAX_RB) {
    for (int r0 = 0; r0 < B; r0 += 32 * MAX_RB) {
      const int nb = B - r0 < 32 * MAX_RB ? B - r0 : 32 * MAX_RB;
      const int rc = sivae_linear_dgrad(dy + (size_t)r0 * N, w, dx + (size_t)r0 * K, nb, K, N, workspace, workspace_bytes,
                                        stream);
      if (rc != SIVAE_OK) return rc;
    }
    return SIVAE_OK;
  }
  LinArgs a;
  a.a = dy;
  a.w = w;
  a.bias = nullptr;
  a.B = B;
  a.K = K;
  a.N = N;
  a.relu = 0;
  const int tiles = (K + 31) / 32;
  a.S = split_for(tiles, N, &a.slice_len);
  a.slice_len = ((a.slice_len + 7) / 8) * 8;
  if (a.S > 1) {
    if (!workspace || workspace_bytes < (size_t)a.S * B * K * sizeof(float)) return SIVAE_ERR_WORKSPACE;
    a.out = reinterpret_cast<float*>(workspace);
  } else {
    a.out = dx;
  }
  const int rc = dispatch_rb(B, [&](auto rb) {
    hipLaunchKernelGGL((linear_dgrad_kernel<decltype(rb)::value>), dim3(tiles * a.S), dim3(64), 0, stream, a);
    return sivae_launch_status();
  });
  if (rc != SIVAE_OK || a.S == 1) return rc;
  const size_t numel = (size_t)B * K;
  hipLaunchKernelGGL(linear_reduce_kernel, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, stream, a.out, dx,
                     (const float*)nullptr, a.S, numel, K, 0);
  return sivae_launch_status();
}

extern "C" int sivae_linear_wgrad(const float* dy, const float* x, float* dw, int B, int K, int N,
                                  hipStream_t stream) {
  if (!dy || !x || !dw) return SIVAE_ERR_NULL;
  if (!lin_ok(B, K, N)) return SIVAE_ERR_SHAPE;
  LinArgs a;
  a.a = dy;
  a.w = x;
  a.out = dw;
  a.bias = nullptr;
  a.B = B;
  a.K = K;
  a.N = N;
  a.S = 1;
  a.slice_len = 0;
  a.relu = 0;
  const long long nblk = (long long)((N + 31) / 32) * ((K + 127) / 128);
  hipLaunchKernelGGL(linear_wgrad_kernel, dim3((unsigned)nblk), dim3(64), 0, stream, a);
  return sivae_launch_status();
}
