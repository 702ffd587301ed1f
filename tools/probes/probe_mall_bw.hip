// Read / write bandwidth against working-set size on one MI355X: where do L2 (8 x 4 MB), the 256 MB Infinity Cache and
// HBM show?   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/probe_mall_bw tools/probes/probe_mall_bw.hip && gpurun_out/probe_mall_bw
// (input to the BatchNorm-backward design of the next round: can phase 2 re-read its slab from the Infinity Cache?)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void __launch_bounds__(256) read_kernel(const float4* __restrict__ p, size_t n4, int reps, float* sink) {
  float acc = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  for (int r = 0; r < reps; ++r)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      const float4 v = p[i];
      acc += v.x + v.y + v.z + v.w;
    }
  if (acc == 123.456f) sink[0] = acc;
}
__global__ void __launch_bounds__(256) copy_kernel(const float4* __restrict__ p, float4* __restrict__ q, size_t n4, int reps) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (int r = 0; r < reps; ++r)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) q[i] = p[i];
}
// two-phase pattern of a BatchNorm backward group: read A and B (phase 1), then read A and B again and write C (phase 2)
__global__ void __launch_bounds__(256) two_phase_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                        float4* __restrict__ c, size_t n4, float* sink) {
  float acc = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const float4 u = a[i], v = b[i];
    acc += u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
  }
  if (acc == 123.456f) sink[0] = acc;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const float4 u = a[i], v = b[i];
    c[i] = make_float4(u.x - v.x, u.y - v.y, u.z - v.z, u.w - v.w);
  }
}

int main() {
  const size_t maxb = (size_t)2 << 30;
  float4 *a, *b, *c;
  float* sink;
  hipMalloc(&a, maxb); hipMalloc(&b, maxb); hipMalloc(&c, maxb); hipMalloc(&sink, 4);
  hipMemset(a, 0, maxb); hipMemset(b, 0, maxb); hipMemset(c, 0, maxb);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t sizes[] = {4u << 20, 16u << 20, 32u << 20, 64u << 20, 96u << 20, 128u << 20, 192u << 20, 256u << 20, 512u << 20, (size_t)1 << 30, (size_t)2 << 30};
  printf("%10s %14s %14s %22s\n", "MB", "read GB/s", "copy GB/s(r+w)", "two-phase GB/s (2r+2r+w)");
  for (size_t sz : sizes) {
    const size_t n4 = sz / 16;
    int reps = (int)(((size_t)8 << 30) / sz);
    if (reps < 2) reps = 2;
    float ms;
    hipLaunchKernelGGL(read_kernel, dim3(2048), dim3(256), 0, 0, a, n4, 2, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL(read_kernel, dim3(2048), dim3(256), 0, 0, a, n4, reps, sink);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    const double rd = (double)sz * reps / ms / 1e6;
    hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, 0, a, b, n4 / 2, 2);
    hipEventRecord(e0);
    hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, 0, a, b, n4 / 2, reps);  // working set = sz (half read, half written)
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    const double cp = (double)sz * reps / ms / 1e6;
    // two-phase: working set = a + b of sz/2 each (re-read), c of sz/2 written
    const size_t h4 = n4 / 2;
    hipLaunchKernelGGL(two_phase_kernel, dim3(2048), dim3(256), 0, 0, a, b, c, h4, sink);
    hipEventRecord(e0);
    const int tr = reps < 4 ? 4 : reps;
    for (int r = 0; r < tr; ++r) hipLaunchKernelGGL(two_phase_kernel, dim3(2048), dim3(256), 0, 0, a, b, c, h4, sink);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    const double tp = (double)(sz / 2) * 5.0 * tr / ms / 1e6;  // 2 reads + 2 re-reads + 1 write of sz/2 each
    printf("%10.0f %14.0f %14.0f %22.0f\n", sz / 1048576.0, rd, cp, tp);
  }
  return 0;
}
