// Weight re-layout for the implicit-GEMM kernels.
//   mode 0 (forward):  wp[tap][ci][co]  = w[co][ci][tap]                 dims [T][ci_pad(Ci)][co_pad(Co)]
//   mode 1 (dgrad):    wp[tap][co][ci]  = w[co][ci][T-1-tap]  (180° flip) dims [T][ci_pad(Co)][co_pad(Ci)]
// Padding entries are zero, so the GEMM kernels need no bounds checks on the weight operand.
// Runs once per optimizer step per layer (weights are reused by 5-8 passes per iteration).
#include "common.h"
#include "pack_batch.h"

extern "C" int sivae_conv_ci_pad(int ks, int ci);
extern "C" int sivae_conv_co_pad(int co);

__device__ __forceinline__ void pack_weight_body(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci,
                                                 int taps, int mode, int kpad, int npad, size_t total, size_t i,
                                                 const size_t stride) {
  for (; i < total; i += stride) {
    const int n = (int)(i % npad);
    const size_t t = i / npad;
    const int k = (int)(t % kpad);
    const int tap = (int)(t / kpad);
    float v = 0.f;
    if (mode == 0) {
      // k = ci, n = co
      if (k < Ci && n < Co) v = w[((size_t)n * Ci + k) * taps + tap];
    } else {
      // k = co, n = ci, flipped tap
      if (k < Co && n < Ci) v = w[((size_t)k * Ci + n) * taps + (taps - 1 - tap)];
    }
    wp[i] = v;
  }
}

__global__ void __launch_bounds__(256) pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                                          int Co, int Ci, int taps, int mode, int kdim,
                                                          int kpad, int npad, size_t total) {
  (void)kdim;
  pack_weight_body(w, wp, Co, Ci, taps, mode, kpad, npad, total, (size_t)blockIdx.x * 256 + threadIdx.x,
                   (size_t)gridDim.x * 256);
}

__global__ void __launch_bounds__(256) pack_weight_batch_kernel(const SivaePackJob* __restrict__ jobs,
                                                                const unsigned short* __restrict__ block_job) {
  const SivaePackJob j = jobs[block_job[blockIdx.x]];
  pack_weight_body(j.w, j.dst, j.Co, j.Ci, j.taps, j.mode, j.kpad, j.npad, (size_t)j.total,
                   (size_t)(blockIdx.x - j.blk0) * 256 + threadIdx.x, (size_t)j.nblk * 256);
}

extern "C" size_t sivae_pack_conv_weight_bytes(int Co, int Ci, int ks, int mode) {
  if (ks != 1 && ks != 3 && ks != 5) return 0;
  const int kdim = mode == 0 ? Ci : Co, ndim = mode == 0 ? Co : Ci;
  return (size_t)ks * ks * sivae_conv_ci_pad(ks, kdim) * sivae_conv_co_pad(ndim) * sizeof(float);
}

extern "C" int sivae_pack_conv_weight(const float* w, float* wp, int Co, int Ci, int ks, int mode,
                                      hipStream_t stream) {
  if (!w || !wp) return SIVAE_ERR_NULL;
  if (Co <= 0 || Ci <= 0) return SIVAE_ERR_SHAPE;
  if (ks != 1 && ks != 3 && ks != 5) return SIVAE_ERR_KSIZE;
  if (mode != 0 && mode != 1) return SIVAE_ERR_MODE;
  const int kdim = mode == 0 ? Ci : Co, ndim = mode == 0 ? Co : Ci;
  const int kpad = sivae_conv_ci_pad(ks, kdim), npad = sivae_conv_co_pad(ndim);
  const size_t total = (size_t)ks * ks * kpad * npad;
  int nb = cdiv((long long)total, 256);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(nb), dim3(256), 0, stream, w, wp, Co, Ci, ks * ks, mode, kdim,
                     kpad, npad, total);
  return sivae_launch_status();
}

// ---- batched packing: job shapes of the direct form + the dispatchers of every form (pack_batch.h)
int sivae_packjob_direct(SivaePackJob* j, int Co, int Ci, int ks, int mode) {
  if (ks != 1 && ks != 3 && ks != 5) return SIVAE_ERR_KSIZE;
  const int kdim = mode == 0 ? Ci : Co, ndim = mode == 0 ? Co : Ci;
  j->taps = ks * ks;
  j->kdim = kdim;
  j->ndim = ndim;
  j->kpad = sivae_conv_ci_pad(ks, kdim);
  j->npad = sivae_conv_co_pad(ndim);
  j->total = (unsigned long long)ks * ks * j->kpad * j->npad;
  return SIVAE_OK;
}
void sivae_packbatch_direct(const SivaePackJob* jobs, const unsigned short* block_job, int nblocks, hipStream_t s) {
  hipLaunchKernelGGL(pack_weight_batch_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, jobs, block_job);
}

extern "C" int sivae_pack_job_bytes() { return (int)sizeof(SivaePackJob); }

// Fill job `index` of a HOST job table (sivae_pack_job_bytes() bytes per job) for operand form `form` (0 direct [ks, mode],
// 1 Winograd F(2x2,3x3) [mode], 2 Winograd F(4x4,3x3) [mode], 3 upsample-phase forward, 4 upsample-phase data gradient,
// 5 Winograd F(4x4,3x3) pre-split into three bf16 pieces [mode] (conv_wino4_b6.hip), 6 bf16 operand slabs of the bf16 mode
// [ks incl. the code 51, mode] (bf16_conv.hip; dst is the bf16 buffer of sivae_bf16_pack_conv_weight))
// of the weight w [Co][Ci][ks][ks] -> dst (the buffer the per-weight sivae_pack_* call of that form writes);
// first_block: the job's first block in the batch launch.  Returns the number of blocks the job takes, or < 0.
extern "C" int sivae_pack_job_fill(void* jobs_host, int index, int form, const float* w, float* dst, int Co, int Ci,
                                   int ks, int mode, int first_block) {
  if (!jobs_host || !w || !dst) return SIVAE_ERR_NULL;
  if (index < 0 || Co <= 0 || Ci <= 0 || first_block < 0) return SIVAE_ERR_SHAPE;
  if (mode != 0 && mode != 1) return SIVAE_ERR_MODE;
  SivaePackJob* j = reinterpret_cast<SivaePackJob*>(jobs_host) + index;
  j->w = w;
  j->dst = dst;
  j->Co = Co;
  j->Ci = Ci;
  j->mode = mode;
  j->taps = ks * ks;
  int rc;
  switch (form) {
    case SIVAE_PACK_DIRECT: rc = sivae_packjob_direct(j, Co, Ci, ks, mode); break;
    case SIVAE_PACK_WINO: rc = ks == 3 ? sivae_packjob_wino(j, Co, Ci, mode) : SIVAE_ERR_KSIZE; break;
    case SIVAE_PACK_WINO4: rc = ks == 3 ? sivae_packjob_wino4(j, Co, Ci, mode) : SIVAE_ERR_KSIZE; break;
    case SIVAE_PACK_WINO4_B6: rc = ks == 3 ? sivae_packjob_wino4_b6(j, Co, Ci, mode) : SIVAE_ERR_KSIZE; break;
    case SIVAE_PACK_WINO_UP: rc = ks == 3 ? sivae_packjob_wino_up(j, Co, Ci) : SIVAE_ERR_KSIZE; break;
    case SIVAE_PACK_WINO_UP_DGRAD: rc = ks == 3 ? sivae_packjob_wino_up_dgrad(j, Co, Ci) : SIVAE_ERR_KSIZE; break;
    case SIVAE_PACK_BF16: rc = sivae_packjob_bf16(j, Co, Ci, ks, mode); break;
    default: rc = SIVAE_ERR_MODE;
  }
  if (rc != SIVAE_OK) return rc;
  j->blk0 = (unsigned)first_block;
  j->nblk = sivae_pack_job_blocks(j->total);
  return (int)j->nblk;
}

// One launch rebuilds every job of a DEVICE job table of one operand form (block_job: job index per block, uint16)
extern "C" int sivae_pack_batch(int form, const void* jobs_dev, const unsigned short* block_job_dev, int n_blocks,
                                hipStream_t stream) {
  if (!jobs_dev || !block_job_dev) return SIVAE_ERR_NULL;
  if (n_blocks <= 0) return SIVAE_ERR_SHAPE;
  const SivaePackJob* jobs = reinterpret_cast<const SivaePackJob*>(jobs_dev);
  switch (form) {
    case SIVAE_PACK_DIRECT: sivae_packbatch_direct(jobs, block_job_dev, n_blocks, stream); break;
    case SIVAE_PACK_WINO: sivae_packbatch_wino(jobs, block_job_dev, n_blocks, stream); break;
    case SIVAE_PACK_WINO4: sivae_packbatch_wino4(jobs, block_job_dev, n_blocks, stream); break;
    case SIVAE_PACK_WINO4_B6: sivae_packbatch_wino4_b6(jobs, block_job_dev, n_blocks, stream); break;
    case SIVAE_PACK_WINO_UP: sivae_packbatch_wino_up(jobs, block_job_dev, n_blocks, stream); break;
    case SIVAE_PACK_WINO_UP_DGRAD: sivae_packbatch_wino_up_dgrad(jobs, block_job_dev, n_blocks, stream); break;
    case SIVAE_PACK_BF16: sivae_packbatch_bf16(jobs, block_job_dev, n_blocks, stream); break;
    default: return SIVAE_ERR_MODE;
  }
  return sivae_launch_status();
}
