"""Driver for the rocprofv3 --pmc passes (HBM traffic): two calibration dispatches of known size, then ONE bench step.

  dispatch 1: fill_  of 1 GiB   (writes 2^30 B, reads 0)          -> calibrates WRITE_SIZE
  dispatch 2: copy_  of 1 GiB   (reads 2^30 B, writes 2^30 B)     -> calibrates FETCH_SIZE
  then `bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing` (bs128, 256x256)

usage (one counter family per pass; FETCH_SIZE and WRITE_SIZE do not fit in one):
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc/fetch -- python tools/pmc_step.py
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc/write -- python tools/pmc_step.py
  python tools/pmc_traffic.py gpurun_out/pmc/fetch gpurun_out/pmc/write > profiles/r1_pmc_traffic.json
"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

a = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
a.fill_(1.0)
b = torch.empty_like(a)
b.copy_(a)
torch.cuda.synchronize()
del a, b
torch.cuda.empty_cache()

import bench  # noqa: E402

sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-kernel-timing"] + sys.argv[1:]
bench.main()
