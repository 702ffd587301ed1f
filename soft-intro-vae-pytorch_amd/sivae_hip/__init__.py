"""sivae_hip — Python host layer over libsivae_hip.so (hand-written HIP kernels for gfx950).

`lib`         ctypes loader, prototypes parsed from include/sivae_hip.h
`ops`         tensor-level kernel wrappers (no autograd)
`functional`  torch.autograd.Function wrappers composing the kernels into differentiable blocks
`dp`          one-process-per-GPU data parallelism (flat gradient buffers, RCCL all-reduce over xGMI)
"""
from . import lib  # noqa: F401
