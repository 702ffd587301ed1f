"""Tensor-level wrappers over the libsivae_hip C ABI (no autograd here — see `functional.py`).

Every function takes contiguous fp32 ROCm-device tensors, allocates the outputs, launches on torch's
current HIP stream and returns immediately (stream-async). CPU tensors are rejected loudly: the
product path has no CPU fallback.
"""
import ctypes
import os

import torch

from . import lib as _lib

LRELU_SLOPE = 0.2
LOSS_TYPES = {"mse": 0, "l1": 1, "bce": 2}

_workspaces = {}


class KernelTimer:
    """Optional per-launch HIP-event timing of the MFMA conv kernels (used by bench.py for the roofline
    object).  Events are recorded on torch's current stream — the stream the kernels are launched on —
    and only read back after the timed region, so the instrumented run stays asynchronous."""

    def __init__(self):
        self.records = []  # (kernel_key, flops, start_event, end_event)

    def begin(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def end(self, key, flops, start, executed=None):
        """flops: algorithmic FLOPs of the reference op; executed: FLOPs the kernel actually issues to the matrix
        pipe when that differs (Winograd)."""
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.records.append((key, flops, start, ev, flops if executed is None else executed))

    def summary(self):
        """-> {kernel_key: dict(launches, total_ms, avg_ms, flops)} (call after torch.cuda.synchronize())"""
        out = {}
        for key, flops, s, e, ex in self.records:
            d = out.setdefault(key, dict(launches=0, total_ms=0.0, flops=0.0, executed_flops=0.0))
            d["launches"] += 1
            d["total_ms"] += s.elapsed_time(e)
            d["flops"] += flops
            d["executed_flops"] += ex
        for d in out.values():
            d["avg_ms"] = d["total_ms"] / max(d["launches"], 1)
        return out


TIMER = None  # set to a KernelTimer() to instrument conv2d_fwd / conv2d_wgrad launches


def conv_fwd_kernel_key(ks, Co, pro):
    """name of the template instantiation sivae_conv2d_fwd dispatches to (mirrors conv_fwd.hip)"""
    tile = {3: ("1,2,1,4,8,3", "2,2,1,4,2,3", "2,2,2,2,2,2"), 1: ("1,2,1,4,32,1", "2,2,1,4,16,1", "2,2,2,2,16,1"),
            5: ("1,2,1,4,4,4", "2,2,1,4,4,4", "2,2,2,2,4,3")}[ks][0 if Co <= 32 else (1 if Co <= 64 else 2)]
    return "conv_fwd_kernel<%d,%s,%s>" % (ks, tile, "true" if pro else "false")


def conv_wgrad_kernel_key(ks, Co, pro):
    tile = {3: "3,1,1,2,2,3", 1: "1,2,2,2,2,1",
            5: ("1,1,1,1,2,2" if Co <= 32 else "1,1,1,2,1,2")}[ks]
    return "conv_wgrad_kernel<%d,%s,%s>" % (ks, tile, "true" if pro else "false")


def u8_to_f32(src, flip=None, nhwc=False, scale=1.0 / 255.0):
    """uint8 image batch (NCHW, or NHWC with nhwc=True) -> fp32 NCHW * scale; flip: int32 [B], nonzero = mirror."""
    if not src.is_cuda or src.dtype != torch.uint8 or not src.is_contiguous() or src.dim() != 4:
        raise TypeError("sivae_hip.u8_to_f32: expected a contiguous 4-D uint8 ROCm tensor")
    if nhwc:
        B, H, W, C = src.shape
    else:
        B, C, H, W = src.shape
    if flip is not None and (not flip.is_cuda or flip.dtype != torch.int32 or flip.numel() != B):
        raise TypeError("sivae_hip.u8_to_f32: flip must be an int32 ROCm tensor with one entry per sample")
    dst = torch.empty((B, C, H, W), dtype=torch.float32, device=src.device)
    _lib.call("sivae_u8_to_f32", _p(src), _p(dst), _p(flip), B, C, H, W, int(bool(nhwc)), float(scale), _s())
    return dst


def f32_to_u8(src, scale=255.0):
    """fp32 images -> uint8 the way the reference quantises generated batches (clip(x*255, 0, 255) truncated)"""
    if not src.is_cuda or src.dtype != torch.float32 or not src.is_contiguous():
        raise TypeError("sivae_hip.f32_to_u8: expected a contiguous float32 ROCm tensor")
    dst = torch.empty(src.shape, dtype=torch.uint8, device=src.device)
    _lib.call("sivae_f32_to_u8", _p(src), _p(dst), src.numel(), float(scale), _s())
    return dst


def _require(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("sivae_hip: tensor is on %s — the Soft-IntroVAE HIP kernels need a ROCm device "
                               "tensor and have no CPU fallback" % t.device)
        if t.dtype != torch.float32 and t.dtype != torch.int64:
            raise TypeError("sivae_hip: expected float32, got %s" % t.dtype)
        if not t.is_contiguous():
            raise ValueError("sivae_hip: tensor must be contiguous")


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _s(t=None):
    """HIP stream the launch goes to: torch's current stream of the TENSOR's device (the drop-in entry points also
    make that device current, so legacy `_s()` calls agree with it)"""
    return ctypes.c_void_p(torch.cuda.current_stream(None if t is None else t.device).cuda_stream)


def workspace(nbytes, device):
    """Grow-only scratch buffer per (device, stream); kernels on one stream are ordered, so sharing is safe."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


_counters = {}
# SIVAE_BN_FUSED_FINALIZE=1: the per-channel finalize of the BatchNorm backward runs in the last block of the reduction
# kernel (agent-scope fence + counter) instead of as its own ~5 us launch.  Measured a LOSS (round 3, one call, 256x256):
# 16-image shard 213.9 vs 227.3 img/s, bootstrap 8-image shard 196.4 vs 215.0 — the release fence every block needs
# (buffer_wbl2 + buffer_inv at agent scope: the 8 XCD L2s are not coherent inside a kernel) costs far more than the
# ~110 launches it removes -> off by default.
BN_FUSED_FINALIZE = os.environ.get("SIVAE_BN_FUSED_FINALIZE", "0") == "1"


def counters(device):
    """8192 zero-initialised unsigned ints per (device, stream) for kernels that finish a reduction in their last block
    (each such kernel leaves them zero)"""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _counters.get(key)
    if buf is None:
        buf = torch.zeros(8192, dtype=torch.int32, device=device)
        _counters[key] = buf
    return buf


# SIVAE_BN_FUSED=0: the three-launch BatchNorm backward (reduce / finalize / dx: two reads of dy and x) instead of the
# one-pass persistent kernel of bn_fused.hip (one read, activations held in registers across a grid barrier).  The
# persistent kernel needs its whole grid resident — two PROCESSES sharing one GPU (the 2-rank same-device tests) must
# switch it off; one process per GPU (the deployment) is fine.
BN_FUSED = os.environ.get("SIVAE_BN_FUSED", "1") != "0"
_bn_states = {}


def bn_fused_state(device):
    """barrier / counter state of sivae_bn_bwd_fused per (device, stream): zeroed once, left consistent by every call"""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _bn_states.get(key)
    if buf is None:
        buf = torch.zeros(_lib.load().sivae_bn_bwd_fused_state_uints(), dtype=torch.int32, device=device)
        _bn_states[key] = buf
    return buf


BN_FUSED_POISON_MSG = ("sivae_hip: a one-pass BatchNorm backward (sivae_bn_bwd_fused) gave up at its grid barrier on "
                       "device/stream %s: its grid was not fully resident (another persistent kernel or process on "
                       "the GPU, or a CU mask).  The iteration's gradients are invalid.  Set SIVAE_BN_FUSED=0 (the "
                       "three-launch form) or SIVAE_BN_FUSED_PERSISTENT=0 for this deployment.")


def bn_fused_poisoned():
    """The non-raising form of `bn_fused_check`: None, or the message to raise with.  Data-parallel callers fold the
    answer into a collective flag first (sivae_hip.dp.any_rank) so that every rank raises together — a rank raising alone
    leaves the others waiting in the next all-reduce."""
    if not _bn_states:
        return None
    w = _lib.load().sivae_bn_bwd_fused_poison_word()
    bad = [k for k, buf in _bn_states.items() if int(buf[w].item()) != 0]
    if not bad:
        return None
    for k in bad:
        _bn_states[k].zero_()
    return BN_FUSED_POISON_MSG % (bad,)


def bn_fused_check():
    """Was a persistent BatchNorm-backward launch abandoned (its grid barrier timed out because the grid was not fully
    resident — another persistent kernel, a CU mask the runtime does not report)?  Reads one word per barrier state (a
    device->host copy: call it where results are read back anyway — the training loops do, once per logging interval);
    if set: zeroes the state (the counters of an abandoned launch are inconsistent) and raises.  The outputs of the
    abandoned launch and of everything after it are garbage."""
    msg = bn_fused_poisoned()
    if msg is not None:
        raise RuntimeError(msg)


def _bn_bwd_fused_ok(x, nseg):
    if not BN_FUSED or SYNC_BN is not None or x.dim() != 4:
        return False
    if os.environ.get("SIVAE_DP_SAME_DEVICE", "0") == "1":
        return False  # several ranks share this GPU (tests only): their persistent grids could starve each other
    B, C, H, W = x.shape
    return _lib.load().sivae_bn_bwd_fused_supported(B, C, H, W, B // nseg) == 1


def _bn_bwd_fused(dy, y, mask, x, mean, invstd, gamma, beta, act_mode, slope, dx, dz, dgamma, dbeta, dy_pooled, dz_sum,
                  nseg):
    B, C, H, W = x.shape
    L = _lib.load()
    ws = workspace(L.sivae_bn_bwd_fused_workspace_bytes(B, C, H, W, B // nseg), x.device)
    try:
        _lib.call("sivae_bn_bwd_fused", _p(dy), _p(y), _p(mask), _p(x), _p(mean), _p(invstd), _p(gamma), _p(beta),
                  int(act_mode), float(slope), _p(dx), _p(dz), _p(dgamma), _p(dbeta), B, C, H, W, int(bool(dy_pooled)),
                  int(bool(dz_sum)), B // nseg, _p(bn_fused_state(x.device)), _p(ws), ws.numel(), _s(x))
    except _lib.SivaeError as e:
        if e.code != -2:
            raise
        # a shape the plan of this variant does not take after all (the query and the launch plan with the same
        # register budget since round 5, so this is a safety net): the three-launch form takes every shape
        _bn_bwd_seg(dy, y, mask, x, mean, invstd, gamma, beta, act_mode, slope, dx, dz, dgamma, dbeta, dy_pooled, dz_sum,
                    nseg)


def _out(out, shape, device):
    """a caller-provided destination (a contiguous fp32 device tensor of exactly `shape`: the parameter-gradient slab
    views of optim.FlatAdam) or a fresh tensor"""
    if out is None:
        return torch.empty(shape, dtype=torch.float32, device=device)
    if (tuple(out.shape) != tuple(shape) or out.dtype != torch.float32 or not out.is_contiguous()
            or out.device != device):
        raise ValueError("sivae_hip: out must be a contiguous float32 tensor of shape %s on %s" % (tuple(shape), device))
    return out


def _pg(pg_out, C, device, want):
    """(dgamma, dbeta) destinations of a BatchNorm backward"""
    if not want:
        return None, None
    if pg_out is not None:
        return _out(pg_out[0], (C,), device), _out(pg_out[1], (C,), device)
    return (torch.empty(C, dtype=torch.float32, device=device), torch.empty(C, dtype=torch.float32, device=device))


# ------------------------------------------------------------------------------------------------ conv
def pack_weight(w, mode):
    """w [Co, Ci, k, k] (or [out, in] for Linear) -> packed GEMM operand. mode 0 fwd, 1 dgrad."""
    _require(w)
    if w.dim() == 2:
        Co, Ci, ks = w.shape[0], w.shape[1], 1
    else:
        Co, Ci, ks = w.shape[0], w.shape[1], w.shape[2]
    L = _lib.load()
    nbytes = L.sivae_pack_conv_weight_bytes(Co, Ci, ks, mode)
    if nbytes == 0:
        raise _lib.SivaeError("sivae_pack_conv_weight_bytes", -3)
    wp = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
    _lib.call("sivae_pack_conv_weight", _p(w), _p(wp), Co, Ci, ks, mode, _s())
    return wp


def pack_wino(w, mode):
    """w [Co, Ci, 3, 3] -> Winograd F(2x2,3x3) filter transform U = G g G^T, packed [j][ci][co][i]."""
    _require(w)
    Co, Ci = w.shape[0], w.shape[1]
    assert w.dim() == 4 and w.shape[2] == 3 and w.shape[3] == 3
    nbytes = _lib.load().sivae_pack_wino_weight_bytes(Co, Ci, mode)
    if nbytes == 0:
        raise _lib.SivaeError("sivae_pack_wino_weight_bytes", -3)
    up = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
    _lib.call("sivae_pack_wino_weight", _p(w), _p(up), Co, Ci, mode, _s())
    return up


# SIVAE_WINO=0 keeps every 3x3 conv on the direct implicit-GEMM kernel (A/B measurements, debugging)
WINO = os.environ.get("SIVAE_WINO", "1") != "0"
WINO_UP = os.environ.get("SIVAE_WINO_UP", "1") != "0"  # phase-decomposed F(2x2,2x2) kernel for conv-after-upsample
# Winograd F(4x4,3x3) for the large-map 3x3 convs without a fused prologue (SIVAE_WINO4=0: F(2x2,3x3) everywhere);
# SIVAE_WINO4_MAXC: largest channel count it takes (its U slab per 64-channel tile is 2.25x the F(2x2,3x3) one)
WINO4 = os.environ.get("SIVAE_WINO4", "1") != "0"
# SIVAE_WINO4_B6: the F(4x4,3x3) forward / data gradient on the bf16 matrix pipe with fp32-exact products (six bf16 MFMAs
# per fp32 product, conv_wino4_b6.hip) for layers with at least SIVAE_WINO4_B6_MINC input channels; 0: the fp32-MFMA kernel
WINO4_B6 = os.environ.get("SIVAE_WINO4_B6", "0") != "0"
WINO4_B6_MINC = int(os.environ.get("SIVAE_WINO4_B6_MINC", "16"))
WINO4_B6_PRO = os.environ.get("SIVAE_WINO4_B6_PRO", "1") != "0"


def _w4_key(b6, pro, sup):
    """KernelTimer key = the kernel rocprofv3 names: conv_wino4_kernel<PRO>, or conv_wino4_grid_kernel<PRO> on 8x8 / 4x4 maps"""
    p = "true" if pro is not None else "false"
    return ("conv_wino4_b6_kernel<%s>" if b6 else ("conv_wino4_grid_kernel<%s>" if sup >= 3 else "conv_wino4_kernel<%s>")) % p


def wino4_b6_takes(Ci, pro):
    return WINO4_B6 and Ci >= WINO4_B6_MINC and (pro is None or WINO4_B6_PRO)
WINO4_MAXC = int(os.environ.get("SIVAE_WINO4_MAXC", "512"))
# F(4x4,3x3) weight gradient (conv_wino4_wgrad.hip); SIVAE_WINO4_WGRAD=0: the F(2x2,3x3) one everywhere
WINO4_WGRAD = WINO4 and os.environ.get("SIVAE_WINO4_WGRAD", "1") != "0"
# SIVAE_WINO4_FORCE=1: take the F(4x4,3x3) kernels wherever they are SUPPORTED, not only where they pay (tests: the
# oracle comparisons run at batch 2-4, below the work-item thresholds)
WINO4_FORCE = os.environ.get("SIVAE_WINO4_FORCE", "0") == "1"
# SIVAE_WINO4_SPLITK=0: no split-K form of the F(4x4,3x3) kernel (launches below one work item per CU stay on F(2x2,3x3))
WINO4_SPLITK = os.environ.get("SIVAE_WINO4_SPLITK", "1") != "0"
# SIVAE_WINO4_SMALL=0: 8x8 / 4x4 maps stay on F(2x2,3x3) (the F(4x4,3x3) kernel runs them as grids of 8 / 32 images per item)
WINO4_SMALL = os.environ.get("SIVAE_WINO4_SMALL", "1") != "0"
WINO4_SMALL_FORCE = os.environ.get("SIVAE_WINO4_SMALL_FORCE", "0") == "1"  # (tests / tools: wherever it is supported)
# SIVAE_WINO4_UP_SMALL=0: convs of an upsampled input on 16x16 / 8x8 output maps stay on F(2x2,3x3) with upsample addressing
WINO4_UP_SMALL = os.environ.get("SIVAE_WINO4_UP_SMALL", "1") != "0"
# SIVAE_WINO4_DGRAD_POOL=0: the data gradient of an upsample-conv with <= 64 input channels stays on conv_wino_up_dgrad.hip
WINO4_DGRAD_POOL = os.environ.get("SIVAE_WINO4_DGRAD_POOL", "1") != "0"
WINO4_PRO = os.environ.get("SIVAE_WINO4_PRO", "1") != "0"  # ... also with a fused BatchNorm prologue (conv2 forward)
# SIVAE_FUSE_BN_BWD=1: reduce BatchNorm-1's backward sums in the epilogue of conv2's data gradient.  Measured a LOSS at
# 256x256 bs128 (593 vs 585 ms per iteration: the extra tensor read sits on the kernel's critical path and disables its
# next-item prefetch, which costs more than the 14 ms reduction pass it removes) -> off by default.
FUSE_BN_BWD = os.environ.get("SIVAE_FUSE_BN_BWD", "0") == "1"
WINO_WGRAD = os.environ.get("SIVAE_WINO_WGRAD", os.environ.get("SIVAE_WINO", "1")) != "0"
# streaming kernel for the 1x1 convs (SIVAE_CONV1_STREAM=0: the LDS-tiled direct kernel)
CONV1_STREAM = os.environ.get("SIVAE_CONV1_STREAM", "1") != "0"
# merged-contraction kernel for the 5x5 convs from <= 3 into <= 64 channels (SIVAE_CONV5_K75=0: the direct kernel)
CONV5_K75 = os.environ.get("SIVAE_CONV5_K75", "1") != "0"
# 1-bit LeakyReLU sign mask written by the block's last BatchNorm pass and read by its backward instead of the saved
# output (SIVAE_SIGNMASK=0: the backward reads the output tensor again)
SIGNMASK = os.environ.get("SIVAE_SIGNMASK", "1") != "0"


class PackedW:
    """Both GEMM-operand forms of one weight tensor (direct pack; Winograd transform for 3x3), built lazily.
    conv2d_fwd picks the kernel per call from the feature-map size."""

    def __init__(self, w, mode):
        self.w, self.mode = w, mode
        self._direct = self._wino = None

    def direct(self):
        if self._direct is None:
            self._direct = pack_weight(self.w, self.mode)
        return self._direct

    # operand forms sivae_pack_batch rebuilds in place: (form id of include/sivae_hip.h, attribute holding the buffer)
    _BATCH_FORMS = ((0, "_direct"), (1, "_wino"), (2, "_wino4"), (3, "_wino_up"), (4, "_wino_up_dgrad"),
                    (5, "_wino4_b6"))

    def batch_forms(self):
        """[(form id, buffer)] of the operand forms built so far that the batched repack can rebuild in place"""
        return [(f, getattr(self, a)) for f, a in self._BATCH_FORMS if getattr(self, a, None) is not None]

    def refreshed(self):
        """the weight changed and the batched repack rebuilt the forms of batch_forms() in place: drop the others (they
        are rebuilt on demand)"""
        self._k75 = None

    def wino(self):
        if self._wino is None:
            self._wino = pack_wino(self.w, self.mode)
        return self._wino

    def wino4(self):
        """F(4x4,3x3) filter transform [6][Ci_pad][Co_pad][6] (conv_wino4.hip)"""
        if getattr(self, "_wino4", None) is None:
            w = self.w
            Co, Ci = w.shape[0], w.shape[1]
            nbytes = _lib.load().sivae_pack_wino4_weight_bytes(Co, Ci, self.mode)
            up = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
            _lib.call("sivae_pack_wino4_weight", _p(w), _p(up), Co, Ci, self.mode, _s(w))
            self._wino4 = up
        return self._wino4

    def wino4_b6(self):
        """the F(4x4,3x3) filter transform pre-split into three bf16 pieces, MFMA-ready (conv_wino4_b6.hip)"""
        if getattr(self, "_wino4_b6", None) is None:
            w = self.w
            Co, Ci = w.shape[0], w.shape[1]
            nbytes = _lib.load().sivae_pack_wino4_b6_weight_bytes(Co, Ci, self.mode)
            up = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
            _lib.call("sivae_pack_wino4_b6_weight", _p(w), _p(up), Co, Ci, self.mode, _s(w))
            self._wino4_b6 = up
        return self._wino4_b6

    def wino_up_dgrad(self):
        if getattr(self, "_wino_up_dgrad", None) is None:
            assert self.mode == 0
            w = self.w
            Co, Ci = w.shape[0], w.shape[1]
            nbytes = _lib.load().sivae_pack_wino_up_dgrad_weight_bytes(Co, Ci)
            ud = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
            _lib.call("sivae_pack_wino_up_dgrad_weight", _p(w), _p(ud), Co, Ci, _s())
            self._wino_up_dgrad = ud
        return self._wino_up_dgrad

    def k75(self):
        """merged-contraction operand [76][64] of a 5x5 conv from <= 3 into <= 64 channels (conv5_k75.hip); mode 0:
        forward of w [Cb][Cs][5][5], mode 1: data gradient of w [Cs][Cb][5][5]"""
        if getattr(self, "_k75", None) is None:
            w = self.w
            n_big, n_small = (w.shape[0], w.shape[1]) if self.mode == 0 else (w.shape[1], w.shape[0])
            wq = torch.empty(_lib.load().sivae_pack_conv5_k75_bytes() // 4, dtype=torch.float32, device=w.device)
            _lib.call("sivae_pack_conv5_k75", _p(w), _p(wq), n_small, n_big, self.mode, _s(w))
            self._k75 = wq
        return self._k75

    def wino_up(self):
        """phase-decomposed F(2x2,2x2) transform for the conv-after-upsample forward kernel (mode 0 only)"""
        if getattr(self, "_wino_up", None) is None:
            assert self.mode == 0
            w = self.w
            Co, Ci = w.shape[0], w.shape[1]
            nbytes = _lib.load().sivae_pack_wino_up_weight_bytes(Co, Ci)
            up = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
            _lib.call("sivae_pack_wino_up_weight", _p(w), _p(up), Co, Ci, _s())
            self._wino_up = up
        return self._wino_up


def space_to_depth2(x):
    """[B, C, 2Hs, 2Ws] -> parity planes [B, 4, C, Hs, Ws] (plane 2p+q = x[..., p::2, q::2])"""
    _require(x)
    B, C, H, W = x.shape
    out = torch.empty((B, 4, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
    _lib.call("sivae_space_to_depth2", _p(x), _p(out), B, C, H // 2, W // 2, _s())
    return out


def conv2d_dgrad_bnbwd_supported(H, W):
    return WINO and FUSE_BN_BWD and SYNC_BN is None and _lib.load().sivae_conv2d_wino_supported(H, W) == 1


def conv2d_dgrad_bnbwd(dy, wp, Cm, bn_x, mean, invstd, gamma, beta, slope=LRELU_SLOPE):
    """3x3 data gradient dh = dgrad(dy) whose epilogue also reduces the BatchNorm-backward sums of
    h = LeakyReLU(BN(bn_x)): -> (dh, partials [n_tiles, Cm, 2])"""
    _require(dy, bn_x, mean, invstd, gamma, beta)
    B, Ci, H, W = dy.shape
    L = _lib.load()
    dh = torch.empty((B, Cm, H, W), dtype=torch.float32, device=dy.device)
    part = torch.empty((L.sivae_conv2d_wino_num_px_tiles(B, H, W), Cm, 2), dtype=torch.float32, device=dy.device)
    t0 = TIMER.begin() if TIMER is not None else None
    _lib.call("sivae_conv2d_wino_dgrad_bnbwd", _p(dy), _p(wp.wino()), _p(dh), _p(bn_x), _p(mean), _p(invstd), _p(gamma),
              _p(beta), float(slope), _p(part), B, Ci, Cm, H, W, _s())
    if t0 is not None:
        flops = 2.0 * B * H * W * Cm * Ci * 9
        TIMER.end("conv_wino_kernel<%s,false>" % ("1,4" if W >= 32 else ("2,3" if W >= 16 else ("2,2" if W == 8 else "1,1"))),
                  flops, t0, executed=flops * 16.0 / 36.0)
    return dh, part


def bn_bwd_from_partials(dy, x, mean, invstd, gamma, beta, partials, slope=LRELU_SLOPE, want_param_grads=True,
                         pg_out=None):
    """BatchNorm(+LeakyReLU, sign recomputed from x) backward whose reduction pass was done by the producer of dy"""
    _require(dy, x, mean, invstd, gamma, beta, partials)
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    ws = workspace(_lib.load().sivae_bn_workspace_bytes(B, C, HW), x.device)
    dx = torch.empty_like(x)
    dgamma, dbeta = _pg(pg_out, C, x.device, want_param_grads)
    _lib.call("sivae_bn_bwd_from_partials", _p(dy), _p(x), _p(mean), _p(invstd), _p(gamma), _p(beta), float(slope),
              _p(partials), partials.shape[0], _p(dx), _p(dgamma), _p(dbeta), B, C, HW, _p(ws), ws.numel(), _s())
    return dx, dgamma, dbeta


def conv2d_up_dgrad_supported(Hs, Ws):
    return WINO_UP and _lib.load().sivae_conv2d_wino_up_dgrad_supported(Hs, Ws) == 1


def conv2d_up_dgrad(dy, wp, N, out=None, accumulate=False, wp1=None):
    """gradient of conv3x3(Upsample2(x)) with respect to the LOW-resolution x: dy [B, C, 2Hs, 2Ws] -> [B, N, Hs, Ws]
    (wp: PackedW of the conv's weight [C, N, 3, 3], mode 0; wp1: the mode-1 PackedW of the same weight — with it the
    launches with N <= 64 take the F(4x4,3x3) kernel with the 2x2 block sum folded into its output transform)."""
    _require(dy, out)
    B, C, H, W = dy.shape
    Hs, Ws = H // 2, W // 2
    dx = out if out is not None else torch.empty((B, N, Hs, Ws), dtype=torch.float32, device=dy.device)
    assert dx.shape == (B, N, Hs, Ws)
    t0 = TIMER.begin() if TIMER is not None else None
    L = _lib.load()
    if (WINO4_DGRAD_POOL and WINO4 and wp1 is not None and max(C, N) <= WINO4_MAXC
            and L.sivae_conv2d_wino4_dgrad_pool_pays(B, C, N, H, W) == 1):
        assert dx.is_contiguous()
        _lib.call("sivae_conv2d_wino4_dgrad_pool", _p(dy), _p(wp1.wino4()), _p(dx), B, C, N, H, W, int(bool(accumulate)),
                  _s(dy))
        if t0 is not None:
            flops = 2.0 * B * H * W * C * N * 9
            TIMER.end("conv_wino4_pool_kernel<false>", flops, t0, executed=flops * 36.0 / 144.0)
        return dx
    if L.sivae_conv2d_wino_up_dgrad_splitk(B, C, N, Hs, Ws) > 1:  # small shards: split the 4C input planes
        ws = workspace(L.sivae_conv2d_wino_up_dgrad_splitk_workspace_bytes(B, C, N, Hs, Ws), dy.device)
        _lib.call("sivae_conv2d_wino_up_dgrad_splitk_run", _p(dy), _p(wp.wino_up_dgrad()), _p(dx), B, C, N, Hs, Ws,
                  int(bool(accumulate)), _p(ws), ws.numel(), _s(dy))
    else:
        _lib.call("sivae_conv2d_wino_up_dgrad", _p(dy), _p(wp.wino_up_dgrad()), _p(dx), B, C, N, Hs, Ws,
                  int(bool(accumulate)), _s(dy))
    if t0 is not None:
        flops = 2.0 * B * H * W * C * N * 9
        TIMER.end("conv_wino_up_dgrad_kernel<%s>" % ("1,4" if Ws >= 32 else "2,3"), flops, t0, executed=flops * 9.0 / 36.0)
    return dx


def seg_prologue_supported(H, W):
    """maps on which a SEGMENTED batch (nseg > 1) can keep the BatchNorm prologue fused into conv2 and into conv2's
    weight gradient (the Winograd kernels carry per-segment parameter tables); elsewhere the block stores h"""
    L = _lib.load()
    return (WINO and WINO_WGRAD and L.sivae_conv2d_wino_supported(H, W) == 1
            and L.sivae_conv2d_wino_wgrad_supported(H, W) == 1)


def conv2d_fwd(x, wp, Co, ks, bias=None, pro=None, upsample=False, want_stats=False, out=None,
               accumulate=False, nseg=1):
    """x [B, Ci, H, W] (or [B, Ci, H/2, W/2] with upsample) -> y [B, Co, H, W] (+ stats partials).

    wp: a direct pack (tensor from pack_weight) or a PackedW; with a PackedW, 3x3 convs on maps the Winograd
    kernel supports (even H >= 8, even W >= 16) run sivae_conv2d_wino_fwd.
    pro = (mean, invstd, gamma, beta, slope): fused producer BatchNorm + LeakyReLU on load."""
    B, Ci, Hs, Ws = x.shape
    H, W = (2 * Hs, 2 * Ws) if upsample else (Hs, Ws)
    L = _lib.load()
    if (CONV5_K75 and ks == 5 and isinstance(wp, PackedW) and pro is None and not upsample and not accumulate
            and out is None and L.sivae_conv5_k75_supported(Ci, Co) == 1):
        # the encoder stem / the data gradient of Decoder.predict: whole contraction (K = 75) merged
        _require(x, bias)
        y = torch.empty((B, Co, H, W), dtype=torch.float32, device=x.device)
        stats = (torch.empty((L.sivae_conv5_k75_num_px_tiles(B, H, W), Co, 2), dtype=torch.float32, device=x.device)
                 if want_stats else None)
        t0 = TIMER.begin() if TIMER is not None else None
        _lib.call("sivae_conv5_k75_fwd", _p(x), _p(wp.k75()), _p(y), _p(bias), _p(stats), B, Ci, Co, H, W, _s(x))
        if t0 is not None:
            TIMER.end("conv5_k75_kernel", 2.0 * B * H * W * Co * Ci * 25, t0,
                      executed=2.0 * B * H * W * 64 * 76)  # (64 output rows x 76 contraction columns are issued)
        return (y, stats) if want_stats else y
    if (CONV1_STREAM and ks == 1 and bias is None and pro is None and not upsample and not want_stats
            and L.sivae_conv1x1_stream_supported(B, Ci, Co, H * W) == 1):
        # ResidualBlock.conv_expand (forward / data gradient): x streamed straight into the MFMA operand
        wd = wp.direct() if isinstance(wp, PackedW) else wp
        _require(x, wd, out)
        y = out if out is not None else torch.empty((B, Co, H, W), dtype=torch.float32, device=x.device)
        assert y.shape == (B, Co, H, W) and y.is_contiguous()
        t0 = TIMER.begin() if TIMER is not None else None
        _lib.call("sivae_conv1x1_stream", _p(x), _p(wd), _p(y), B, Ci, Co, H * W, int(bool(accumulate)), _s(x))
        if t0 is not None:
            TIMER.end("conv1x1_stream_kernel", 2.0 * B * H * W * Co * Ci, t0)
        return y
    w4_sup = L.sivae_conv2d_wino4_supported(H, W) if (WINO and WINO4) else 0
    if w4_sup > 2 and not (WINO4_SMALL and (WINO4_SMALL_FORCE or L.sivae_conv2d_wino4_small_pays(B, Ci, Co, H, W) == 1)):
        w4_sup = 0  # (8x8 / 4x4 maps: only where the image-grid launch beats F(2x2,3x3) — few items or short K slices do not)
    w4_ipi = L.sivae_conv2d_wino4_images_per_item(H, W) if w4_sup else 1
    # an upsampled input: the phase-form kernels (conv_wino_up.hip) cover the maps from 32x32 up; on the 16x16 / 8x8 outputs of
    # the deep decoder blocks (Decoder res_in_16 / res_in_8, train_soft_intro_vae.py:153-158) the upsampled tensor is a
    # few MB — materialise it (one small launch) and take the F(4x4,3x3) image-grid kernel instead of F(2x2,3x3)
    w4_up = bool(upsample) and w4_sup >= 2 and WINO4_UP_SMALL and pro is None
    w4_ok = (w4_sup and ks == 3 and bias is None and (not upsample or w4_up) and isinstance(wp, PackedW)
             and 16 <= Ci and max(Ci, Co) <= WINO4_MAXC
             and (B // nseg) % w4_ipi == 0  # (maps up to 16x16: whole image grids per work item, inside one segment)
             and (pro is None or (WINO4_PRO and nseg * ((Ci + 31) // 32) * 32 <= 1024)))
    # fewer work items than CUs (the deep layers of a per-GPU shard): split over K when that fills the chip
    w4_S = (L.sivae_conv2d_wino4_splitk(B, Ci, Co, H, W) if (w4_ok and WINO4_SPLITK
                                                              and L.sivae_conv2d_wino4_pays(B, Ci, Co, H, W) != 1) else 1)
    if w4_ok and upsample:
        if w4_S > 1 or L.sivae_conv2d_wino4_pays(B, Ci, Co, H, W) == 1 or WINO4_FORCE:
            x, upsample = upsample2_fwd(x), False
        else:
            w4_ok = False
    if w4_ok and w4_S > 1:
        _require(x, out)
        y = out if out is not None else torch.empty((B, Co, H, W), dtype=torch.float32, device=x.device)
        assert y.shape == (B, Co, H, W) and y.is_contiguous()
        stats = torch.empty((B, Co, 2), dtype=torch.float32, device=x.device) if want_stats else None  # (rows per image)
        ws = workspace(L.sivae_conv2d_wino4_splitk_workspace_bytes(B, Ci, Co, H, W), x.device)
        pm = pi = pg = pb = None
        slope = 1.0
        if pro is not None:
            pm, pi, pg, pb, slope = pro
            _require(pm, pi, pg, pb)
        t0 = TIMER.begin() if TIMER is not None else None
        b6 = wino4_b6_takes(Ci, pro) and w4_sup <= 2
        _lib.call("sivae_conv2d_wino4_b6_fwd_splitk" if b6 else "sivae_conv2d_wino4_fwd_splitk", _p(x),
                  _p(wp.wino4_b6() if b6 else wp.wino4()), _p(y), _p(pm), _p(pi), _p(pg), _p(pb),
                  float(slope), _p(stats), B, Ci, Co, H, W, int(bool(accumulate)), (B // nseg) if nseg > 1 else 0,
                  _p(ws), ws.numel(), _s(x))
        if t0 is not None:
            flops = 2.0 * B * H * W * Co * Ci * 9
            TIMER.end(_w4_key(b6, pro, w4_sup), flops, t0, executed=flops * 36.0 / 144.0)
        return (y, stats) if want_stats else y
    if w4_ok and (L.sivae_conv2d_wino4_pays(B, Ci, Co, H, W) == 1 or WINO4_FORCE):
        # large maps: F(4x4,3x3) — 2.25 multiplies per output pixel instead of 4
        _require(x, out)
        y = out if out is not None else torch.empty((B, Co, H, W), dtype=torch.float32, device=x.device)
        assert y.shape == (B, Co, H, W) and y.is_contiguous()
        stats = (torch.empty((L.sivae_conv2d_wino4_num_px_tiles(B, H, W), Co, 2), dtype=torch.float32, device=x.device)
                 if want_stats else None)
        t0 = TIMER.begin() if TIMER is not None else None
        b6 = wino4_b6_takes(Ci, pro) and w4_sup <= 2
        if b6:
            pm = pi = pg = pb = None
            slope = 1.0
            if pro is not None:
                pm, pi, pg, pb, slope = pro
                _require(pm, pi, pg, pb)
            _lib.call("sivae_conv2d_wino4_b6_fwd", _p(x), _p(wp.wino4_b6()), _p(y), _p(pm), _p(pi), _p(pg), _p(pb),
                      float(slope), _p(stats), B, Ci, Co, H, W, int(bool(accumulate)), (B // nseg) if nseg > 1 else 0,
                      _s(x))
        elif pro is not None:
            pm, pi, pg, pb, slope = pro
            _require(pm, pi, pg, pb)
            _lib.call("sivae_conv2d_wino4_fwd_pro", _p(x), _p(wp.wino4()), _p(y), _p(pm), _p(pi), _p(pg), _p(pb),
                      float(slope), _p(stats), B, Ci, Co, H, W, int(bool(accumulate)), (B // nseg) if nseg > 1 else 0,
                      _s(x))
        else:
            _lib.call("sivae_conv2d_wino4_fwd", _p(x), _p(wp.wino4()), _p(y), _p(stats), B, Ci, Co, H, W,
                      int(bool(accumulate)), _s(x))
        if t0 is not None:
            flops = 2.0 * B * H * W * Co * Ci * 9
            TIMER.end(_w4_key(b6, pro, w4_sup), flops, t0, executed=flops * 36.0 / 144.0)
        return (y, stats) if want_stats else y
    wino = (WINO and ks == 3 and bias is None and isinstance(wp, PackedW)
            and L.sivae_conv2d_wino_supported(H, W) == 1)
    wino_up = (wino and WINO_UP and upsample and not accumulate and wp.mode == 0
               and L.sivae_conv2d_wino_up_supported(H, W) == 1)
    if isinstance(wp, PackedW):
        wp = wp.wino_up() if wino_up else (wp.wino() if wino else wp.direct())
    _require(x, wp, bias, out)
    y = out if out is not None else torch.empty((B, Co, H, W), dtype=torch.float32, device=x.device)
    assert y.shape == (B, Co, H, W)
    # split-K for launches that would leave most of the chip idle (deep small maps at small batch)
    splitk = wino and not wino_up and L.sivae_conv2d_wino_splitk(B, Ci, Co, H, W) > 1
    stats = None
    if want_stats:
        nt = (L.sivae_conv2d_wino_up_num_px_tiles(B, H, W) if wino_up else
              (L.sivae_conv2d_wino_splitk_stats_rows(B, Ci, Co, H, W) if splitk else
               (L.sivae_conv2d_wino_num_px_tiles(B, H, W) if wino else L.sivae_conv2d_fwd_num_px_tiles(B, Co, H, W))))
        stats = torch.empty((nt, Co, 2), dtype=torch.float32, device=x.device)
    pm = pi = pg = pb = None
    slope = 1.0
    if pro is not None:
        pm, pi, pg, pb, slope = pro
        _require(pm, pi, pg, pb)
    if nseg > 1 and (pro is not None or want_stats) and not (wino or wino_up):
        raise ValueError("sivae_hip: a segmented batch needs the Winograd 3x3 kernels for fused BatchNorm work")
    if nseg > 1 and pro is not None and wino_up:
        raise ValueError("sivae_hip: no segmented prologue in the upsample-phase kernel")
    t0 = TIMER.begin() if TIMER is not None else None
    if wino_up:
        _lib.call("sivae_conv2d_wino_up_fwd", _p(x), _p(wp), _p(y), _p(pm), _p(pi), _p(pg), _p(pb), float(slope),
                  _p(stats), B, Ci, Co, H, W, _s())
    elif splitk:
        ws = workspace(L.sivae_conv2d_wino_splitk_workspace_bytes(B, Ci, Co, H, W), x.device)
        if nseg > 1:
            _lib.call("sivae_conv2d_wino_fwd_splitk_seg", _p(x), _p(wp), _p(y), _p(pm), _p(pi), _p(pg), _p(pb),
                      float(slope), _p(stats), B, Ci, Co, H, W, int(bool(upsample)), int(bool(accumulate)), B // nseg,
                      _p(ws), ws.numel(), _s())
        else:
            _lib.call("sivae_conv2d_wino_fwd_splitk", _p(x), _p(wp), _p(y), _p(pm), _p(pi), _p(pg), _p(pb),
                      float(slope), _p(stats), B, Ci, Co, H, W, int(bool(upsample)), int(bool(accumulate)), _p(ws),
                      ws.numel(), _s())
    elif wino and nseg > 1:
        _lib.call("sivae_conv2d_wino_fwd_seg", _p(x), _p(wp), _p(y), _p(pm), _p(pi), _p(pg), _p(pb), float(slope),
                  _p(stats), B, Ci, Co, H, W, int(bool(upsample)), int(bool(accumulate)), B // nseg, _s())
    elif wino:
        _lib.call("sivae_conv2d_wino_fwd", _p(x), _p(wp), _p(y), _p(bias), _p(pm), _p(pi), _p(pg), _p(pb),
                  float(slope), _p(stats), B, Ci, Co, H, W, int(bool(upsample)), int(bool(accumulate)), _s())
    else:
        _lib.call("sivae_conv2d_fwd", _p(x), _p(wp), _p(y), _p(bias), _p(pm), _p(pi), _p(pg), _p(pb), float(slope),
                  _p(stats), B, Ci, Co, H, W, ks, int(bool(upsample)), int(bool(accumulate)), _s())
    if t0 is not None:
        flops = 2.0 * B * H * W * Co * Ci * ks * ks  # algorithmic (reference nn.Conv2d) FLOPs
        if wino_up:
            TIMER.end("conv_wino_up_kernel<%s,%s>" % ("1,4" if W >= 64 else "2,3", "true" if pro is not None else "false"),
                      flops, t0, executed=flops * 9.0 / 36.0)
        elif wino:
            key = "conv_wino_kernel<%s,%s>" % ("1,4" if W >= 32 else ("2,3" if W >= 16 else ("2,2" if W == 8 else "1,1")),
                                               "true" if pro is not None else "false")
            TIMER.end(key, flops, t0, executed=flops * 16.0 / 36.0)
        else:
            TIMER.end(conv_fwd_kernel_key(ks, Co, pro is not None), flops, t0)
    return (y, stats) if want_stats else y


def conv2d_wgrad(x, dy, ks, pro=None, upsample=False, out=None, nseg=1):
    """-> dW [Co, Ci, ks, ks]  (written into `out` when given)"""
    _require(x, dy)
    B, Ci = x.shape[0], x.shape[1]
    _, Co, H, W = dy.shape
    L = _lib.load()
    if (WINO_WGRAD and WINO_UP and upsample and ks == 3 and pro is None
            and L.sivae_conv2d_wino_up_wgrad_supported(H // 2, W // 2) == 1):
        # conv after nearest-2x upsample: phase-form F(2x2,2x2) weight gradient on the low-resolution x
        ws = workspace(L.sivae_conv2d_wino_up_wgrad_workspace_bytes(B, Ci, Co, H // 2, W // 2), x.device)
        dw = _out(out, (Co, Ci, 3, 3), x.device)
        t0 = TIMER.begin() if TIMER is not None else None
        _lib.call("sivae_conv2d_wino_up_wgrad", _p(x), _p(dy), _p(dw), B, Ci, Co, H // 2, W // 2, _p(ws), ws.numel(),
                  _s())
        if t0 is not None:
            flops = 2.0 * B * H * W * Co * Ci * 9
            TIMER.end("wino_up_wgrad_kernel", flops, t0, executed=flops * 9.0 / 36.0)
        return dw
    w4g_ips = L.sivae_conv2d_wino4_wgrad_images_per_stage(H, W) if ks == 3 else 0  # (8x8 / 4x4 maps: 2 / 4 images per strip)
    # (an upsampled x on the 16x16 / 8x8 maps, which the phase-form weight gradient above does not take: materialised, as
    # in conv2d_fwd)
    w4g_up = bool(upsample) and WINO4_UP_SMALL and pro is None and H <= 16 and W <= 16
    if (WINO4_WGRAD and WINO_WGRAD and ks == 3 and (not upsample or w4g_up) and (pro is None or nseg <= 2)
            and max(Ci, Co) <= WINO4_MAXC and w4g_ips > 0 and (B // nseg) % w4g_ips == 0
            and (w4g_ips == 1 or WINO4_SMALL)
            and (L.sivae_conv2d_wino4_wgrad_pays(B, Ci, Co, H, W) == 1
                 or (WINO4_FORCE and min(Ci, Co) >= 16))):
        if upsample:
            x = upsample2_fwd(x)
        # Winograd F(4x4,3x3) weight gradient (conv_wino4_wgrad.hip): 36 instead of 64 multiplies per tile, co, ci
        ws = workspace(L.sivae_conv2d_wino4_wgrad_workspace_bytes(B, Ci, Co, H, W), x.device)
        dw = _out(out, (Co, Ci, 3, 3), x.device)
        pm = pi = pg = pb = None
        slope = 1.0
        if pro is not None:
            pm, pi, pg, pb, slope = pro
            _require(pm, pi, pg, pb)
        t0 = TIMER.begin() if TIMER is not None else None
        _lib.call("sivae_conv2d_wino4_wgrad", _p(x), _p(dy), _p(dw), _p(pm), _p(pi), _p(pg), _p(pb), float(slope), B, Ci,
                  Co, H, W, (B // nseg) if (nseg > 1 and pro is not None) else 0, _p(ws), ws.numel(), _s())
        if t0 is not None:
            flops = 2.0 * B * H * W * Co * Ci * 9
            TIMER.end("wino4_wgrad_kernel<%s,%s>" % ("true" if pro is not None else "false",
                                                     "true" if w4g_ips > 1 else "false"), flops, t0,
                      executed=flops * 36.0 / 144.0)
        return dw
    wino = WINO_WGRAD and ks == 3 and L.sivae_conv2d_wino_wgrad_supported(H, W) == 1
    nbytes = (L.sivae_conv2d_wino_wgrad_workspace_bytes(B, Ci, Co, H, W) if wino
              else L.sivae_conv2d_wgrad_workspace_bytes(B, Ci, Co, H, W, ks))
    ws = workspace(nbytes, x.device)
    dw = _out(out, (Co, Ci, ks, ks), x.device)
    pm = pi = pg = pb = None
    slope = 1.0
    if pro is not None:
        pm, pi, pg, pb, slope = pro
        _require(pm, pi, pg, pb)
    if nseg > 1 and pro is not None and not wino:
        raise ValueError("sivae_hip: a segmented batch needs the Winograd weight-gradient kernel for a fused prologue")
    t0 = TIMER.begin() if TIMER is not None else None
    if wino and nseg > 1 and pro is not None:
        _lib.call("sivae_conv2d_wino_wgrad_seg", _p(x), _p(dy), _p(dw), _p(pm), _p(pi), _p(pg), _p(pb), float(slope),
                  B, Ci, Co, H, W, int(bool(upsample)), B // nseg, _p(ws), ws.numel(), _s())
    elif wino:
        _lib.call("sivae_conv2d_wino_wgrad", _p(x), _p(dy), _p(dw), _p(pm), _p(pi), _p(pg), _p(pb), float(slope),
                  B, Ci, Co, H, W, int(bool(upsample)), _p(ws), ws.numel(), _s())
    else:
        _lib.call("sivae_conv2d_wgrad", _p(x), _p(dy), _p(dw), _p(pm), _p(pi), _p(pg), _p(pb), float(slope), B, Ci,
                  Co, H, W, ks, int(bool(upsample)), _p(ws), ws.numel(), _s())
    if t0 is not None:  # (includes the tiny slice-reduce launch that follows the MFMA kernel)
        flops = 2.0 * B * H * W * Co * Ci * ks * ks
        if wino:
            TIMER.end("wino_wgrad_kernel<%s,1>" % ("true" if pro is not None else "false"), flops, t0,
                      executed=flops * 16.0 / 36.0)
        else:
            TIMER.end(conv_wgrad_kernel_key(ks, Co, pro is not None), flops, t0)
    return dw


# ------------------------------------------------------------------------------------------------ linear
def linear_supported(B, K, N):
    return _lib.load().sivae_linear_supported(B, K, N) == 1


def linear_fwd(x, w, bias=None, relu=False):
    """y = x W^T + b (optionally ReLU) for a small batch: x [B, K], w [N, K] -> [B, N]"""
    _require(x, w, bias)
    B, K = x.shape
    N = w.shape[0]
    ws = workspace(_lib.load().sivae_linear_workspace_bytes(B, K, N), x.device)
    y = torch.empty((B, N), dtype=torch.float32, device=x.device)
    t0 = TIMER.begin() if TIMER is not None else None
    _lib.call("sivae_linear_fwd", _p(x), _p(w), _p(bias), _p(y), int(bool(relu)), B, K, N, _p(ws), ws.numel(), _s(x))
    if t0 is not None:
        TIMER.end("linear_fwd_kernel", 2.0 * B * K * N, t0)
    return y


def linear_dgrad(dy, w):
    _require(dy, w)
    B, N = dy.shape
    K = w.shape[1]
    ws = workspace(_lib.load().sivae_linear_workspace_bytes(B, K, N), dy.device)
    dx = torch.empty((B, K), dtype=torch.float32, device=dy.device)
    t0 = TIMER.begin() if TIMER is not None else None
    _lib.call("sivae_linear_dgrad", _p(dy), _p(w), _p(dx), B, K, N, _p(ws), ws.numel(), _s(dy))
    if t0 is not None:
        TIMER.end("linear_dgrad_kernel", 2.0 * B * K * N, t0)
    return dx


def linear_wgrad(dy, x, out=None):
    _require(dy, x)
    B, N = dy.shape
    K = x.shape[1]
    dw = _out(out, (N, K), dy.device)
    t0 = TIMER.begin() if TIMER is not None else None
    _lib.call("sivae_linear_wgrad", _p(dy), _p(x), _p(dw), B, K, N, _s(dy))
    if t0 is not None:
        TIMER.end("linear_wgrad_kernel", 2.0 * B * K * N, t0)
    return dw


# ------------------------------------------------------------------------------------------------ 5x5 edges
def pack5_smallco(w, mode):
    """mode 0: w [Cs<=3, Cb, 5, 5] (predict) ; mode 1: w [Cb, Cs<=3, 5, 5] (stem, data-gradient operand)"""
    _require(w)
    n_small, n_big = (w.shape[0], w.shape[1]) if mode == 0 else (w.shape[1], w.shape[0])
    nbytes = _lib.load().sivae_pack_conv5_smallco_bytes(n_small, n_big)
    if nbytes == 0:
        raise _lib.SivaeError("sivae_pack_conv5_smallco_bytes", -2)
    wq = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
    _lib.call("sivae_pack_conv5_smallco", _p(w), _p(wq), n_small, n_big, mode, _s())
    return wq


def conv5_smallco_fwd(x, wq, Co, bias=None):
    _require(x, wq, bias)
    B, Ci, H, W = x.shape
    y = torch.empty((B, Co, H, W), dtype=torch.float32, device=x.device)
    t0 = TIMER.begin() if TIMER is not None else None
    _lib.call("sivae_conv5_smallco_fwd", _p(x), _p(wq), _p(y), _p(bias), B, Ci, Co, H, W, _s())
    if t0 is not None:
        TIMER.end("conv5_smallco_fwd_kernel<9>", 2.0 * B * H * W * Co * Ci * 25, t0)
    return y


def conv5_edge_wgrad(x, dy, out=None):
    _require(x, dy)
    B, Ci, H, W = x.shape
    Co = dy.shape[1]
    ws = workspace(_lib.load().sivae_conv5_edge_wgrad_workspace_bytes(B, Ci, Co, H, W), x.device)
    dw = _out(out, (Co, Ci, 5, 5), x.device)
    t0 = TIMER.begin() if TIMER is not None else None
    _lib.call("sivae_conv5_edge_wgrad", _p(x), _p(dy), _p(dw), B, Ci, Co, H, W, _p(ws), ws.numel(), _s())
    if t0 is not None:
        TIMER.end("conv5_edge_wgrad_kernel<%s>" % ("true" if Co <= 3 else "false"), 2.0 * B * H * W * Co * Ci * 25, t0)
    return dw


# ------------------------------------------------------------------------------------------------ BN
def bn_stats(x, running_mean=None, running_var=None, num_batches_tracked=None, eps=1e-5, momentum=0.1):
    _require(x, running_mean, running_var, num_batches_tracked)
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    L = _lib.load()
    ws = workspace(L.sivae_bn_workspace_bytes(B, C, HW), x.device)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty(C, dtype=torch.float32, device=x.device)
    _lib.call("sivae_bn_stats", _p(x), B, C, HW, float(eps), float(momentum), _p(running_mean), _p(running_var),
              _p(num_batches_tracked), _p(mean), _p(invstd), _p(ws), ws.numel(), _s())
    return mean, invstd


# Synchronised BatchNorm (opt-in for data-parallel runs): SYNC_BN = None, or a callable
# `sync(sums_fp64[C, 2]) -> world_size` that all-reduces (SUM) the per-channel sums in place.  With it, the batch
# statistics (forward) and the two gradient means (backward) are those of the GLOBAL batch, so N shards of B/N
# reproduce the single-process batch-B numbers; the default (None) is local BatchNorm, what DDP does to this model.
SYNC_BN = None


def bn_stats_from_conv(partials, B, C, HW, running_mean=None, running_var=None, num_batches_tracked=None,
                       eps=1e-5, momentum=0.1, nseg=1, seg_rev=False):
    """nseg > 1: B = nseg * B_seg images, rows of `partials` in image order; -> mean, invstd of nseg * C entries
    ([nseg][C]); the running buffers get one update per segment (seg_rev: last segment first)"""
    _require(partials, running_mean, running_var, num_batches_tracked)
    mean = torch.empty(nseg * C, dtype=torch.float32, device=partials.device)
    invstd = torch.empty(nseg * C, dtype=torch.float32, device=partials.device)
    if nseg > 1:
        # a partial row covers a pixel tile of ONE image, or — image pairs on 16 x 16 maps, several images per tile on the
        # 8 x 8 / 4 x 4 maps — a whole number of images: it must never straddle two segments (their statistics would mix)
        rows = partials.shape[0]
        per_row = B // rows if rows < B else 1
        if rows % nseg or (rows < B and B % rows) or (rows >= B and rows % B) or (B // nseg) % per_row:
            raise ValueError("sivae_hip: %d statistics rows of %d images cannot be cut into %d segments" % (rows, B, nseg))
    if nseg > 1 and SYNC_BN is not None:
        raise RuntimeError("sivae_hip: segmented batches and synchronised BatchNorm do not combine")
    if SYNC_BN is None:
        # (thousands of partial rows — the large-map layers at batch 128 — are folded in two coalesced stages through a
        # scratch buffer; the call is the one-stage form below that)
        nws = _lib.load().sivae_bn_stats_from_conv_workspace_bytes(partials.shape[0], nseg, C)
        ws = workspace(nws, partials.device) if nws else None
        _lib.call("sivae_bn_stats_from_conv_ws", _p(partials), partials.shape[0], nseg, int(bool(seg_rev)), B // nseg,
                  C, HW, float(eps), float(momentum), _p(running_mean), _p(running_var), _p(num_batches_tracked),
                  _p(mean), _p(invstd), _p(ws), ws.numel() if ws is not None else 0, _s())
        return mean, invstd
    if SYNC_BN is not None:
        sums = torch.empty((C, 2), dtype=torch.float64, device=partials.device)
        _lib.call("sivae_bn_sums_from_conv", _p(partials), partials.shape[0], C, _p(sums), _s())
        world = SYNC_BN(sums)
        _lib.call("sivae_bn_finalize_sums", _p(sums), C, float(B) * HW * world, float(eps), float(momentum),
                  _p(running_mean), _p(running_var), _p(num_batches_tracked), _p(mean), _p(invstd), _s())
        return mean, invstd
    _lib.call("sivae_bn_stats_from_conv", _p(partials), partials.shape[0], B, C, HW, float(eps), float(momentum),
              _p(running_mean), _p(running_var), _p(num_batches_tracked), _p(mean), _p(invstd), _s())
    return mean, invstd


def bn_update_running(mean, invstd, count, running_mean, running_var, num_batches_tracked, eps=1e-5, momentum=0.1,
                      nseg=1, seg_rev=False):
    _require(mean, invstd, running_mean, running_var, num_batches_tracked)
    if nseg > 1:
        _lib.call("sivae_bn_update_running_seg", _p(mean), _p(invstd), nseg, int(bool(seg_rev)), mean.numel() // nseg,
                  float(count), float(eps), float(momentum), _p(running_mean), _p(running_var),
                  _p(num_batches_tracked), _s())
        return
    _lib.call("sivae_bn_update_running", _p(mean), _p(invstd), mean.numel(), float(count), float(eps),
              float(momentum), _p(running_mean), _p(running_var), _p(num_batches_tracked), _s())


def bn_apply_act(x, res, mean, invstd, gamma, beta, slope=LRELU_SLOPE, out=None, res_up=False, nseg=1):
    """res_up: res is [B, C, H/2, W/2] and is added through nearest-2x upsample addressing"""
    _require(x, res, mean, invstd, gamma, beta, out)
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    y = out if out is not None else torch.empty_like(x)
    if nseg > 1:
        H, W = (x.shape[2], x.shape[3]) if x.dim() == 4 else (1, HW)
        _lib.call("sivae_bn_apply_act_seg", _p(x), _p(res), int(bool(res_up)), _p(mean), _p(invstd), _p(gamma),
                  _p(beta), float(slope), _p(y), None, B, C, H, W, B // nseg, _s())
        return y
    if res_up:
        H, W = x.shape[2], x.shape[3]
        assert res.shape == (B, C, H // 2, W // 2)
        _lib.call("sivae_bn_apply_act_resup", _p(x), _p(res), _p(mean), _p(invstd), _p(gamma), _p(beta), float(slope),
                  _p(y), B, C, H, W, _s())
        return y
    _lib.call("sivae_bn_apply_act", _p(x), _p(res), _p(mean), _p(invstd), _p(gamma), _p(beta), float(slope), _p(y),
              B, C, HW, _s())
    return y


def bn_apply_act_pool(x, res, mean, invstd, gamma, beta, slope=LRELU_SLOPE, want_full=True, nseg=1):
    """-> (y, AvgPool2d(2)(y)) in one pass (y is None with want_full=False); None when the shape is not covered
    (odd H or W % 4 != 0)"""
    B, C, H, W = x.shape
    if (H & 1) or (W & 3):
        return None
    _require(x, res, mean, invstd, gamma, beta)
    y = torch.empty_like(x) if want_full else None
    yp = torch.empty((B, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
    if nseg > 1:
        _lib.call("sivae_bn_apply_act_seg", _p(x), _p(res), 0, _p(mean), _p(invstd), _p(gamma), _p(beta), float(slope),
                  _p(y), _p(yp), B, C, H, W, B // nseg, _s())
        return y, yp
    _lib.call("sivae_bn_apply_act_pool", _p(x), _p(res), _p(mean), _p(invstd), _p(gamma), _p(beta), float(slope),
              _p(y), _p(yp), B, C, H, W, _s())
    return y, yp


def bn_signmask_supported(x):
    """the 1-bit LeakyReLU sign mask path: local BatchNorm, H even, W % 8 == 0"""
    return SIGNMASK and SYNC_BN is None and x.dim() == 4 and not (x.shape[2] & 1) and not (x.shape[3] & 7)


def bn_apply_act_signmask(x, res, mean, invstd, gamma, beta, slope=LRELU_SLOPE, res_up=False, pool=False,
                          want_full=True, nseg=1):
    """LeakyReLU(BN(x) + res) -> (y, y_pooled, mask): y is None with want_full=False (pool only), y_pooled is None
    without pool; mask = the activation's sign bits (uint8, 1 bit per element) for bn_bwd_signmask"""
    _require(x, res, mean, invstd, gamma, beta)
    B, C, H, W = x.shape
    assert pool or want_full
    y = torch.empty_like(x) if want_full else None
    yp = torch.empty((B, C, H // 2, W // 2), dtype=torch.float32, device=x.device) if pool else None
    mask = torch.empty(_lib.load().sivae_bn_signmask_bytes(B, C, H * W), dtype=torch.uint8, device=x.device)
    if nseg > 1:
        _lib.call("sivae_bn_apply_act_signmask_seg", _p(x), _p(res), int(bool(res_up)), _p(mean), _p(invstd),
                  _p(gamma), _p(beta), float(slope), _p(y), _p(yp), _p(mask), B, C, H, W, B // nseg, _s())
        return y, yp, mask
    _lib.call("sivae_bn_apply_act_signmask", _p(x), _p(res), int(bool(res_up)), _p(mean), _p(invstd), _p(gamma),
              _p(beta), float(slope), _p(y), _p(yp), _p(mask), B, C, H, W, _s())
    return y, yp, mask


def _bn_bwd_seg(dy, y, mask, x, mean, invstd, gamma, beta, act_mode, slope, dx, dz, dgamma, dbeta, dy_pooled, dz_sum,
                nseg):
    """every BatchNorm-backward variant (local statistics), segmented or not (mean / invstd [nseg][C]), with the
    per-channel finalize fused into the reduction kernel"""
    if SYNC_BN is not None:
        raise RuntimeError("sivae_hip: segmented batches and synchronised BatchNorm do not combine")
    B, C = x.shape[0], x.shape[1]
    H, W = (x.shape[2], x.shape[3]) if x.dim() == 4 else (1, x.numel() // (B * C))
    ws = workspace(_lib.load().sivae_bn_workspace_bytes(B // nseg, nseg * C, H * W), x.device)
    cnt = counters(x.device) if (BN_FUSED_FINALIZE and C <= 8192) else None
    _lib.call("sivae_bn_bwd_seg", _p(dy), _p(y), _p(mask), _p(x), _p(mean), _p(invstd), _p(gamma), _p(beta),
              int(act_mode), float(slope), _p(dx), _p(dz), _p(dgamma), _p(dbeta), B, C, H, W, int(bool(dy_pooled)),
              int(bool(dz_sum)), B // nseg, _p(cnt), _p(ws), ws.numel(), _s())


def bn_bwd_signmask(dy, mask, x, mean, invstd, gamma, slope=LRELU_SLOPE, dy_pooled=False, dz_sum=False,
                    want_dz=True, want_param_grads=True, pg_out=None, nseg=1):
    """backward of bn_apply_act_signmask -> dx, dz (full resolution, or its 2x2 block sums with dz_sum), dgamma,
    dbeta.  dy_pooled: dy is the gradient of the pooled output."""
    _require(dy, x, mean, invstd, gamma)
    B, C, H, W = x.shape
    if mask.dtype != torch.uint8 or not mask.is_cuda or mask.numel() < _lib.load().sivae_bn_signmask_bytes(B, C, H * W):
        raise TypeError("sivae_hip: bn_bwd_signmask needs the uint8 device mask bn_apply_act_signmask returned")
    ws = workspace(_lib.load().sivae_bn_workspace_bytes(B, C, H * W), x.device)
    dx = torch.empty_like(x)
    dz = None
    if dz_sum:
        dz = torch.empty((B, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
    elif want_dz:
        dz = torch.empty_like(x)
    dgamma, dbeta = _pg(pg_out, C, x.device, want_param_grads)
    if _bn_bwd_fused_ok(x, nseg):
        _bn_bwd_fused(dy, None, mask, x, mean, invstd, gamma, None, 3, slope, dx, dz, dgamma, dbeta, dy_pooled, dz_sum,
                      nseg)
        return dx, dz, dgamma, dbeta
    if nseg > 1 or (BN_FUSED_FINALIZE and SYNC_BN is None):
        _bn_bwd_seg(dy, None, mask, x, mean, invstd, gamma, None, 3, slope, dx, dz, dgamma, dbeta, dy_pooled, dz_sum,
                    nseg)
        return dx, dz, dgamma, dbeta
    _lib.call("sivae_bn_bwd_signmask", _p(dy), _p(mask), _p(x), _p(mean), _p(invstd), _p(gamma), float(slope), _p(dx),
              _p(dz), _p(dgamma), _p(dbeta), B, C, H, W, int(bool(dy_pooled)), int(bool(dz_sum)), _p(ws), ws.numel(),
              _s())
    return dx, dz, dgamma, dbeta


def bn_bwd_dzsum_supported(x):
    return SYNC_BN is None and not (x.shape[2] & 1) and not (x.shape[3] & 3)


def bn_bwd_dzsum(dy, y, x, mean, invstd, gamma, slope=LRELU_SLOPE, want_param_grads=True, pg_out=None, nseg=1):
    """act_mode-1 BatchNorm(+residual+LeakyReLU) backward -> dx, dz_half (2x2 block sums of the residual-branch
    gradient, [B, C, H/2, W/2]), dgamma, dbeta"""
    _require(dy, y, x, mean, invstd, gamma)
    B, C, H, W = x.shape
    ws = workspace(_lib.load().sivae_bn_workspace_bytes(B, C, H * W), x.device)
    dx = torch.empty_like(x)
    dzh = torch.empty((B, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
    dgamma, dbeta = _pg(pg_out, C, x.device, want_param_grads)
    if _bn_bwd_fused_ok(x, nseg):
        _bn_bwd_fused(dy, y, None, x, mean, invstd, gamma, None, 1, slope, dx, dzh, dgamma, dbeta, False, True, nseg)
        return dx, dzh, dgamma, dbeta
    if nseg > 1 or (BN_FUSED_FINALIZE and SYNC_BN is None):
        _bn_bwd_seg(dy, y, None, x, mean, invstd, gamma, None, 1, slope, dx, dzh, dgamma, dbeta, False, True, nseg)
        return dx, dzh, dgamma, dbeta
    _lib.call("sivae_bn_bwd_dzsum", _p(dy), _p(y), _p(x), _p(mean), _p(invstd), _p(gamma), float(slope), _p(dx), _p(dzh),
              _p(dgamma), _p(dbeta), B, C, H, W, _p(ws), ws.numel(), _s())
    return dx, dzh, dgamma, dbeta


def bn_bwd(dy, y, x, mean, invstd, gamma, slope=LRELU_SLOPE, want_dz=False, want_param_grads=True, beta=None,
           act_mode=None, dy_pooled=False, pg_out=None, nseg=1):
    """-> dx, dz (or None), dgamma, dbeta (or None, None).
    act_mode: 0 none, 1 LeakyReLU sign from the saved output y, 2 sign recomputed from x (needs beta).
    dy_pooled: dy is the gradient of AvgPool2d(2)(output) at half resolution; the pool's adjoint is applied on load."""
    if act_mode is None:
        act_mode = 1 if y is not None else 0
    if dy_pooled and (SYNC_BN is not None or (x.shape[2] & 1) or (x.shape[3] & 3)):
        dy = avgpool2_bwd(dy, x.shape[2], x.shape[3])  # (shapes / modes the fused read does not cover)
        dy_pooled = False
    _require(dy, y, x, mean, invstd, gamma, beta)
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    L = _lib.load()
    ws = workspace(L.sivae_bn_workspace_bytes(B, C, HW), x.device)
    dx = torch.empty_like(x)
    dz = torch.empty_like(x) if want_dz else None
    dgamma, dbeta = _pg(pg_out, C, x.device, want_param_grads)
    if _bn_bwd_fused_ok(x, nseg):
        _bn_bwd_fused(dy, y, None, x, mean, invstd, gamma, beta, act_mode, slope, dx, dz, dgamma, dbeta, dy_pooled, False,
                      nseg)
        return dx, dz, dgamma, dbeta
    if nseg > 1 or (BN_FUSED_FINALIZE and SYNC_BN is None and x.dim() == 4):
        _bn_bwd_seg(dy, y, None, x, mean, invstd, gamma, beta, act_mode, slope, dx, dz, dgamma, dbeta, dy_pooled, False,
                    nseg)
        return dx, dz, dgamma, dbeta
    if SYNC_BN is not None:
        local = torch.empty((C, 2), dtype=torch.float64, device=x.device)
        _lib.call("sivae_bn_bwd_reduce", _p(dy), _p(y), _p(x), _p(mean), _p(invstd), _p(gamma), _p(beta),
                  int(act_mode), float(slope), _p(local), B, C, HW, _p(ws), ws.numel(), _s())
        glob = local.clone()
        world = SYNC_BN(glob)
        _lib.call("sivae_bn_bwd_apply", _p(dy), _p(y), _p(x), _p(mean), _p(invstd), _p(gamma), _p(beta),
                  int(act_mode), float(slope), _p(local), _p(glob), float(B) * HW * world, _p(dx), _p(dz),
                  _p(dgamma), _p(dbeta), B, C, HW, _p(ws), ws.numel(), _s())
        return dx, dz, dgamma, dbeta
    if dy_pooled:
        _lib.call("sivae_bn_bwd_pooled_dy", _p(dy), _p(y), _p(x), _p(mean), _p(invstd), _p(gamma), _p(beta),
                  int(act_mode), float(slope), _p(dx), _p(dz), _p(dgamma), _p(dbeta), B, C, x.shape[2], x.shape[3],
                  _p(ws), ws.numel(), _s())
        return dx, dz, dgamma, dbeta
    _lib.call("sivae_bn_bwd", _p(dy), _p(y), _p(x), _p(mean), _p(invstd), _p(gamma), _p(beta), int(act_mode),
              float(slope), _p(dx), _p(dz), _p(dgamma), _p(dbeta), B, C, HW, _p(ws), ws.numel(), _s())
    return dx, dz, dgamma, dbeta


def channel_sum(x, out=None):
    _require(x)
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    L = _lib.load()
    ws = workspace(L.sivae_bn_workspace_bytes(B, C, HW), x.device)
    out = _out(out, (C,), x.device)
    _lib.call("sivae_channel_sum", _p(x), _p(out), B, C, HW, _p(ws), ws.numel(), _s())
    return out


# ------------------------------------------------------------------------------------------------ eltwise
def avgpool2_fwd(x):
    _require(x)
    B, C, H, W = x.shape
    y = torch.empty((B, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
    _lib.call("sivae_avgpool2_fwd", _p(x), _p(y), B * C, H, W, _s())
    return y


def avgpool2_bwd(dy, H, W):
    _require(dy)
    B, C = dy.shape[0], dy.shape[1]
    dx = torch.empty((B, C, H, W), dtype=torch.float32, device=dy.device)
    _lib.call("sivae_avgpool2_bwd", _p(dy), _p(dx), B * C, H, W, _s())
    return dx


def upsample2_fwd(x):
    _require(x)
    B, C, H, W = x.shape
    y = torch.empty((B, C, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    _lib.call("sivae_upsample2_fwd", _p(x), _p(y), B * C, H, W, _s())
    return y


def upsample2_bwd(dy):
    _require(dy)
    B, C, H2, W2 = dy.shape
    dx = torch.empty((B, C, H2 // 2, W2 // 2), dtype=torch.float32, device=dy.device)
    _lib.call("sivae_upsample2_bwd", _p(dy), _p(dx), B * C, H2 // 2, W2 // 2, _s())
    return dx


def relu_fwd(x, inplace=False):
    _require(x)
    y = x if inplace else torch.empty_like(x)
    _lib.call("sivae_relu_fwd", _p(x), _p(y), x.numel(), _s())
    return y


def relu_bwd(dy, y):
    _require(dy, y)
    dx = torch.empty_like(dy)
    _lib.call("sivae_relu_bwd", _p(dy), _p(y), _p(dx), dy.numel(), _s())
    return dx


def add_(y, x):
    _require(y, x)
    assert y.numel() == x.numel()
    _lib.call("sivae_add_inplace", _p(y), _p(x), y.numel(), _s())
    return y


# ------------------------------------------------------------------------------------------------ losses
def _ld(t):
    """[B, Z] view with unit inner stride -> leading dimension"""
    assert t.dim() == 2 and t.stride(1) == 1, "mu/logvar must be [B, Z] with unit inner stride"
    if not t.is_cuda:
        raise RuntimeError("sivae_hip: tensor is on %s — no CPU fallback" % t.device)
    return t.stride(0)


def reparam_fwd(mu, logvar, eps):
    _require(eps)
    B, Z = mu.shape
    ld = _ld(mu)
    assert _ld(logvar) == ld
    z = torch.empty((B, Z), dtype=torch.float32, device=mu.device)
    _lib.call("sivae_reparam_fwd", _p(mu), _p(logvar), ld, _p(eps), _p(z), B, Z, _s())
    return z


def reparam_bwd(dz, logvar, eps):
    _require(dz, eps)
    B, Z = dz.shape
    dmu = torch.empty_like(dz)
    dlv = torch.empty_like(dz)
    _lib.call("sivae_reparam_bwd", _p(dz), _p(logvar), _ld(logvar), _p(eps), _p(dmu), _p(dlv), Z, B, Z, _s())
    return dmu, dlv


def kl_fwd(logvar, mu, mu_o=0.0, logvar_o=0.0):
    B, Z = mu.shape
    ld = _ld(mu)
    assert _ld(logvar) == ld
    out = torch.empty(B, dtype=torch.float32, device=mu.device)
    _lib.call("sivae_kl_fwd", _p(logvar), _p(mu), ld, float(mu_o), float(logvar_o), _p(out), B, Z, _s())
    return out


def kl_bwd(g, per_sample, g_scale, logvar, mu, mu_o=0.0, logvar_o=0.0):
    _require(g)
    B, Z = mu.shape
    ld = _ld(mu)
    dlv = torch.empty((B, Z), dtype=torch.float32, device=mu.device)
    dmu = torch.empty((B, Z), dtype=torch.float32, device=mu.device)
    _lib.call("sivae_kl_bwd", _p(g), int(bool(per_sample)), float(g_scale), _p(logvar), _p(mu), ld, float(mu_o),
              float(logvar_o), _p(dlv), _p(dmu), Z, B, Z, _s())
    return dlv, dmu


def _prior(t, B, Z, like):
    """calc_kl prior as (fp32 device tensor, row stride, column stride) broadcastable to [B, Z]"""
    t = t.detach().to(device=like.device, dtype=torch.float32)
    if t.dim() > 2:
        raise ValueError("calc_kl: prior of shape %s does not broadcast to [B, Z]" % (tuple(t.shape),))
    while t.dim() < 2:
        t = t.unsqueeze(0)
    r, c = t.shape
    if r not in (1, B) or c not in (1, Z):
        raise ValueError("calc_kl: prior of shape %s does not broadcast to [%d, %d]" % (tuple(t.shape), B, Z))
    t = t.contiguous()
    return t, (c if r == B else 0), (1 if c == Z else 0)


def kl_fwd_t(logvar, mu, mu_o, logvar_o):
    B, Z = mu.shape
    ld = _ld(mu)
    assert _ld(logvar) == ld
    mo, mo_rs, mo_cs = _prior(mu_o, B, Z, mu)
    lo, lo_rs, lo_cs = _prior(logvar_o, B, Z, mu)
    out = torch.empty(B, dtype=torch.float32, device=mu.device)
    _lib.call("sivae_kl_fwd_t", _p(logvar), _p(mu), ld, _p(mo), mo_rs, mo_cs, _p(lo), lo_rs, lo_cs, _p(out), B, Z,
              _s(mu))
    return out


def kl_bwd_t(g, per_sample, g_scale, logvar, mu, mu_o, logvar_o):
    _require(g)
    B, Z = mu.shape
    ld = _ld(mu)
    mo, mo_rs, mo_cs = _prior(mu_o, B, Z, mu)
    lo, lo_rs, lo_cs = _prior(logvar_o, B, Z, mu)
    dlv = torch.empty((B, Z), dtype=torch.float32, device=mu.device)
    dmu = torch.empty((B, Z), dtype=torch.float32, device=mu.device)
    _lib.call("sivae_kl_bwd_t", _p(g), int(bool(per_sample)), float(g_scale), _p(logvar), _p(mu), ld, _p(mo), mo_rs,
              mo_cs, _p(lo), lo_rs, lo_cs, _p(dlv), _p(dmu), Z, B, Z, _s(mu))
    return dlv, dmu


def recon_rowsum_fwd(x, recon, loss_type):
    _require(x, recon)
    B = x.shape[0]
    D = x.numel() // B
    L = _lib.load()
    ws = workspace(L.sivae_recon_workspace_bytes(B, D), x.device)
    out = torch.empty(B, dtype=torch.float32, device=x.device)
    _lib.call("sivae_recon_rowsum_fwd", _p(x), _p(recon), LOSS_TYPES[loss_type], _p(out), B, D, _p(ws), ws.numel(),
              _s())
    return out


def recon_bwd(x, recon, loss_type, g, g_mode, g_scale, want_drecon=True, want_dx=False):
    _require(x, recon, g)
    B = x.shape[0]
    D = x.numel() // B
    d_r = torch.empty_like(recon) if want_drecon else None
    d_x = torch.empty_like(x) if want_dx else None
    _lib.call("sivae_recon_bwd", _p(x), _p(recon), LOSS_TYPES[loss_type], _p(g), int(g_mode), float(g_scale),
              _p(d_r), _p(d_x), B, D, _s())
    return d_r, d_x


def recon_elem_fwd(x, recon, loss_type):
    _require(x, recon)
    out = torch.empty_like(x)
    _lib.call("sivae_recon_elem_fwd", _p(x), _p(recon), LOSS_TYPES[loss_type], _p(out), x.numel(), _s())
    return out


def vec_sum(v, scale=1.0):
    _require(v)
    out = torch.empty((), dtype=torch.float32, device=v.device)
    _lib.call("sivae_vec_sum", _p(v), v.numel(), float(scale), _p(out), _s())
    return out


def expelbo_fwd(L_, KL, scale, beta_rec, beta_neg):
    _require(L_, KL)
    B = L_.numel()
    e = torch.empty(B, dtype=torch.float32, device=L_.device)
    out = torch.empty((), dtype=torch.float32, device=L_.device)
    _lib.call("sivae_expelbo_fwd", _p(L_), _p(KL), float(scale), float(beta_rec), float(beta_neg), B, _p(e), _p(out),
              _s())
    return out, e


def expelbo_bwd(gout, e, scale, beta_rec, beta_neg):
    _require(gout, e)
    B = e.numel()
    dL = torch.empty_like(e)
    dKL = torch.empty_like(e)
    _lib.call("sivae_expelbo_bwd", _p(gout), _p(e), float(scale), float(beta_rec), float(beta_neg), B, _p(dL),
              _p(dKL), _s())
    return dL, dKL


def lincomb(ts, ws):
    """sum_i ws[i] * ts[i] of up to six device scalars, one launch (lossE / lossD)"""
    n = len(ts)
    _require(*ts)
    out = torch.empty((), dtype=torch.float32, device=ts[0].device)
    ps = [_p(t) for t in ts] + [None] * (6 - n)
    w = [float(v) for v in ws] + [0.0] * (6 - n)
    _lib.call("sivae_lincomb", *ps, *w, n, _p(out), _s())
    return out


def lincomb_bwd(g, ws):
    n = len(ws)
    _require(g)
    out = torch.empty(n, dtype=torch.float32, device=g.device)
    w = [float(v) for v in ws] + [0.0] * (6 - n)
    _lib.call("sivae_lincomb_bwd", _p(g), *w, n, _p(out), _s())
    return out


def randn(shape, seed, offset, device):
    out = torch.empty(shape, dtype=torch.float32, device=device)
    _require(out)
    _lib.call("sivae_randn", _p(out), out.numel(), int(seed) & 0xFFFFFFFFFFFFFFFF, int(offset) & 0xFFFFFFFFFFFFFFFF,
              _s())
    return out


def randn_dev(shape, seed, offset_dev, device):
    """normals from the Philox stream whose position is the device scalar offset_dev (uint64 stored in an int64
    tensor); the position is advanced on the device"""
    out = torch.empty(shape, dtype=torch.float32, device=device)
    _lib.call("sivae_randn_dev", _p(out), out.numel(), int(seed) & 0xFFFFFFFFFFFFFFFF, _p(offset_dev), _s())
    return out


# ------------------------------------------------------------------------------------------------ optimizer
def adam_step_dev(param, grad, exp_avg, exp_avg_sq, state, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    """state: float64 device tensor [4] = {t, lr, step_size, sqrt(bias_correction2)} (t and the factors are advanced
    on the device)"""
    _require(param, grad, exp_avg, exp_avg_sq)
    _lib.call("sivae_adam_step_dev", _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), _p(state),
              float(beta1), float(beta2), float(eps), float(grad_scale), _s())


def sum_slabs(flat_grad, slabs):
    """flat_grad += slabs[0] + slabs[1] + ... (<= 4 slabs; the per-use parameter-gradient slabs of optim.FlatAdam)"""
    _require(flat_grad, *slabs)
    n = flat_grad.numel()
    assert 1 <= len(slabs) <= 4 and all(t.numel() == n for t in slabs)
    ptrs = [_p(t) for t in slabs] + [None] * (4 - len(slabs))
    _lib.call("sivae_sum_slabs", _p(flat_grad), ptrs[0], ptrs[1], ptrs[2], ptrs[3], n, _s(flat_grad))


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    _require(param, grad, exp_avg, exp_avg_sq)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    _lib.call("sivae_adam_step", _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), float(lr / bc1),
              float(beta1), float(beta2), float(eps), float(bc2 ** 0.5), float(grad_scale), _s())
