"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol declared in
include/sivae_hip.h, argument errors are reported through the documented codes (no GPU needed: validation
happens before any launch), and the Python host mirror has the reference's surface."""
import ctypes
import inspect
import os

import numpy as np
import pytest
import torch

from sivae_hip import lib


def test_library_exports_every_declared_symbol():
    L = lib.load()
    protos = lib.parse_header()
    assert len(protos) >= 38
    for name in protos:
        assert hasattr(L, name), name
    assert L.sivae_abi_version() == 1
    assert L.sivae_arch() == b"gfx950"
    assert L.sivae_device_count() >= 0
    # and nothing sivae_* is exported that the header does not declare
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib.LIB_PATH]).decode()
    exported = {l.split()[-1] for l in out.splitlines() if " T sivae_" in l}
    assert exported == set(protos), exported ^ set(protos)


def test_argument_errors_use_documented_codes():
    L = lib.load()
    null = None
    one = ctypes.c_void_p(16)  # never dereferenced: validation fails first
    assert L.sivae_conv2d_fwd(null, one, one, null, null, null, null, null, 0.2, null, 1, 1, 1, 4, 4, 3, 0, 0, null) == -1
    assert L.sivae_conv2d_fwd(one, one, one, null, null, null, null, null, 0.2, null, 1, 1, 1, 4, 4, 2, 0, 0, null) == -3
    assert L.sivae_conv2d_fwd(one, one, one, null, null, null, null, null, 0.2, null, 0, 1, 1, 4, 4, 3, 0, 0, null) == -2
    assert L.sivae_conv2d_fwd(one, one, one, null, null, null, null, null, 0.2, null, 1, 1, 1, 5, 5, 3, 1, 0, null) == -2
    assert L.sivae_conv2d_fwd(one, one, one, null, one, null, null, null, 0.2, null, 1, 1, 1, 4, 4, 3, 0, 0, null) == -1
    assert L.sivae_conv2d_wgrad(one, one, one, null, null, null, null, 0.2, 1, 8, 8, 4, 4, 3, 0, null, 0, null) == -4
    assert L.sivae_bn_bwd(one, null, one, one, one, one, null, 1, 0.2, one, null, null, null, 1, 1, 4, one, 1 << 20, null) == -1
    assert L.sivae_bn_bwd(one, one, one, one, one, one, null, 7, 0.2, one, null, null, null, 1, 1, 4, one, 1 << 20, null) == -6
    assert L.sivae_recon_rowsum_fwd(one, one, 9, one, 1, 4, one, 1 << 20, null) == -6
    assert L.sivae_pack_conv_weight(one, one, 4, 4, 7, 0, null) == -3
    assert L.sivae_pack_conv_weight(one, one, 4, 4, 3, 5, null) == -6


def test_round1_late_entry_points_validate_arguments():
    """sign-mask BatchNorm passes, the uint8 output quantisation and the Winograd shape predicates: pure host checks
    (every call returns before any kernel launch)"""
    L = lib.load()
    null = None
    one = ctypes.c_void_p(16)
    assert L.sivae_bn_signmask_bytes(2, 3, 16) == 12 and L.sivae_bn_signmask_bytes(1, 1, 4) == 1
    assert L.sivae_bn_signmask_bytes(0, 3, 16) == 0
    # mask missing / W % 8 != 0 / pooled output together with a half-resolution residual
    assert L.sivae_bn_apply_act_signmask(one, one, 0, one, one, one, one, 0.2, one, null, null, 1, 1, 4, 8, null) == -1
    assert L.sivae_bn_apply_act_signmask(one, one, 0, one, one, one, one, 0.2, one, null, one, 1, 1, 4, 12, null) == -2
    assert L.sivae_bn_apply_act_signmask(one, one, 1, one, one, one, one, 0.2, one, one, one, 1, 1, 4, 8, null) == -6
    assert L.sivae_bn_bwd_signmask(one, null, one, one, one, one, 0.2, one, null, null, null, 1, 1, 4, 8, 0, 0, one, 1 << 20, null) == -1
    assert L.sivae_bn_bwd_signmask(one, one, one, one, one, one, 0.2, one, one, null, null, 1, 1, 4, 8, 1, 1, one, 1 << 20, null) == -6
    assert L.sivae_bn_bwd_signmask(one, one, one, one, one, one, 0.2, one, null, null, null, 1, 1, 3, 8, 0, 0, one, 1 << 20, null) == -2
    assert L.sivae_f32_to_u8(null, one, 16, 255.0, null) == -1
    assert L.sivae_f32_to_u8(one, one, 0, 255.0, null) == -2
    assert L.sivae_f32_to_u8(ctypes.c_void_p(20), one, 16, 255.0, null) == -2  # 16-byte alignment
    assert L.sivae_conv2d_wino_supported(256, 256) == 1 and L.sivae_conv2d_wino_supported(4, 4) == 1
    assert L.sivae_conv2d_wino_supported(7, 7) == 0 and L.sivae_conv2d_wino_supported(28, 28) == 1
    assert L.sivae_conv2d_wino_up_supported(16, 32) == 1 and L.sivae_conv2d_wino_up_supported(8, 8) == 0
    # (round 4: the forward kernel stores four output pixels per 16-byte store: W % 4 == 0 and a 16-byte aligned output)
    assert L.sivae_conv2d_wino_up_supported(16, 34) == 0 and L.sivae_conv2d_wino_up_supported(16, 36) == 1
    assert L.sivae_conv2d_wino_up_fwd(one, one, ctypes.c_void_p(20), null, null, null, null, 0.2, null, 2, 16, 16, 16, 32,
                                      null) == -2
    # weight-gradient partials are written transformed: 16 (upsample-phase) / 18 (F(4x4,3x3)) values per (co, ci) and slice
    nb = L.sivae_conv2d_wino_up_wgrad_workspace_bytes(16, 64, 64, 64, 64)
    assert nb > 0 and nb % (16 * 64 * 64 * 4) == 0


def test_round2_entry_points_validate_arguments():
    """bf16-mode kernels, the small-batch Linear kernels and the tensor-prior KL: host-side checks only (every call
    returns before any kernel launch), plus the layout / tiling / workspace queries"""
    L = lib.load()
    null = None
    one = ctypes.c_void_p(16)
    # blocked bf16 layout: channel counts are padded to a multiple of 16 (two 8-channel blocks)
    assert [L.sivae_bf16_cblocks(c) for c in (1, 3, 16, 17, 64, 512)] == [2, 2, 2, 4, 8, 64]
    assert L.sivae_bf16_cblocks(0) == -2
    # weight packs: [co_tile][chunk][tap][k-step][half][TCO][8] bf16
    assert L.sivae_bf16_pack_conv_weight_bytes(128, 64, 3, 0) == 1 * 4 * 9 * 1 * 2 * 128 * 16
    assert L.sivae_bf16_pack_conv_weight_bytes(64, 3, 5, 0) == 1 * 1 * 25 * 1 * 2 * 64 * 16
    assert L.sivae_bf16_pack_conv_weight_bytes(64, 3, 5, 1) == 1 * 4 * 25 * 1 * 2 * 32 * 16   # dgrad: 3 outputs -> 32-row tile
    assert L.sivae_bf16_pack_conv_weight_bytes(128, 64, 1, 0) == 1 * 1 * 1 * 4 * 2 * 128 * 16  # 1x1: 64-channel chunks
    assert L.sivae_bf16_pack_conv_weight_bytes(64, 64, 2, 0) == 0
    assert L.sivae_bf16_pack_conv_weight(one, one, 4, 4, 7, 0, null) == -3
    assert L.sivae_bf16_pack_conv_weight(null, one, 4, 4, 3, 0, null) == -1
    a = [null, null, null, null]
    assert L.sivae_bf16_conv2d_fwd(null, one, one, null, *a, 0.2, null, 1, 16, 16, 4, 4, 3, 0, 0, 0, null) == -1
    assert L.sivae_bf16_conv2d_fwd(one, one, one, null, *a, 0.2, null, 1, 16, 16, 4, 4, 2, 0, 0, 0, null) == -3
    assert L.sivae_bf16_conv2d_fwd(one, one, one, null, *a, 0.2, null, 1, 16, 16, 5, 5, 3, 1, 0, 0, null) == -2
    assert L.sivae_bf16_conv2d_fwd(one, one, one, null, one, one, one, one, 0.2, null, 1, 16, 16, 4, 4, 1, 0, 0, 0, null) == -6
    assert L.sivae_bf16_conv2d_fwd(one, one, one, null, *a, 0.2, null, 1, 16, 64, 4, 4, 5, 0, 0, 1, null) == -6  # fp32 out: Co <= 32
    assert L.sivae_bf16_conv2d_wgrad(one, one, one, *a, 0.2, 1, 16, 16, 4, 4, 3, 0, null, 0, null) == -1
    assert L.sivae_bf16_conv2d_wgrad(one, one, one, *a, 0.2, 1, 16, 16, 4, 4, 3, 0, one, 8, null) == -4
    assert L.sivae_bf16_conv2d_wgrad_workspace_bytes(128, 64, 64, 128, 128, 3) >= 64 * 64 * 9 * 4
    assert L.sivae_bf16_conv2d_wgrad_workspace_bytes(128, 64, 64, 128, 128, 4) == 0
    assert L.sivae_bf16_conv2d_num_px_tiles(128, 128, 64, 64, 3) == 128 * 64 * 64 // 256   # big pixel tile (fills the chip)
    assert L.sivae_bf16_conv2d_num_px_tiles(2, 128, 8, 8, 3) == 1                         # small grid: 128-pixel tile
    # the 5x5 stem keeps its 256-pixel tile where the 3x3 kernel of the same width takes 512 (rows of the stats buffer!)
    assert L.sivae_bf16_conv2d_num_px_tiles(128, 64, 128, 128, 5) == 128 * 128 * 128 // 256
    assert L.sivae_bf16_conv2d_num_px_tiles(128, 64, 128, 128, 3) == 128 * 128 * 128 // 512
    assert L.sivae_bf16_conv2d_num_px_tiles(128, 64, 128, 128, 4) == -3
    assert L.sivae_bf16_bn_apply_act(one, null, 0, one, one, one, one, 0.2, null, null, null, 1, 16, 4, 4, null) == -1
    assert L.sivae_bf16_bn_apply_act(one, null, 0, one, one, one, one, 0.2, one, one, null, 1, 16, 3, 4, null) == -2  # pool: even H
    assert L.sivae_bf16_bn_apply_act(one, null, 1, one, one, one, one, 0.2, one, null, null, 1, 16, 4, 4, null) == -1  # res_up w/o res
    assert L.sivae_bf16_bn_signmask_bytes(2, 24, 4, 4) == 2 * 4 * 16 and L.sivae_bf16_bn_signmask_bytes(0, 8, 4, 4) == 0
    # no sign source at all (y, mask and beta NULL) / dz_sum without dz / workspace too small
    assert L.sivae_bf16_bn_bwd(one, 0, null, null, one, one, one, one, null, 0.2, one, null, 0, null, null, 1, 16, 4, 4, one, 1 << 20, null) == -1
    assert L.sivae_bf16_bn_bwd(one, 0, one, null, one, one, one, one, null, 0.2, one, null, 1, null, null, 1, 16, 4, 4, one, 1 << 20, null) == -1
    assert L.sivae_bf16_bn_bwd(one, 0, null, one, one, one, one, one, null, 0.2, one, null, 0, null, null, 1, 16, 4, 4, one, 8, null) == -4
    assert L.sivae_bf16_bn_bwd_workspace_bytes(128, 64, 64, 64) > 0 and L.sivae_bf16_bn_bwd_workspace_bytes(0, 64, 8, 8) == 0
    assert L.sivae_bf16_from_f32_nchw(null, one, 1, 3, 4, 4, 1.0, null) == -1
    assert L.sivae_bf16_to_f32_nchw(one, one, 1, 0, 4, 4, null) == -2
    assert L.sivae_bf16_add_inplace(one, null, 4, null) == -1 and L.sivae_bf16_add_inplace(one, one, 0, null) == 0
    # small-batch Linear: B <= 256, K % 4 == 0, N % 4 == 0
    assert L.sivae_linear_supported(128, 8192, 1024) == 1 and L.sivae_linear_supported(257, 64, 64) == 1 and L.sivae_linear_supported(16385, 64, 64) == 0
    assert L.sivae_linear_supported(8, 66, 64) == 0 and L.sivae_linear_supported(8, 64, 30) == 0
    assert L.sivae_linear_workspace_bytes(128, 8192, 512) >= 128 * 512 * 4
    assert L.sivae_linear_fwd(null, one, null, one, 0, 8, 64, 64, one, 1 << 20, null) == -1
    assert L.sivae_linear_fwd(one, one, null, one, 0, 20000, 64, 64, one, 1 << 20, null) == -2  # (batches above 256 rows run as chunks; 16384 is the cap)
    assert L.sivae_linear_fwd(one, one, null, one, 0, 128, 8192, 512, one, 16, null) == -4
    assert L.sivae_linear_dgrad(one, one, one, 128, 256, 8192, one, 16, null) == -4
    assert L.sivae_linear_wgrad(one, null, one, 8, 64, 64, null) == -1
    # tensor-prior KL: null priors / negative strides
    assert L.sivae_kl_fwd_t(one, one, 8, null, 0, 0, one, 0, 0, one, 2, 8, null) == -1
    assert L.sivae_kl_fwd_t(one, one, 8, one, -1, 0, one, 0, 0, one, 2, 8, null) == -6
    assert L.sivae_kl_bwd_t(one, 0, 1.0, one, one, 4, one, 0, 0, one, 0, 0, one, one, 8, 2, 8, null) == -2  # ld < Z


def test_round3_entry_points_validate_arguments():
    """Winograd F(4x4,3x3) forward / weight gradient, its image-pair and split-K forms, and the segmented-batch entry
    points: host-side validation and the shape / plan queries (every call returns before any kernel launch)"""
    L = lib.load()
    null = None
    one = ctypes.c_void_p(16)
    # domains: 1 = whole 32 x 16 pixel blocks, 2 / 3 / 4 = 16 x 16 / 8 x 8 / 4 x 4 maps as grids of whole images per work
    # item (2 / 8 / 32 of them), 0 = unsupported
    maps = ((256, 256), (32, 32), (16, 32), (16, 16), (8, 8), (4, 4), (24, 32), (8, 16), (2, 2))
    assert [L.sivae_conv2d_wino4_supported(h, w) for h, w in maps] == [1, 1, 1, 2, 3, 4, 0, 0, 0]
    assert [L.sivae_conv2d_wino4_images_per_item(h, w) for h, w in maps] == [1, 1, 1, 2, 8, 32, 0, 0, 0]
    assert L.sivae_conv2d_wino4_num_px_tiles(16, 8, 8) == 2 and L.sivae_conv2d_wino4_num_px_tiles(64, 4, 4) == 2
    assert L.sivae_conv2d_wino4_num_px_tiles(12, 8, 8) == -2 and L.sivae_conv2d_wino4_num_px_tiles(48, 4, 4) == -2
    assert L.sivae_conv2d_wino4_pays(12, 64, 64, 8, 8) == 0 and L.sivae_conv2d_wino4_pays(256, 512, 512, 8, 8) == 1
    # the pooled data gradient of an upsample-conv: whole-tile maps, <= 64 input channels, an item for every CU
    pp = L.sivae_conv2d_wino4_dgrad_pool_pays
    assert pp(256, 64, 64, 256, 256) == 1 and pp(128, 64, 64, 256, 256) == 1 and pp(256, 64, 128, 256, 256) == 0
    assert pp(256, 64, 64, 16, 16) == 0 and pp(1, 64, 64, 32, 32) == 0
    assert L.sivae_conv2d_wino4_dgrad_pool(one, one, one, 2, 64, 64, 16, 16, 0, null) == -2  # (16 x 16: not a whole-tile map)
    assert L.sivae_conv2d_wino4_dgrad_pool(one, one, ctypes.c_void_p(20), 2, 64, 64, 32, 32, 0, null) == -2  # (8-byte stores)
    assert L.sivae_conv2d_wino4_dgrad_pool(null, one, one, 2, 64, 64, 32, 32, 0, null) == -1
    # the launch-level policy of the image-grid form against F(2x2,3x3), pinned to the measured table
    # (profiles/r6_wino4_small_maps_vs_f23.txt: wins need (slices x items) for every CU AND a long K slice)
    fwd = {(256, 512, 512, 8, 8): 1, (128, 512, 512, 8, 8): 1, (64, 512, 512, 8, 8): 0, (32, 512, 512, 8, 8): 0,
           (512, 256, 256, 8, 8): 1, (256, 256, 256, 8, 8): 0, (256, 128, 256, 8, 8): 0, (512, 128, 256, 8, 8): 1,
           (128, 512, 512, 4, 4): 1, (64, 512, 512, 4, 4): 0, (256, 512, 512, 4, 4): 1, (256, 256, 256, 4, 4): 0,
           (256, 512, 512, 16, 16): 0, (256, 512, 512, 32, 32): 0}  # (modes 1 / 2 are not this query's business)
    assert {k: L.sivae_conv2d_wino4_small_pays(*k) for k in fwd} == fwd
    assert L.sivae_conv2d_wino4_num_px_tiles(4, 32, 64) == 4 * 2 * 2 and L.sivae_conv2d_wino4_num_px_tiles(6, 16, 16) == 3
    assert L.sivae_conv2d_wino4_num_px_tiles(5, 16, 16) == -2  # (an odd batch has no image pairs)
    assert L.sivae_conv2d_wino4_pays(5, 64, 64, 16, 16) == 0 and L.sivae_conv2d_wino4_pays(2, 8, 64, 256, 256) == 0
    # packed U: [6][Ci_pad32][Co_pad64][6] floats
    assert L.sivae_pack_wino4_weight_bytes(64, 40, 0) == 36 * 64 * 64 * 4 and L.sivae_pack_wino4_weight_bytes(64, 40, 1) == 36 * 64 * 64 * 4
    assert L.sivae_pack_wino4_weight_bytes(64, 40, 2) == 0
    assert L.sivae_pack_wino4_weight(null, one, 4, 4, 0, null) == -1 and L.sivae_pack_wino4_weight(one, one, 4, 4, 3, null) == -6
    # forward: null / unsupported map / odd pair batch / pairs straddling a segment / prologue slope / alignment of y
    assert L.sivae_conv2d_wino4_fwd(null, one, one, null, 2, 32, 32, 32, 32, 0, null) == -1
    assert L.sivae_conv2d_wino4_fwd(one, one, one, null, 2, 32, 32, 8, 8, 0, null) == -2
    assert L.sivae_conv2d_wino4_fwd(one, one, one, null, 3, 32, 32, 16, 16, 0, null) == -2
    assert L.sivae_conv2d_wino4_fwd_pro(one, one, one, one, one, one, one, 0.2, null, 6, 32, 32, 16, 16, 0, 3, null) == -2
    assert L.sivae_conv2d_wino4_fwd_pro(one, one, one, one, one, one, one, 1.5, null, 2, 32, 32, 32, 32, 0, 0, null) == -6
    assert L.sivae_conv2d_wino4_fwd_pro(one, one, one, one, null, one, one, 0.2, null, 2, 32, 32, 32, 32, 0, 0, null) == -1
    assert L.sivae_conv2d_wino4_fwd(one, one, ctypes.c_void_p(20), null, 2, 32, 32, 32, 32, 0, null) == -2
    # split-K plan: only below one work item per CU, powers of two, >= 8 chunks of 8 channels per slice
    assert L.sivae_conv2d_wino4_splitk(128, 512, 512, 32, 32) == 1
    S = L.sivae_conv2d_wino4_splitk(16, 512, 512, 16, 16)
    assert S in (1, 2, 4, 8) and L.sivae_conv2d_wino4_splitk(16, 64, 512, 16, 16) == 1  # (64 channels: one slice)
    assert L.sivae_conv2d_wino4_splitk_workspace_bytes(16, 512, 512, 16, 16) == (S * 16 * 512 * 256 * 4 if S > 1 else 0)
    assert L.sivae_conv2d_wino4_splitk(3, 512, 512, 16, 16) == -2
    # every plan the query returns is one the kernel accepts: slices of an EVEN number (>= 8) of 8-channel chunks — the
    # padded widths 288 / 352 / 416 / 480 once gave S = 4 with 9 / 11 / 13 / 15 chunks per slice (round-3 advisor finding)
    for Ci in list(range(272, 289)) + [320, 352, 384, 416, 448, 480, 512]:
        for B, H, W in ((2, 32, 32), (4, 32, 32), (2, 16, 16), (8, 16, 16), (16, 16, 16), (1, 32, 64)):
            for Co in (64, 128, 512):
                Sx = L.sivae_conv2d_wino4_splitk(B, Ci, Co, H, W)
                assert Sx in (1, 2, 4, 8), (B, Ci, Co, H, W, Sx)
                if Sx > 1:
                    chunks = ((Ci + 31) // 32) * 32 // 8
                    assert chunks % Sx == 0 and (chunks // Sx) % 2 == 0 and chunks // Sx >= 8, (Ci, Sx, chunks)
    assert L.sivae_conv2d_wino4_splitk(2, 288, 64, 32, 32) in (1, 2)  # (the advisor's example: not 4)
    if S > 1:
        assert L.sivae_conv2d_wino4_fwd_splitk(one, one, one, null, null, null, null, 1.0, null, 16, 512, 512, 16, 16, 0,
                                               0, null, 0, null) == -1
        assert L.sivae_conv2d_wino4_fwd_splitk(one, one, one, null, null, null, null, 1.0, null, 16, 512, 512, 16, 16, 0,
                                               0, one, 64, null) == -4
    # weight gradient: strips of 4 x 16 pixels; at most two segments; 16-byte aligned operands
    # (round 6: the 8 x 8 / 4 x 4 maps run as strips of 2 / 4 whole images)
    assert [L.sivae_conv2d_wino4_wgrad_supported(h, w) for h, w in ((16, 16), (4, 16), (8, 8), (4, 4), (6, 16), (256, 256), (8, 4))] \
        == [1, 1, 1, 1, 0, 1, 0]
    assert [L.sivae_conv2d_wino4_wgrad_images_per_stage(h, w) for h, w in ((16, 16), (8, 8), (4, 4), (6, 16))] == [1, 2, 4, 0]
    assert L.sivae_conv2d_wino4_wgrad_pays(256, 512, 512, 8, 8) == 1 and L.sivae_conv2d_wino4_wgrad_pays(255, 512, 512, 8, 8) == 0
    # small maps below "a block of 24 stages per CU": the cost model against the direct kernel (same measured table)
    wg = {(32, 512, 512, 8, 8): 1, (128, 256, 256, 8, 8): 1, (64, 256, 256, 8, 8): 0, (256, 128, 256, 8, 8): 1,
          (64, 128, 128, 8, 8): 0, (32, 512, 512, 4, 4): 1, (32, 256, 256, 4, 4): 1, (256, 256, 256, 4, 4): 0,
          (256, 128, 128, 4, 4): 0}
    assert {k: L.sivae_conv2d_wino4_wgrad_pays(*k) for k in wg} == wg
    assert L.sivae_conv2d_wino4_wgrad_pays(128, 128, 128, 128, 128) == 1 and L.sivae_conv2d_wino4_wgrad_pays(1, 64, 64, 16, 16) == 0
    nb = L.sivae_conv2d_wino4_wgrad_workspace_bytes(128, 128, 128, 128, 128)
    assert nb > 0 and nb % (18 * 128 * 128 * 4) == 0  # (row-transformed partials: 3 x 6 values per (co, ci))
    assert L.sivae_conv2d_wino4_wgrad(one, one, one, null, null, null, null, 0.2, 3, 32, 32, 8, 8, 0, one, 1 << 30, null) == -2
    assert L.sivae_conv2d_wino4_wgrad(one, one, one, null, null, null, null, 0.2, 2, 32, 32, 8, 4, 0, one, 1 << 30, null) == -2
    assert L.sivae_conv2d_wino4_wgrad(one, null, one, null, null, null, null, 0.2, 2, 32, 32, 16, 16, 0, one, 1 << 30, null) == -1
    assert L.sivae_conv2d_wino4_wgrad(one, one, one, null, null, null, null, 0.2, 2, 32, 32, 16, 16, 0, one, 16, null) == -4
    assert L.sivae_conv2d_wino4_wgrad(one, one, one, one, one, one, one, 0.2, 6, 32, 32, 16, 16, 2, one, 1 << 30, null) == -2
    assert L.sivae_conv2d_wino4_wgrad(ctypes.c_void_p(20), one, one, null, null, null, null, 0.2, 2, 32, 32, 16, 16, 0, one,
                                      1 << 30, null) == -2
    # segmented batches: the images of a segment must divide the batch
    assert L.sivae_conv2d_wino_fwd_seg(one, one, one, one, one, one, one, 0.2, null, 6, 16, 16, 16, 16, 0, 0, 4,
                                       null) == -2


def test_workspace_and_padding_queries():
    L = lib.load()
    assert L.sivae_conv_ck(3) == 8 and L.sivae_conv_ck(1) == 32 and L.sivae_conv_ck(5) == 4
    assert L.sivae_conv_ci_pad(5, 3) == 4 and L.sivae_conv_co_pad(3) == 128 and L.sivae_conv_co_pad(129) == 256
    assert L.sivae_pack_conv_weight_bytes(64, 3, 5, 0) == 25 * 4 * 128 * 4
    assert L.sivae_pack_conv_weight_bytes(64, 3, 5, 1) == 25 * 64 * 128 * 4
    assert L.sivae_conv2d_wgrad_workspace_bytes(128, 64, 64, 256, 256, 3) >= 64 * 64 * 9 * 4
    assert L.sivae_conv2d_wgrad_workspace_bytes(0, 64, 64, 8, 8, 3) == 0
    assert L.sivae_bn_workspace_bytes(128, 64, 65536) > 0
    assert L.sivae_conv2d_fwd_num_px_tiles(128, 64, 256, 256) == 128 * 256  # 256-pixel tiles for Co <= 64
    assert L.sivae_conv2d_fwd_num_px_tiles(128, 512, 4, 4) == 16            # 8 images per 128-pixel tile


def test_cpu_tensors_are_rejected_loudly():
    import train_soft_intro_vae as T
    from sivae_hip import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.bn_stats(torch.zeros(2, 4, 4, 4))
    with pytest.raises(RuntimeError, match="no CPU"):
        T.train_soft_intro_vae(dataset="synthetic-cifar10", device=torch.device("cpu"), num_epochs=1)
    with pytest.raises(NotImplementedError):
        T.train_soft_intro_vae(dataset="imagenet", device=torch.device("cpu"))
    with pytest.raises(NotImplementedError):
        T.calc_reconstruction_loss(torch.zeros(2, 3), torch.zeros(2, 3), loss_type="mse", reduction="avg")
    with pytest.raises(NotImplementedError):
        T.calc_reconstruction_loss(torch.zeros(2, 3), torch.zeros(2, 3), loss_type="huber", reduction="sum")


def test_python_surface_matches_reference_signatures():
    """names + parameter lists of the reference entry points (train_soft_intro_vae.py:337-341,
    bootstrap :360-364, 2d :486-489, SoftIntroVAE/Encoder/Decoder ctor args)"""
    import train_soft_intro_vae as T
    import train_soft_intro_vae_2d as T2
    import train_soft_intro_vae_bootstrap as TB
    want = ["dataset", "z_dim", "lr_e", "lr_d", "batch_size", "num_workers", "start_epoch", "exit_on_negative_diff",
            "num_epochs", "num_vae", "save_interval", "recon_loss_type", "beta_kl", "beta_rec", "beta_neg",
            "test_iter", "seed", "pretrained", "device", "num_row", "gamma_r", "with_fid"]
    assert list(inspect.signature(T.train_soft_intro_vae).parameters) == want
    wb = want[:8] + ["copy_to_target_freq"] + want[8:]
    assert list(inspect.signature(TB.train_soft_intro_vae).parameters) == wb
    assert inspect.signature(TB.train_soft_intro_vae).parameters["gamma_r"].default == 1.0
    assert list(inspect.signature(T2.train_soft_intro_vae_toy).parameters) == [
        "z_dim", "lr_e", "lr_d", "batch_size", "n_iter", "num_vae", "save_interval", "recon_loss_type", "beta_kl",
        "beta_rec", "beta_neg", "test_iter", "seed", "pretrained", "scale", "device", "dataset", "gamma_r"]
    for mod in (T, TB):
        for name in ("ResidualBlock", "Encoder", "Decoder", "SoftIntroVAE", "calc_kl", "reparameterize",
                     "calc_reconstruction_loss", "load_model", "save_checkpoint", "str_to_list", "is_image_file"):
            assert hasattr(mod, name), name
    assert list(inspect.signature(T.SoftIntroVAE.__init__).parameters)[1:] == [
        "cdim", "zdim", "channels", "image_size", "conditional", "cond_dim"]
    assert hasattr(TB.SoftIntroVAE, "decode_target")


GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name,seed,mod", [("step_cifar_narrow", 0, "train_soft_intro_vae"),
                                           ("step_deep64_narrow", 1, "train_soft_intro_vae"),
                                           ("step_mnist_narrow", 2, "train_soft_intro_vae"),
                                           ("step_bootstrap_narrow", 3, "train_soft_intro_vae_bootstrap")])
def test_constructor_reproduces_reference_state_dict(name, seed, mod):
    """same seed -> bit-identical initial state_dict as the reference (same key order, same init draws, and the
    encoder BatchNorm buffers mutated by the reference constructor's dummy forward)"""
    import importlib
    M = importlib.import_module(mod)
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    torch.manual_seed(seed)
    model = M.SoftIntroVAE(cdim=int(fx["meta_cdim"]), zdim=int(fx["meta_zdim"]),
                           channels=[int(c) for c in fx["meta_channels"]], image_size=int(fx["meta_image_size"]))
    sd = model.state_dict()
    ref = {k[len("init/"):]: fx[k] for k in fx.files if k.startswith("init/")}
    assert list(sd.keys()) == list(ref.keys())
    for k, v in ref.items():
        assert np.array_equal(sd[k].numpy(), v), k
    assert model.zdim == int(fx["meta_zdim"]) and callable(model.sample)  # what metrics/fid_score.py:246-247 needs


def test_2d_model_state_dict_keys():
    import train_soft_intro_vae_2d as T2
    fx = np.load(os.path.join(GOLD, "loop_2d.npz"))
    m = T2.SoftIntroVAESimple(x_dim=2, zdim=2, n_layers=3, num_hidden=int(fx["meta_num_hidden"]))
    ref = [k[len("init/"):] for k in fx.files if k.startswith("init/")]
    assert list(m.state_dict().keys()) == ref
    for k in ref:
        assert tuple(m.state_dict()[k].shape) == fx["init/" + k].shape


def test_philox_offsets_and_multistep_lr():
    from sivae_hip.optim import MultiStepLR
    from sivae_hip.rng import PhiloxStream

    class Opt:
        param_groups = [{"lr": 2e-4}]
    o = Opt()
    s = MultiStepLR(o, milestones=(2, 4), gamma=0.1)
    lrs = []
    for _ in range(5):
        s.step()
        lrs.append(o.param_groups[0]["lr"])
    assert np.allclose(lrs, [2e-4, 2e-5, 2e-5, 2e-6, 2e-6])
    a, b = PhiloxStream(5, 0), PhiloxStream(5, 1)
    assert a.seed != b.seed and a.offset == 0


def test_round4_entry_points_validate_arguments():
    """the one-pass BatchNorm backward (fp32 + bf16), the two-stage statistics finalize and the batched weight packing:
    plan / support queries and host-side validation (every call returns before any kernel launch)"""
    L = lib.load()
    null = None
    one = ctypes.c_void_p(16)
    # one-pass BatchNorm backward: power-of-two maps, plane sets within 32-bit byte offsets, <= 8192 channels
    assert [L.sivae_bn_bwd_fused_supported(*a) for a in ((128, 64, 256, 256, 128), (256, 64, 256, 256, 128), (8, 512, 4, 4, 8),
                                                        (2, 8, 28, 28, 2), (4, 16, 8, 12, 4), (4, 9000, 8, 8, 4),
                                                        (6, 16, 8, 8, 4), (2, 16, 2, 4, 2))] == [1, 1, 1, 0, 0, 0, 0, 1]
    assert L.sivae_bn_bwd_fused_supported(512, 64, 256, 256, 512) == 0  # (a plane set's window must stay below 4 GB)
    assert L.sivae_bn_bwd_fused_state_uints() >= 2 * 17 * 32 + 8192
    assert L.sivae_bn_bwd_fused_workspace_bytes(8, 512, 4, 4, 8) == (512 * 1 * 2 + 512 * 2) * 8   # one slab per channel
    assert L.sivae_bn_bwd_fused_workspace_bytes(2, 8, 28, 28, 2) == 0
    args = (one, null, null, one, one, one, one, one)
    assert L.sivae_bn_bwd_fused(*args, 2, 0.2, one, null, null, null, 8, 16, 8, 8, 0, 0, 8, null, one, 1 << 20, null) == -1  # no state
    assert L.sivae_bn_bwd_fused(*args, 2, 0.2, one, null, null, null, 8, 16, 8, 12, 0, 0, 8, one, one, 1 << 20, null) == -2
    assert L.sivae_bn_bwd_fused(*args, 2, 0.2, one, null, null, null, 8, 16, 8, 8, 0, 0, 8, one, one, 8, null) == -4
    assert L.sivae_bn_bwd_fused(*args, 7, 0.2, one, null, null, null, 8, 16, 8, 8, 0, 0, 8, one, one, 1 << 20, null) == -6
    assert L.sivae_bn_bwd_fused(*args, 3, 0.2, one, null, null, null, 8, 16, 8, 8, 0, 0, 8, one, one, 1 << 20, null) == -1  # no mask
    assert L.sivae_bn_bwd_fused(*args, 1, 0.2, one, one, null, null, 8, 16, 8, 8, 1, 1, 8, one, one, 1 << 20, null) == -1  # act 1: no y
    assert L.sivae_bn_bwd_fused(one, one, null, one, one, one, one, one, 1, 0.2, one, one, null, null, 8, 16, 8, 8, 1, 1, 8,
                                one, one, 1 << 20, null) == -6  # pooled dy and block-summed dz do not combine
    assert L.sivae_bn_bwd_fused(one, null, null, one, one, one, one, one, 2, 0.2, ctypes.c_void_p(20), null, null, null, 8, 16,
                                8, 8, 0, 0, 8, one, one, 1 << 20, null) == -2  # 16-byte aligned tensors
    # bf16 form
    assert [L.sivae_bf16_bn_bwd_fused_supported(*a) for a in ((128, 64, 128, 128), (8, 512, 4, 4), (2, 24, 8, 12),
                                                             (128, 64, 256, 256))] == [1, 1, 0, 0]
    assert L.sivae_bf16_bn_bwd_fused_workspace_bytes(8, 512, 4, 4) == (64 * 1 * 16 + 64 * 16) * 8 + 16  # partials + sums
    # segmented form: B = nseg * seg_images, one plane set per (segment, channel block); a 128-image pass at 256 x 256 does
    # not fit the grid, its two 64-image halves do not make it fit either (the plane set is per segment)
    assert [L.sivae_bf16_bn_bwd_fused_seg_supported(*a) for a in ((256, 64, 128, 128, 128), (16, 512, 4, 4, 8),
                                                                 (16, 512, 4, 4, 5), (16, 512, 4, 4, 0),
                                                                 (256, 64, 256, 256, 128))] == [1, 1, 0, 0, 0]
    assert L.sivae_bf16_bn_bwd_fused_seg_workspace_bytes(16, 512, 4, 4, 8) == (128 * 1 * 16 + 128 * 16) * 8 + 16
    assert L.sivae_bf16_bn_bwd_fused_seg(one, 0, null, null, one, one, one, one, one, 0.2, one, null, 0, null, null, 16, 16, 8,
                                         8, 5, one, one, 1 << 20, null) == -2  # 16 images are not passes of 5
    assert L.sivae_bf16_bn_apply_act_seg(one, null, 0, one, one, one, one, 0.2, one, null, null, 16, 16, 8, 8, 5, null) == -2
    assert L.sivae_bf16_bn_apply_act_seg(one, null, 0, one, one, one, one, 0.2, null, null, null, 16, 16, 8, 8, 8, null) == -1
    assert L.sivae_bf16_bn_bwd_fused(one, 0, null, null, one, one, one, one, one, 0.2, one, null, 0, null, null, 8, 16, 8, 8,
                                     null, one, 1 << 20, null) == -1
    assert L.sivae_bf16_bn_bwd_fused(one, 0, null, null, one, one, one, one, null, 0.2, one, null, 0, null, null, 8, 16, 8, 8,
                                     one, one, 1 << 20, null) == -1  # no sign source at all
    assert L.sivae_bf16_bn_bwd_fused(one, 0, null, null, one, one, one, one, one, 0.2, one, null, 0, null, null, 8, 16, 8, 12,
                                     one, one, 1 << 20, null) == -2
    assert L.sivae_bf16_bn_bwd_fused(one, 0, null, null, one, one, one, one, one, 0.2, one, null, 0, null, null, 8, 16, 8, 8,
                                     one, one, 8, null) == -4
    # two-stage statistics finalize: a workspace only from 2048 partial rows per pass on
    assert L.sivae_bn_stats_from_conv_workspace_bytes(1024, 1, 64) == 0
    assert L.sivae_bn_stats_from_conv_workspace_bytes(4096, 1, 64) == 1 * 64 * 32 * 2 * 8
    assert L.sivae_bn_stats_from_conv_workspace_bytes(2 * 3000, 2, 40) == 2 * 40 * 24 * 2 * 8
    assert L.sivae_bn_stats_from_conv_workspace_bytes(4097, 2, 64) == 0  # (rows must split evenly over the passes)
    assert L.sivae_bn_stats_from_conv_ws(one, 4096, 1, 0, 4, 64, 64, 1e-5, 0.1, null, null, null, one, one, null, 0, null) == -4
    assert L.sivae_bn_stats_from_conv_ws(null, 4096, 1, 0, 4, 64, 64, 1e-5, 0.1, null, null, null, one, one, one, 1 << 20,
                                         null) == -1
    assert L.sivae_bn_stats_from_conv_ws(one, 4096, 1, 0, 4, 64, 64, 1e-5, 0.1, one, null, null, one, one, one, 1 << 20,
                                         null) == -1  # running mean without running var
    # batched weight packing: the job table is filled on the host
    jb = L.sivae_pack_job_bytes()
    assert jb >= 64 and jb % 8 == 0
    host = ctypes.create_string_buffer(jb * 4)
    nb = L.sivae_pack_job_fill(host, 0, 2, one, one, 512, 512, 3, 0, 0)       # F(4x4,3x3): 512 x 512 weight pairs
    assert nb == 512 * 512 // 512
    nb2 = L.sivae_pack_job_fill(host, 1, 0, one, one, 64, 3, 5, 1, nb)        # direct 5x5, data-gradient mode
    assert 1 <= nb2 <= 1024
    assert L.sivae_pack_job_fill(host, 2, 1, one, one, 64, 64, 3, 0, nb + nb2) >= 1
    assert L.sivae_pack_job_fill(host, 3, 3, one, one, 64, 64, 5, 0, 0) == -3  # the Winograd forms are 3x3 only
    assert L.sivae_pack_job_fill(host, 3, 9, one, one, 64, 64, 3, 0, 0) == -6
    # form 6: the bf16 operand slabs of the bf16 mode (one launch per optimizer step); ks may be the 5 x 1 code 51
    assert L.sivae_pack_job_fill(host, 3, 6, one, one, 64, 64, 3, 1, 0) >= 1
    assert L.sivae_pack_job_fill(host, 3, 6, one, one, 64, 15, 51, 0, 0) >= 1
    assert L.sivae_pack_job_fill(host, 3, 6, one, one, 64, 64, 4, 0, 0) == -3
    assert L.sivae_pack_job_fill(host, 3, 0, one, one, 64, 64, 4, 0, 0) == -3
    assert L.sivae_pack_job_fill(null, 0, 0, one, one, 64, 64, 3, 0, 0) == -1
    assert L.sivae_pack_job_fill(host, 0, 0, one, one, 0, 64, 3, 0, 0) == -2
    assert L.sivae_pack_batch(0, null, one, 4, null) == -1 and L.sivae_pack_batch(0, one, one, 0, null) == -2
    assert L.sivae_pack_batch(7, one, one, 4, null) == -6


def test_lib_call_raises_on_error_codes():
    """sivae_hip.lib.call (what every wrapper in ops.py goes through) turns a non-zero status into SivaeError naming the
    entry point and the code; sha256() stamps the kernel sources"""
    import pytest
    with pytest.raises(lib.SivaeError) as ei:
        lib.call("sivae_pack_batch", 7, ctypes.c_void_p(16), ctypes.c_void_p(16), 4, None)
    assert ei.value.code == -6 and "sivae_pack_batch" in str(ei.value) and "SIVAE_ERR_MODE" in str(ei.value)
    assert len(lib.sha256()) == 64 and lib.sha256() == lib.sha256()


def test_round5_entry_points_validate_arguments():
    """round 5: the six-product F(4x4,3x3) kernel's pack / launch entry points, the barrier-timeout word and the test-support
    kernels — size queries and host-side validation (every call returns before any kernel launch)"""
    L = lib.load()
    null = None
    one = ctypes.c_void_p(16)
    # pre-split operand: 36 frequencies x padded channels x 3 pieces x 2 bytes = 1.5x the fp32 F(4x4,3x3) pack
    for (Co, Ci, mode) in ((64, 64, 0), (72, 100, 0), (40, 130, 1), (512, 512, 1)):
        assert L.sivae_pack_wino4_b6_weight_bytes(Co, Ci, mode) * 2 == L.sivae_pack_wino4_weight_bytes(Co, Ci, mode) * 3
    assert L.sivae_pack_wino4_b6_weight_bytes(0, 64, 0) == 0 and L.sivae_pack_wino4_b6_weight_bytes(64, 64, 2) == 0
    assert L.sivae_pack_wino4_b6_weight(null, one, 64, 64, 0, null) == -1
    assert L.sivae_pack_wino4_b6_weight(one, one, 64, 0, 0, null) == -2
    assert L.sivae_pack_wino4_b6_weight(one, one, 64, 64, 3, null) == -6
    assert L.sivae_pack_wino4_b6_weight(one, ctypes.c_void_p(20), 64, 64, 0, null) == -2  # 16-byte aligned operand
    f = L.sivae_conv2d_wino4_b6_fwd
    assert f(null, one, one, null, null, null, null, 1.0, null, 2, 64, 64, 32, 32, 0, 0, null) == -1
    assert f(one, one, one, null, null, null, null, 1.0, null, 2, 64, 64, 30, 32, 0, 0, null) == -2   # map not in 16 x 32 blocks
    assert f(one, one, one, null, null, null, null, 1.0, null, 3, 64, 64, 16, 16, 0, 0, null) == -2   # 16 x 16 maps: image pairs
    assert f(one, one, ctypes.c_void_p(20), null, null, null, null, 1.0, null, 2, 64, 64, 32, 32, 0, 0, null) == -2
    assert f(one, one, one, one, null, null, null, 0.2, null, 2, 64, 64, 32, 32, 0, 0, null) == -1    # prologue: all four arrays
    assert f(one, one, one, one, one, one, one, 1.5, null, 2, 64, 64, 32, 32, 0, 0, null) == -6       # slope in [0, 1]
    assert f(one, one, one, null, null, null, null, 1.0, null, 4, 64, 64, 32, 32, 0, 3, null) == -2   # segments divide B
    g = L.sivae_conv2d_wino4_b6_fwd_splitk
    assert L.sivae_conv2d_wino4_splitk(2, 256, 64, 32, 32) > 1
    assert g(one, one, one, null, null, null, null, 1.0, null, 2, 256, 64, 32, 32, 0, 0, null, 0, null) == -1   # no workspace
    assert g(one, one, one, null, null, null, null, 1.0, null, 2, 256, 64, 32, 32, 0, 0, one, 16, null) == -4   # too small
    # pack-batch form 5 (the pre-split operand) fills a job like form 2
    jb = L.sivae_pack_job_bytes()
    host = ctypes.create_string_buffer(jb)
    assert L.sivae_pack_job_fill(host, 0, 5, one, one, 64, 64, 3, 0, 0) > 0
    assert L.sivae_pack_job_fill(host, 0, 5, one, one, 64, 64, 1, 0, 0) == -3   # 3x3 only
    assert L.sivae_pack_job_fill(host, 0, 7, one, one, 64, 64, 3, 0, 0) == -6   # unknown form (6: the bf16 slabs)
    # the grid-barrier timeout word lies inside the barrier state, on its own 128-byte line behind the 17 used lines
    w = L.sivae_bn_bwd_fused_poison_word()
    assert 17 * 32 <= w < 1024 and w % 32 == 0 and w < L.sivae_bn_bwd_fused_state_uints()
    # a (256, 64, 256, 256) batch of 129..160 images per segment fits the 10-quad plan only: the query now plans with the
    # tightest variant's budget (8 quads), so the callers fall back instead of hitting SIVAE_ERR_SHAPE in the launch
    assert L.sivae_bn_bwd_fused_supported(160, 64, 256, 256, 160) == 0
    assert L.sivae_bn_bwd_fused_supported(128, 64, 256, 256, 128) == 1
    # loss assembly: 1..6 terms, every used pointer present
    lc = L.sivae_lincomb
    assert lc(one, one, null, null, null, null, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 3, one, null) == -1
    assert lc(one, one, null, null, null, null, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 2, null, null) == -1
    assert lc(one, one, one, one, one, one, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 7, one, null) == -2
    assert lc(one, null, null, null, null, null, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0, one, null) == -2
    assert L.sivae_lincomb_bwd(null, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 2, one, null) == -1
    assert L.sivae_lincomb_bwd(one, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 9, one, null) == -2
    # the streaming 1x1 kernel takes up to 512 input channels (32-channel tile above 256)
    assert L.sivae_conv1x1_stream_supported(2, 512, 256, 1024) == 1 and L.sivae_conv1x1_stream_supported(2, 514, 256, 1024) == 0
    # test hooks are not part of the product ABI (the squatter kernel lives in tests/support/, the barrier-timeout test
    # corrupts the caller-owned state instead of calling a hook)
    assert not [n for n in lib.prototypes() if "debug" in n]
    assert not hasattr(L, "sivae_debug_squatter") and not hasattr(L, "sivae_debug_bn_fused_break_next")
    import support
    sq = support.load().testsupport_squatter
    assert sq(0, 256, 1024, 0, 1000, null, null) == -2 and sq(4, 2048, 1024, 0, 1000, null, null) == -2
    assert sq(4, 256, 1 << 20, 0, 1000, null, null) == -2 and sq(4, 256, 1024, 0, 0, null, null) == -2


def test_bn_fused_persistent_form_is_refused_under_a_cu_mask():
    """plans of the one-pass BatchNorm backward refuse the grid-barrier form when the process runs under a CU mask (the
    device still reports every CU: the grid could never become resident) or with SIVAE_BN_FUSED_PERSISTENT=0; the
    barrier-free form for plane sets of one block stays available"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from sivae_hip import lib; L = lib.load(); "
            "print(L.sivae_bn_bwd_fused_supported(256, 64, 256, 256, 128), L.sivae_bn_bwd_fused_supported(8, 512, 4, 4, 8), "
            "L.sivae_bf16_bn_bwd_fused_supported(128, 64, 128, 128), L.sivae_bf16_bn_bwd_fused_supported(8, 512, 4, 4))"
            % os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "soft-intro-vae-pytorch_amd"))
    for env_add, want in (({}, "1 1 1 1"), ({"HSA_CU_MASK": "0:0-31"}, "0 1 0 1"), ({"SIVAE_BN_FUSED_PERSISTENT": "0"}, "0 1 0 1")):
        env = dict(os.environ)
        for k in ("HSA_CU_MASK", "ROC_GLOBAL_CU_MASK", "SIVAE_BN_FUSED_PERSISTENT"):
            env.pop(k, None)
        env.update(env_add)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-400:]
        assert out.stdout.strip().splitlines()[-1] == want, (env_add, out.stdout)
