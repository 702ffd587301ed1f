"""bf16-mode kernels (sivae_bf16_*) against stock torch CPU fp64 on the same bf16-rounded operands (pytest -m gpu)."""
import os
import subprocess
import sys

import pytest
import torch

import kernel_checks16 as kc

CHECKS = kc.all_checks()


@pytest.mark.gpu
@pytest.mark.parametrize("label,thunk", CHECKS, ids=[c[0] for c in CHECKS])
def test_kernel16(label, thunk):
    results = thunk()
    torch.cuda.synchronize()
    bad = [(n, e, t) for (n, e, t) in results if not e <= t]
    assert not bad, "bf16 kernel parity failures: %s" % bad


@pytest.mark.gpu
def test_conv16_staged_form_behind_its_switch():
    """SIVAE_BF16_CONV_AD=0 (the A/B switch of bf16_conv.hip: weight slabs staged through LDS again) is read once per
    process: the 3x3 forward / data-gradient / split-K / fused / big-tile checks in a child process with the switch off"""
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, SIVAE_BF16_CONV_AD="0")
    r = subprocess.run([sys.executable, os.path.join(here, "kernel_checks16.py"), "conv16", "dgrad16", "splitk16"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "kernel_checks16: 0 failures" in r.stdout
