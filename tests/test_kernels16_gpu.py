"""bf16-mode kernels (sivae_bf16_*) against stock torch CPU fp64 on the same bf16-rounded operands (pytest -m gpu)."""
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
