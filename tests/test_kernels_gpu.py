"""T2 — every HIP kernel against stock torch CPU fp64 (runs on the MI355X box: pytest -m gpu)."""
import pytest
import torch

import kernel_checks as kc

CHECKS = kc.all_checks()


@pytest.mark.gpu
@pytest.mark.parametrize("label,thunk", CHECKS, ids=[c[0] for c in CHECKS])
def test_kernel(label, thunk):
    results = thunk()
    torch.cuda.synchronize()
    bad = [(n, e, t) for (n, e, t) in results if not e <= t]
    assert not bad, "kernel parity failures: %s" % bad
