"""A bounded, fixed-seed slice of the randomised GPU-vs-oracle campaign (tools/fuzz_parity.py) inside the suite the driver
runs: deterministic case list (case count, not wall time, ends it), every model / rule / source class hit at least once."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fixed_seed_fuzz_slice():
    import torch
    assert torch.cuda.is_available()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity as F
    # window lengths up to 2^22 and at most 2^21 coefficients per case keep the CPU oracle's share to ~20 s on 16 host cores
    cases, samples, by = F.fuzz(budget=None, seed=20261004, max_cases=160, max_count=1 << 21, max_pw=22, verbose=False)
    assert cases == 160 and samples > 10_000_000
    missing = [c for c in F.CLASSES if by.get(c, 0) == 0]
    assert not missing, (missing, by)


def test_fixed_seed_tile_fuzz_slice():
    """The same for the tile path alone (tools/fuzz_tile.py): whole periods at 2^22 .. 2^24 through every table format (plain,
    delta16, residual, nibble), all models and both cosine-sum rules, built-in and random weights, ownership parts."""
    import torch
    assert torch.cuda.is_available()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_tile as F
    cases, samples, plans = F.fuzz(budget=None, seed=20261005, max_cases=60, pws=(22, 22, 23, 23, 24))
    assert cases == 60 and samples > 60 * (1 << 22)
    for want in ("table[plain]", "table[delta16]", "table[residual]", "table[nibble]"):
        assert plans.get(want, 0) > 0, plans
