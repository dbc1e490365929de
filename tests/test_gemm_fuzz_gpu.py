"""Randomised parity sweep of the nn.Linear kernels (tools/gemm_fuzz.py): forward / dgrad / wgrad with every fused
epilogue in both arithmetic modes against fp64, on shapes drawn around the host-side route changes (tile counts next
to multiples of 256, few tiles, ragged M / N / K, 1-3 weight segments, quad-mapped GELU backward, tile lists)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


@pytest.mark.parametrize("seed", [1, 2])
def test_linear_random_shapes_epilogues_and_modes_vs_fp64(seed):
    import gemm_fuzz
    worst, bad = gemm_fuzz.sweep(cases=120, seed=seed, verbose=False)
    assert not bad, bad
