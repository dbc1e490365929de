"""Randomised parity sweep of the attention kernels (tools/attn_fuzz.py): random B, H, head dim, lengths (L_k <= 96: the
barrier-free kernels; L_q up to 130), mask kinds (none / key padding / banded causal / random holes / rows that see no
key) and zero gradient rows, forward and backward against fp64."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


def test_attention_random_shapes_and_masks_vs_fp64():
    import attn_fuzz
    worst = attn_fuzz.sweep(cases=60, dropout=0.0)
    assert worst <= 1.0
