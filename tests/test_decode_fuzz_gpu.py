"""Randomised sweep of the KV-cached decoder (tools/decode_fuzz.py): token ids equal to the reference-style loop over
the un-cached model.decode on random model types / sizes / batch sizes (1 ... 1 200 rows: panel, skinny and bf16x6 GEMM
routes), memory lengths, source masks, scaffold prefixes, cond2dec, <eos> stop, with and without graph replay."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


def test_kv_decode_random_configurations_match_uncached_loop():
    import decode_fuzz
    bad = decode_fuzz.sweep(cases=30, seed=1, verbose=False)
    assert not bad, bad
