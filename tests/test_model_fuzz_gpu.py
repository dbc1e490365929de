"""Randomised model-level parity sweep (tools/model_fuzz.py): forward, loss and every gradient of the HIP training step
against the CPU oracle on random small configurations -- the four model types, 1-3 layers, head dims 16 / 32 / 64,
batch 1-20, padded lengths 3-140, ragged / full / very short / one-long-sample batches, cond2dec."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


def test_training_step_random_configurations_vs_oracle():
    import model_fuzz
    worst, bad = model_fuzz.sweep(cases=30, seed=1, verbose=False)
    assert not bad, bad
