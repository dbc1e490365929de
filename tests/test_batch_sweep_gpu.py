"""Full-size training steps at batch sizes other than the benchmark's (tools/batch_sweep.py): the data-dependent
shortcuts (compacted decoder backward, visible-rows cross-attention K/V, tail-balanced GEMMs as the tile counts fall)
must give the losses of the plain path.  Batch 128 -- the reference scripts' own batch size -- is the case that used
to read the GELU pre-activation out of bounds in a tail-balanced, quad-mapped dgrad."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


@pytest.mark.parametrize("mtype,B", [("vaetf", 24), ("vaetf", 128), ("pscavaetf", 200), ("vaetf", 512), ("pvaetf", 512)])
def test_shortcuts_agree_with_plain_path(mtype, B):
    import batch_sweep
    rel, on, off = batch_sweep.compare(mtype, B)
    assert rel < 2e-5, (on, off)
