import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _module_switches_restored():
    """The package keeps a few process-wide switches (engine.COMPACT_*, ops.SIDE_ENABLED, the GEMM arithmetic mode);
    tests and the sweep tools flip them.  Whatever a test leaves behind is put back, so that a switch flipped by one
    test can never change what a later one measures (round 3: a sweep left the compaction off for five later tests)."""
    from gct_plus_amd import engine, ops
    names = [n for n in dir(engine) if n.startswith("COMPACT_")]
    keep = {n: getattr(engine, n) for n in names}
    side, defer = ops.SIDE_ENABLED, ops.DEFER_REDUCTIONS
    mode = None
    try:
        import torch
        if torch.cuda.is_available():
            mode = ops.gemm_get_mode()
    except Exception:                      # noqa: BLE001 -- no GPU / library: nothing to restore
        mode = None
    yield
    for n, v in keep.items():
        setattr(engine, n, v)
    ops.SIDE_ENABLED, ops.DEFER_REDUCTIONS = side, defer
    if mode is not None and ops.gemm_get_mode() != mode:
        ops.gemm_set_mode(mode)
