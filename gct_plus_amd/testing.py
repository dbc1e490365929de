"""Test support -- NOT used by any product path (train1.py / FlatDataParallel's default route talk to
RCCL directly).  A one-GPU box cannot give each rank its own device and RCCL refuses two ranks on one
device, so the multi-rank tests (tests/test_dp_gpu.py, `bench.py --share-gpu`) run their ranks on cuda:0
with a gloo process group and stage the two collectives of FlatDataParallel through host memory."""
from __future__ import annotations

import torch.distributed as dist


class _HostStagedWork:
    def __init__(self, t, pg):
        self.t, self.h = t, t.detach().cpu()
        self.w = dist.all_reduce(self.h, op=dist.ReduceOp.SUM, group=pg, async_op=True)

    def wait(self):
        self.w.wait()
        self.t.copy_(self.h)


def host_staged_allreduce(t, pg=None):
    """SUM all-reduce of a device tensor through host memory (async: returns an object with .wait())."""
    return _HostStagedWork(t, pg)


def host_staged_broadcast(t, pg=None):
    h = t.detach().cpu()
    dist.broadcast(h, src=0, group=pg)
    t.copy_(h)
