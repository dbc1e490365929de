"""N>1 path on CPU: two gloo ranks.  Checks FlatDataParallel's contract against
torch.nn.parallel.DistributedDataParallel (the reference's wrapper, train1.py:111-112):
rank-0 parameter broadcast, averaged gradients (mean of per-rank sum-loss gradients), zero
contribution from parameters that received no gradient, 'module.' prefixed state_dict."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from gct_plus_amd.dp import FlatDataParallel
from gct_plus_amd.flat import FlatModelMixin


class Toy(FlatModelMixin, nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(6, 5)
        self.b = nn.Linear(5, 3)
        self.dead = nn.Linear(4, 4)          # never used: like Vaetf's encoder.fc_mu/fc_log_var

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, flat, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)            # different init per rank: broadcast must fix it
    m = Toy()
    if flat:
        m.flatten_parameters()
    w = FlatDataParallel(m)
    torch.manual_seed(7)
    ref = Toy()
    torch.manual_seed(100)                   # rank 0's init
    ref0 = Toy()
    for (k, v), (_, v0) in zip(m.state_dict().items(), ref0.state_dict().items()):
        assert torch.equal(v, v0), f"broadcast {k}"
    assert all(k.startswith("module.") for k in w.state_dict())
    g = torch.Generator().manual_seed(rank)
    x = torch.randn(8, 6, generator=g)
    for step in range(2):
        for p in m.parameters():
            p.grad = None
        loss = w(x).pow(2).sum()             # per-rank SUM loss, as the reference trainer
        loss.backward()
        grads = {n: (p.grad.clone() if p.grad is not None else None) for n, p in m.named_parameters()}
        # expected: mean over ranks of each rank's own gradient
        exp = {}
        for r in range(world):
            xr = torch.randn(8, 6, generator=torch.Generator().manual_seed(r))
            mm = Toy()
            mm.load_state_dict(m.state_dict())
            mm(xr).pow(2).sum().backward()
            for n, p in mm.named_parameters():
                if p.grad is not None:
                    exp[n] = exp.get(n, 0) + p.grad / world
        for n, gval in grads.items():
            if n.startswith("dead"):
                assert gval is None or float(gval.abs().max()) == 0.0, n
            else:
                assert torch.allclose(gval, exp[n], atol=1e-6), (n, step)
        if flat:
            assert m.grads_are_flat()
    out[rank] = True
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("flat", [True, False])
def test_flat_data_parallel_two_ranks(flat):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, flat, out), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world))


def _merge_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gct_plus_amd.Train.trainer1 import merge_history
    h = {"RCE": [1.0 + rank, 2.0], "KLD": [0.5, 0.25 * (rank + 1)], "LOSS": [3.0, 4.0 + 2 * rank],
         "BETA": [0.1, 0.1], "LR": [1e-4, 2e-4]}
    m = merge_history(h, world)
    if rank == 0:
        torch.save(m, out)
    dist.barrier()
    dist.destroy_process_group()


def test_metrics_merge_two_ranks(tmp_path):
    """trainer1.py:134-151 / 237-252: rank-0's merged CSV = mean over ranks of RCE/KLD/LOSS per step, BETA and LR
    from rank 0 -- here through one all-reduce instead of the per-rank CSV round trip."""
    out = str(tmp_path / "merged.pt")
    mp.spawn(_merge_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    m = torch.load(out)
    assert m["RCE"] == [1.5, 2.0] and m["KLD"] == [0.5, 0.375] and m["LOSS"] == [3.0, 5.0]
    assert m["BETA"] == [0.1, 0.1] and m["LR"] == [1e-4, 2e-4]
