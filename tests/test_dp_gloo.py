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


class _TrunkToy(FlatModelMixin, nn.Module):
    """Two 'encoder layers' behind ONE autograd Function with a hand-written backward that writes the gradients through
    engine.GradSink and reports each finished layer through engine._grads_done -- the structure of
    engine.EncoderFn / DecoderFn, on CPU tensors."""

    def __init__(self):
        super().__init__()
        self.encoder = nn.Module()
        self.encoder.layers = nn.ModuleList([nn.Linear(6, 6), nn.Linear(6, 6)])
        self.out = nn.Linear(6, 3)

    def forward(self, x):
        return self.out(_TrunkFn.apply(self, x, *self.encoder.parameters()))


_EVENTS = []


class _TrunkFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, *params):
        l0, l1 = mod.encoder.layers
        a = torch.tanh(x @ l0.weight.t() + l0.bias)
        ctx.mod, ctx.x, ctx.a, ctx.params = mod, x, a, params
        return a @ l1.weight.t() + l1.bias

    @staticmethod
    def backward(ctx, dy):
        from gct_plus_amd import engine
        l0, l1 = ctx.mod.encoder.layers
        G = engine.GradSink()
        G(l1.weight).copy_(dy.t() @ ctx.a)
        G(l1.bias).copy_(dy.sum(0))
        engine._grads_done(l1, G)                         # the last layer finishes first
        _EVENTS.append("layer1 done")
        dh = (dy @ l1.weight) * (1 - ctx.a * ctx.a)
        G(l0.weight).copy_(dh.t() @ ctx.x)
        G(l0.bias).copy_(dh.sum(0))
        engine._grads_done(l0, G)
        _EVENTS.append("layer0 done")
        return (None, dh @ l0.weight) + G.collect(ctx.params)


def _trunk_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    FlatDataParallel.MIN_BUCKET_BYTES = 0                 # every run its own bucket: encoder.layers.0 | .1 | out
    torch.manual_seed(5)
    m = _TrunkToy()
    m.flatten_parameters()
    w = FlatDataParallel(m)
    assert [g["top"] for g in w._buckets] == ["encoder.layers.0.", "encoder.layers.1.", "out"]
    launches = []
    real = w._launch
    w._launch = lambda bi: (launches.append((w._buckets[bi]["top"], list(_EVENTS))), real(bi))[1]
    x = torch.randn(8, 6, generator=torch.Generator().manual_seed(rank))
    for step in range(3):
        for p in m.parameters():
            p.grad = None
        del launches[:], _EVENTS[:]
        w(x).pow(2).sum().backward()
        exp = {}
        for r in range(world):                            # mean over ranks of each rank's own (plain autograd) gradient
            xr = torch.randn(8, 6, generator=torch.Generator().manual_seed(r))
            l0, l1 = m.encoder.layers
            ps = [p.detach().clone().requires_grad_(True) for p in (l0.weight, l0.bias, l1.weight, l1.bias,
                                                                   m.out.weight, m.out.bias)]
            y = (torch.tanh(xr @ ps[0].t() + ps[1]) @ ps[2].t() + ps[3]) @ ps[4].t() + ps[5]
            y.pow(2).sum().backward()
            for n, p in zip(["encoder.layers.0.weight", "encoder.layers.0.bias", "encoder.layers.1.weight",
                             "encoder.layers.1.bias", "out.weight", "out.bias"], ps):
                exp[n] = exp.get(n, 0) + p.grad / world
        for n, p in m.named_parameters():
            assert torch.allclose(p.grad, exp[n], atol=1e-6), (n, step)
        assert m.grads_are_flat()
        order = [t for t, _ in launches]
        if step == 0:                                     # first backward: the dead set is learned, nothing is early
            assert all(ev == ["layer1 done", "layer0 done"] for _, ev in launches), launches
        else:
            # the head's bucket goes first (its AccumulateGrad fires before the trunk runs), layer 1's bucket is
            # launched from INSIDE the trunk backward -- before layer 0 has been computed -- and layer 0's last
            assert order == ["out", "encoder.layers.1.", "encoder.layers.0."], order
            ev = dict(launches)
            assert ev["out"] == [] and ev["encoder.layers.1."] == [] and ev["encoder.layers.0."] == ["layer1 done"], launches
    # a gradient that cannot go to its flat slot (.grad kept from the last step): the bucket falls back to the end
    del launches[:], _EVENTS[:]
    w(x).pow(2).sum().backward()                          # no `p.grad = None`: accumulation semantics
    assert all(ev == ["layer1 done", "layer0 done"] for t, ev in launches if t.startswith("encoder")), launches
    out[rank] = True
    dist.barrier()
    dist.destroy_process_group()


def test_layer_buckets_launch_from_inside_the_trunk_backward():
    """dp.FlatDataParallel + engine.GRAD_NOTIFY: per-layer buckets of the flat gradient buffer are exchanged while the
    layers underneath are still in their backward pass (two gloo ranks, a toy trunk with a hand-written backward)."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_trunk_worker, args=(world, _free_port(), out), nprocs=world, join=True)
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
