"""Single-node data parallelism: one process per GPU, gradients exchanged with RCCL over
xGMI (torch.distributed backend "nccl" IS RCCL on ROCm).

Replaces torch.nn.parallel.DistributedDataParallel at train1.py:111-112 of the reference:
  * construction broadcasts the flat parameter buffer from rank 0 (DDP ctor semantics);
  * after every backward the model's flat gradient buffer (one contiguous 178 MB range, see
    flat.py) is all-reduced and averaged (DDP's mean-of-per-rank-sum-losses semantics,
    SURVEY.md 2.3) -- a few large collectives instead of DDP's 25 MB bucket copies;
  * the `pe` buffers are constants, so DDP's per-forward buffer broadcast is dropped;
  * parameters that received no gradient (Vaetf's dead encoder.fc_*) contribute zeros
    instead of tripping DDP's unused-parameter check.
state_dict() keys carry the 'module.' prefix exactly like a DDP-wrapped reference model
(Train/trainer1.py:42 saves from the wrapper; Model/build_model.py:70-71 strips it)."""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn as nn


class FlatDataParallel(nn.Module):
    def __init__(self, module: nn.Module, n_chunks: int = 4, process_group=None):
        super().__init__()
        self.module = module
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.n_chunks = max(1, n_chunks)
        self._flat_ok = getattr(module, "_gct_flat", None) is not None
        self._pending = False
        with torch.no_grad():
            bufs = [module.flat_params()] if self._flat_ok else [p.data for p in module.parameters()]
            staged = dist.get_backend(process_group) != "nccl" and bufs[0].is_cuda
            for b in bufs:
                if staged:
                    h = b.cpu()
                    dist.broadcast(h, src=0, group=process_group)
                    b.copy_(h)
                else:
                    dist.broadcast(b, src=0, group=process_group)
        self._avg_native = dist.get_backend(process_group) == "nccl"

    def forward(self, *args, **kwargs):
        out = self.module(*args, **kwargs)
        if torch.is_grad_enabled():
            anchor = next((o for o in (out if isinstance(out, (tuple, list)) else (out,))
                           if isinstance(o, torch.Tensor) and o.requires_grad), None)
            if anchor is not None:
                anchor.register_hook(self._arm)
        return out

    # the hook on the first output gradient queues a callback that fires once the whole
    # backward pass (every AccumulateGrad) has finished
    def _arm(self, grad):
        if not self._pending:
            self._pending = True
            torch.autograd.Variable._execution_engine.queue_callback(self._reduce)
        return grad

    @torch.no_grad()
    def _reduce(self):
        self._pending = False
        m = self.module
        if self._flat_ok:
            m.sync_grads_to_flat()
            flat = m.flat_grads()
            n = flat.numel()
            step = (n + self.n_chunks - 1) // self.n_chunks
            works = []
            for s in range(0, n, step):
                works.append(self._allreduce(flat[s:s + step]))
            for w in works:
                w.wait()
            if not self._avg_native:
                flat.mul_(1.0 / self.world)
        else:
            for p in m.parameters():
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                self._allreduce(p.grad).wait()
                if not self._avg_native:
                    p.grad.mul_(1.0 / self.world)

    def _allreduce(self, t):
        op = dist.ReduceOp.AVG if self._avg_native else dist.ReduceOp.SUM
        if t.is_cuda and not self._avg_native:
            # test-only route (gloo ranks sharing one GPU): stage through host memory
            return _HostStaged(t, self.pg)
        return dist.all_reduce(t, op=op, group=self.pg, async_op=True)


class _HostStaged:
    def __init__(self, t, pg):
        self.t, self.h = t, t.detach().cpu()
        self.w = dist.all_reduce(self.h, op=dist.ReduceOp.SUM, group=pg, async_op=True)

    def wait(self):
        self.w.wait()
        self.t.copy_(self.h)
