"""Single-node data parallelism: one process per GPU, gradients exchanged with RCCL over
xGMI (torch.distributed backend "nccl" IS RCCL on ROCm).

Replaces torch.nn.parallel.DistributedDataParallel at train1.py:111-112 of the reference:
  * construction broadcasts the flat parameter buffer from rank 0 (DDP ctor semantics);
  * the model's flat gradient buffer (one contiguous 178 MB range, see flat.py) is split into
    contiguous BUCKETS, one per encoder / decoder LAYER (12.6 / 16.8 MB each) plus the small
    runs between them (embeddings, final norms, fc_z, sampler, out; runs under 2 MB join the
    bucket that follows them in the buffer).  A bucket's averaged all-reduce (ReduceOp.AVG:
    DDP's mean-of-per-rank-sum-loss gradients) is launched the moment its last
    gradient has been written: the trunk backward passes (engine.encoder_trunk_bwd /
    decoder_trunk_bwd, each ONE autograd Function) report every finished layer through
    engine.GRAD_NOTIFY -- post-accumulate-grad hooks cannot see inside a trunk, they only
    fire when the whole trunk returns -- so layer l's exchange runs on RCCL's stream while
    layer l-1 is still in its backward pass and only the first encoder layer's 12.6 MB (plus
    the embedding) is exposed at the end; whatever is left is launched, and everything is
    waited for, in an end-of-backward callback.  Contiguous messages, no bucket copies (DDP's
    25 MB buckets copy every gradient twice);
  * the `pe` buffers are constants, so DDP's per-forward buffer broadcast is dropped;
  * parameters that receive no gradient (Vaetf's dead encoder.fc_*, learned after the first
    backward) contribute zeros instead of tripping DDP's unused-parameter check.
state_dict() keys carry the 'module.' prefix exactly like a DDP-wrapped reference model
(Train/trainer1.py:42 saves from the wrapper; Model/build_model.py:70-71 strips it)."""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn as nn


class FlatDataParallel(nn.Module):
    """`allreduce(tensor, pg) -> work` / `broadcast(tensor, pg)` replace the collectives (SUM semantics; the
    wrapper scales by 1/W): rigs without one GPU per rank inject gct_plus_amd.testing.host_staged_*; the
    product path leaves them None and talks to RCCL directly."""

    def __init__(self, module: nn.Module, process_group=None, overlap: bool = True, allreduce=None,
                 broadcast=None):
        super().__init__()
        self.module = module
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.overlap = overlap
        self._ar_fn, self._bc_fn = allreduce, broadcast
        self._flat_ok = getattr(module, "_gct_flat", None) is not None
        self._armed = False
        self._in_finalize = False
        # Diagnostics for a first run on real multi-GPU hardware (bench.py prints them at N > 1): per backward pass
        # how many buckets / bytes were exchanged from INSIDE the backward (overlapped) and how many from the
        # end-of-backward callback (exposed), and how long the compute stream sat waiting for the exchange there
        # (device time between two events around the waits; host time on CPU rigs).  Off unless `diag` is set.
        self.diag = False
        self.stats = {"backward_passes": 0, "buckets_in_backward": 0, "bytes_in_backward": 0,
                      "buckets_at_finalize": 0, "bytes_at_finalize": 0, "exposed_wait_ms": []}
        self._wait_events = []
        self._nccl = dist.get_backend(process_group) == "nccl" and allreduce is None
        self._avg_native = self._nccl
        if self._nccl:
            # RCCL normally provides ReduceOp.AVG; probe once, fall back to SUM + scale if not
            try:
                probe = torch.ones(4, device=next(module.parameters()).device)
                dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=process_group)
                torch.cuda.synchronize()
                if abs(float(probe[0]) - 1.0) > 1e-6:
                    self._avg_native = False
            except Exception:
                self._avg_native = False
        with torch.no_grad():
            bufs = [module.flat_params()] if self._flat_ok else [p.data for p in module.parameters()]
            if bufs[0].is_cuda and not self._nccl and broadcast is None:
                raise RuntimeError("FlatDataParallel: device tensors need the nccl (RCCL) backend")
            for b in bufs:
                if broadcast is not None:
                    broadcast(b, process_group)
                else:
                    dist.broadcast(b, src=0, group=process_group)
        self._buckets = []
        if self._flat_ok:
            self._make_buckets()

    # ------------------------------------------------------------------ bucket bookkeeping
    MIN_BUCKET_BYTES = 2 << 20          # smaller runs of the flat buffer join the run that follows them

    def _make_buckets(self):
        import re
        flat = self.module._gct_flat
        order, offs, total = flat["order"], flat["offsets"], flat["numel"]
        names = {id(p): n for n, p in self.module.named_parameters()}
        layer_re = re.compile(r"^(encoder|decoder)\.layers\.(\d+)\.")
        groups = []                                   # contiguous runs: one per trunk layer, one per stretch between
        for p, o in zip(order, offs):
            m = layer_re.match(names[id(p)])
            key = m.group(0) if m else names[id(p)].split(".", 1)[0]
            if not groups or groups[-1]["top"] != key:
                groups.append({"top": key, "start": o, "params": []})
            groups[-1]["params"].append(p)
        for i, g in enumerate(groups):
            g["end"] = groups[i + 1]["start"] if i + 1 < len(groups) else total
        merged = []                                   # a tiny run is exchanged together with its successor
        carry = None
        for g in groups:
            if carry is not None:
                g = {"top": carry["top"] + "+" + g["top"], "start": carry["start"], "end": g["end"],
                     "params": carry["params"] + g["params"]}
                carry = None
            if (g["end"] - g["start"]) * 4 < self.MIN_BUCKET_BYTES and g is not groups[-1] and g["end"] < total:
                carry = g
            else:
                merged.append(g)
        if carry is not None:
            merged.append(carry)
        self._buckets = merged
        self._bucket_of = {}
        for bi, g in enumerate(merged):
            for p in g["params"]:
                self._bucket_of[id(p)] = bi
                if self.overlap and p.requires_grad:
                    p.register_post_accumulate_grad_hook(self._on_grad)
        self._dead = set()                            # ids of params that never get a gradient
        self._learned = False
        self._reset_round()

    def _reset_round(self):
        self._fired = set()
        self._works = []
        self._launched = [False] * len(self._buckets)
        self._late = [False] * len(self._buckets)     # a gradient of the bucket did not land in its flat slot yet
        self._left = [sum(1 for p in g["params"] if p.requires_grad and id(p) not in self._dead)
                      for g in self._buckets]

    def _on_grad(self, p):
        if not self._armed or not self.overlap or id(p) in self._fired:      # overlap off: everything from _finalize
            return
        self._fired.add(id(p))
        bi = self._bucket_of[id(p)]
        self._left[bi] -= 1
        if self._left[bi] == 0 and self._learned and not self._launched[bi] and not self._late[bi]:
            self._launch(bi)

    def _notify(self, params, grads):
        """engine.GRAD_NOTIFY: the trunk backward has finished these parameters' gradients (grads[i] is the tensor
        the kernels wrote: the flat slot itself, or a temporary that autograd will accumulate later -- then the bucket
        waits for the end of the backward pass like before)."""
        if not self._armed or not self.overlap:
            return
        ready = []
        for p, gt in zip(params, grads):
            bi = self._bucket_of.get(id(p))
            if bi is None or not p.requires_grad or id(p) in self._dead:
                continue                               # (dead parameters are not counted in _left either)
            if gt is None:
                # the trunk produced no gradient for it: on the first backward that is how the dead set is learned
                # (not fired), afterwards its bucket simply waits for _finalize
                self._late[bi] = True
                continue
            v = getattr(p, "_gct_gview", None)
            if v is None or gt.data_ptr() != v.data_ptr():
                self._late[bi] = True                  # not in its slot: exchanged from _finalize
            ready.append(p)
        for p in ready:
            self._on_grad(p)

    @torch.no_grad()
    def _launch(self, bi):
        g = self._buckets[bi]
        for p in g["params"]:                          # stray / missing gradients -> flat slots
            if p.grad is None:
                # no .grad yet: either a parameter that never gets one (zero contribution), or a gradient the trunk
                # backward has already written into its flat slot and reported (autograd adopts the view afterwards)
                if id(p) not in self._fired:
                    p._gct_gview.zero_()
            elif p.grad.data_ptr() != p._gct_gview.data_ptr():
                p._gct_gview.copy_(p.grad)
                p.grad = p._gct_gview
        t = self.module.flat_grads()[g["start"]:g["end"]]
        self._works.append((self._allreduce(t), t))
        self._launched[bi] = True
        if self.diag:
            k = "at_finalize" if self._in_finalize else "in_backward"
            self.stats["buckets_" + k] += 1
            self.stats["bytes_" + k] += t.numel() * 4

    # ------------------------------------------------------------------------- forward/hooks
    def forward(self, *args, **kwargs):
        if self._armed:
            # a backward pass that raised before its end-of-backward callback left the wrapper armed (and the engine's
            # notifier pointing here): start this pass clean
            self._armed = False
            self._disarm_notifier()
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
        if not self._armed:
            self._armed = True
            if self._flat_ok:
                self._reset_round()
                from . import engine
                engine.GRAD_NOTIFY = self._notify          # finished layers report from inside the trunk backward
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)
        return grad

    def _disarm_notifier(self):
        if self._flat_ok:
            from . import engine
            if engine.GRAD_NOTIFY == self._notify:
                engine.GRAD_NOTIFY = None

    @torch.no_grad()
    def _finalize(self):
        self._armed = False
        m = self.module
        self._disarm_notifier()
        self._in_finalize = True
        try:
            self._finalize_body(m)
        finally:
            self._in_finalize = False

    def _finalize_body(self, m):
        if self._flat_ok:
            if not self._learned:                      # first backward: learn the dead set
                self._dead = {id(p) for g in self._buckets for p in g["params"]
                              if p.requires_grad and id(p) not in self._fired} if self.overlap else set()
                self._learned = True
            for bi in range(len(self._buckets)):
                if not self._launched[bi]:
                    self._launch(bi)
            cuda = self.diag and self._works and self._works[0][1].is_cuda
            if self.diag:
                import time
                self.stats["backward_passes"] += 1
                if cuda:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                else:
                    t0 = time.perf_counter()
            for w, t in self._works:
                w.wait()
                if not self._avg_native:
                    t.mul_(1.0 / self.world)
            if self.diag:
                if cuda:
                    e1.record()                         # resolved later (diag_summary): no synchronisation here
                    self._wait_events.append((e0, e1))
                else:
                    self.stats["exposed_wait_ms"].append((time.perf_counter() - t0) * 1e3)
            self._works = []
        else:
            for p in m.parameters():
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                self._allreduce(p.grad).wait()
                if not self._avg_native:
                    p.grad.mul_(1.0 / self.world)

    def diag_reset(self):
        self.stats = {"backward_passes": 0, "buckets_in_backward": 0, "bytes_in_backward": 0,
                      "buckets_at_finalize": 0, "bytes_at_finalize": 0, "exposed_wait_ms": []}
        self._wait_events = []

    def diag_summary(self):
        """Per-backward-pass averages of the counters + the exposed wait (ms: median and max over the passes).  Resolves
        the pending event pairs, so call it after the timed region (it synchronises on them)."""
        import statistics
        for e0, e1 in self._wait_events:
            e1.synchronize()
            self.stats["exposed_wait_ms"].append(e0.elapsed_time(e1))
        self._wait_events = []
        n = max(self.stats["backward_passes"], 1)
        w = self.stats["exposed_wait_ms"]
        return {"backward_passes": self.stats["backward_passes"], "buckets": len(self._buckets),
                "buckets_in_backward_per_pass": round(self.stats["buckets_in_backward"] / n, 2),
                "mb_in_backward_per_pass": round(self.stats["bytes_in_backward"] / n / 1e6, 2),
                "buckets_at_finalize_per_pass": round(self.stats["buckets_at_finalize"] / n, 2),
                "mb_at_finalize_per_pass": round(self.stats["bytes_at_finalize"] / n / 1e6, 2),
                "exposed_wait_ms_median": round(statistics.median(w), 3) if w else None,
                "exposed_wait_ms_max": round(max(w), 3) if w else None,
                "overlap": bool(self.overlap), "avg_native": bool(self._avg_native)}

    def _allreduce(self, t):
        if self._ar_fn is not None:
            return self._ar_fn(t, self.pg)
        op = dist.ReduceOp.AVG if self._avg_native else dist.ReduceOp.SUM
        return dist.all_reduce(t, op=op, group=self.pg, async_op=True)
