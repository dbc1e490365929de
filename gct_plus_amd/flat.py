"""Flat parameter / gradient storage for a whole model (HBM layout).

All parameters live back to back (each 16-B aligned) in ONE fp32 buffer and all gradients in
a second one, in named_parameters() order.  nn.Parameter objects keep their names and shapes
(the checkpoint layout is untouched: state_dict() still yields the reference's keys) -- they
are views into the flat buffer.  This is what makes the fused Adam a single launch, the
data-parallel gradient exchange a handful of large RCCL calls over contiguous ranges, and
lets the wgrad kernels write gradients in place (engine.GradSink).
"""
from __future__ import annotations

import functools
from typing import List

import torch
import torch.nn as nn

ALIGN = 8  # floats: 16-B aligned fp32 slots AND 16-B aligned slots in the bf16 weight planes

# Gradient-slot ownership inside ONE backward pass (engine.GradSink): a parameter's slot of the flat gradient
# buffer may be handed to the kernels of only one autograd.Function per pass.  If the model (or model.decode /
# a sub-module) is used twice inside one graph -- (loss1 + loss2).backward() -- the second Function must write
# to a temporary, otherwise autograd's input buffer would add a tensor to its own alias (2*g2 instead of g1+g2).
# The id is released by the parameter's post-accumulate hook, and wholesale by sync_grads_to_flat() (optimizer
# step: no backward pass is in flight), which also recovers from passes that never accumulate (autograd.grad).
HANDED_SLOTS = set()


def _release_slot(p):
    HANDED_SLOTS.discard(id(p))


class FlatModelMixin:
    """Mixed into the top-level model classes.  Flattening happens automatically whenever
    the module is moved to a CUDA device (`model.cuda()` in train1.py:102)."""

    _gct_flat = None
    _gct_depth = 0

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        first = next(self.parameters(), None)
        if first is not None and first.is_cuda:
            self.flatten_parameters()
        else:
            self._gct_flat = None
            for p in self.parameters():
                if hasattr(p, "_gct_gview"):
                    del p._gct_gview
        return out

    def flatten_parameters(self):
        params: List[nn.Parameter] = list(self.parameters())
        if not params:
            return
        dev = params[0].device
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        pflat = torch.zeros(total, dtype=torch.float32, device=dev)
        gflat = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(params, offs):
                v = pflat[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                p._gct_gview = gflat[o:o + p.numel()].view(p.shape)
                if not getattr(p, "_gct_release_hook", False):
                    p.register_post_accumulate_grad_hook(_release_slot)
                    p._gct_release_hook = True
                if p.grad is not None:
                    p._gct_gview.copy_(p.grad)
                    p.grad = None
        old = self._gct_flat
        if old is not None and "planes" in old:
            from . import ops
            ops.unregister_planes(old["params"])
        self._gct_flat = {"params": pflat, "grads": gflat, "offsets": offs, "order": params,
                          "numel": total}
        self._install_plane_hooks()

    # -- bf16x6 GEMM mode: the weights' bf16 planes mirror the flat parameter buffer ---------
    def refresh_weight_planes(self):
        """Re-split the flat parameter buffer into its three bf16 planes (one elementwise launch,
        ~0.15 ms for the 44.6 M-parameter model).  Runs at every entry into the model (forward
        pre-hooks below), so the planes can never be older than the weights a forward uses; the
        backward of that forward reuses them (weights do not change in between)."""
        f = self._gct_flat
        if f is None or not f["params"].is_cuda:
            return
        from . import ops
        if ops.gemm_get_mode() != ops.GEMM_BF16X6:
            if "planes" in f:
                ops.unregister_planes(f["params"])
                del f["planes"]
            return
        if "planes" not in f:
            f["planes"] = torch.empty(3, f["numel"], dtype=torch.int16, device=f["params"].device)
        ops.split_planes(f["params"], f["planes"])
        ops.register_planes(f["params"], f["planes"])

    def weights_token(self):
        """Changes whenever the weights may have changed: (raw-write epoch, version counters of the flat buffer and of
        every parameter, the buffer's address).  A torch in-place operation bumps the counter of the tensor it is applied
        to -- a parameter (p.mul_, load_state_dict's copy_: the parameter's own counter, `p.data = view` does not share
        the buffer's) or the flat buffer itself (the data-parallel broadcast); the one writer that goes behind torch's
        back -- FusedAdam's kernel -- reports through invalidate_weight_planes(), which bumps the epoch.  Caches of
        anything derived from the weights alone (decode.KVDecoder's folded cross-attention projections) key on it.
        NOT seen: writes through `p.data` (p.data.copy_, p.data.mul_ -- EMA or manual weight surgery: `.data` carries its
        own version counter) and raw kernels of the caller's own.  Such a writer calls invalidate_weight_planes() after
        writing (it bumps the epoch) -- or decodes with KVDecoder.start(..., refold=True)."""
        f = self._gct_flat
        if f is None:
            return None
        return (getattr(self, "_gct_epoch", 0), f["params"]._version, f["params"].data_ptr(),
                tuple(p._version for p in f["order"]))

    def invalidate_weight_planes(self):
        """Called by whoever rewrites the weights behind autograd's back (FusedAdam): GEMMs fall back
        to the fp32 kernels until the next refresh instead of reading stale planes."""
        self._gct_epoch = getattr(self, "_gct_epoch", 0) + 1
        f = self._gct_flat
        if f is not None and "planes" in f:
            from . import ops
            ops.unregister_planes(f["params"])

    def _install_plane_hooks(self):
        """Trunks called directly (model.encoder(...), model.decoder(...): the sampling scripts)
        refresh the planes themselves; inside forward / encode / decode (`planes_scope`) they do not."""
        if getattr(self, "_gct_plane_hooks", False):
            return
        self._gct_plane_hooks = True

        def child_pre(child, args):
            if self._gct_depth == 0:
                self.refresh_weight_planes()

        for c in self.children():
            c.register_forward_pre_hook(child_pre)


    # -- helpers used by the fused optimiser and the data-parallel wrapper ------------------
    def flat_params(self) -> torch.Tensor:
        self._require_flat()
        return self._gct_flat["params"]

    def flat_grads(self) -> torch.Tensor:
        self._require_flat()
        return self._gct_flat["grads"]

    def grads_are_flat(self) -> bool:
        """True when every existing .grad aliases its slot of the flat gradient buffer."""
        if self._gct_flat is None:
            return False
        for p in self._gct_flat["order"]:
            if p.grad is not None and p.grad.data_ptr() != p._gct_gview.data_ptr():
                return False
        return True

    def sync_grads_to_flat(self):
        """Make the flat gradient buffer authoritative: copy stray .grad tensors in, zero the
        slots of parameters that received no gradient this step (Vaetf's dead encoder.fc_*,
        SURVEY 2.3) and point every .grad at its slot."""
        self._require_flat()
        HANDED_SLOTS.clear()
        for p in self._gct_flat["order"]:
            if p.grad is None:
                p._gct_gview.zero_()
            elif p.grad.data_ptr() != p._gct_gview.data_ptr():
                p._gct_gview.copy_(p.grad)
                p.grad = p._gct_gview

    def _require_flat(self):
        if self._gct_flat is None:
            raise RuntimeError("model parameters are not flattened: move the model to a ROCm "
                               "device first (model.cuda())")


def planes_scope(fn):
    """Decorator for the top-level entry points (forward / encode / decode): refresh the bf16 weight
    planes once on entry (the reference calls model.forward directly, Model/forward_propagation1.py,
    so module hooks would not see it)."""
    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        if self._gct_depth == 0:
            self.refresh_weight_planes()
        self._gct_depth += 1
        try:
            return fn(self, *a, **kw)
        finally:
            self._gct_depth -= 1
    return wrapped
