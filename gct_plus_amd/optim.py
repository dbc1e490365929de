"""Fused Adam (K8) with a torch.optim.Adam-compatible state_dict.

train1.py:116-119 of the reference builds torch.optim.Adam(lr=1e-4, betas=(0.9,0.98),
eps=1e-9) and Train/trainer1.py:33-46 checkpoints optimizer.state_dict(); resume
(train1.py:125-129) loads it back.  This class keeps that layout -- state[p] = {step,
exp_avg, exp_avg_sq}, param_groups with torch's keys, indexed in parameters() order -- while
the update itself is ONE launch of gct_adam_step over the model's flat parameter / gradient /
moment buffers (28 B of HBM traffic per parameter)."""
from __future__ import annotations

import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, model=None):
        defaults = dict(torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))]).defaults)
        defaults.update(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False)
        super().__init__(params, defaults)
        self.model = model
        self.grad_scale = 1.0          # 1/W when gradients arrive SUM-reduced (see dp.py)
        self._flat = None
        self._t = 0

    # ---- flat fast path --------------------------------------------------------------
    def _try_flat(self):
        m = self.model
        if m is None or getattr(m, "_gct_flat", None) is None or len(self.param_groups) != 1:
            return None
        order = m._gct_flat["order"]
        mine = self.param_groups[0]["params"]
        if len(order) != len(mine) or any(a is not b for a, b in zip(order, mine)):
            return None
        if self._flat is None or self._flat["pbuf"] is not m._gct_flat["params"]:
            n = m._gct_flat["numel"]
            dev = m._gct_flat["params"].device
            ea = torch.zeros(n, dtype=torch.float32, device=dev)
            es = torch.zeros(n, dtype=torch.float32, device=dev)
            for p, o in zip(order, m._gct_flat["offsets"]):
                st = self.state[p]
                va, vs = ea[o:o + p.numel()].view(p.shape), es[o:o + p.numel()].view(p.shape)
                if "exp_avg" in st:                   # adopt loaded / earlier state
                    va.copy_(st["exp_avg"])
                    vs.copy_(st["exp_avg_sq"])
                    self._t = max(self._t, int(float(st["step"])))
                st["exp_avg"], st["exp_avg_sq"] = va, vs
                st.setdefault("step", torch.tensor(float(self._t)))
            self._flat = {"pbuf": m._gct_flat["params"], "m": ea, "v": es}
        return self._flat

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        flat = self._try_flat()
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        if flat is not None:
            self.model.sync_grads_to_flat()
            self._t += 1
            ops.adam_step(flat["pbuf"], self.model.flat_grads(), flat["m"], flat["v"],
                          float(g["lr"]), b1, b2, g["eps"], self._t, self.grad_scale)
            self.model.invalidate_weight_planes()      # bf16 planes are re-split at the next forward
            step_t = torch.tensor(float(self._t))
            for p in g["params"]:
                self.state[p]["step"] = step_t
            return loss
        # generic path: one launch per parameter tensor (non-flattened models)
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if "exp_avg" not in st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                t = int(float(st["step"])) + 1
                st["step"] = torch.tensor(float(t))
                ops.adam_step(p.data, p.grad.contiguous(), st["exp_avg"], st["exp_avg_sq"],
                              float(group["lr"]), b1, b2, group["eps"], t, self.grad_scale)
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._flat = None                              # re-adopt the loaded moments lazily
        steps = [int(float(s["step"])) for s in self.state.values() if "step" in s]
        self._t = max(steps) if steps else 0
