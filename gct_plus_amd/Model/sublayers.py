"""Sampler / MultiHeadAttention / FeedForward parameter holders (reference
Model/sublayers.py:7-26, 44-74, 77-89).  Registration order of the projections is q, v, k,
out exactly as in the reference so named_parameters() (= Adam state index order) matches."""
import torch
import torch.nn as nn

from .. import engine, ops


class Sampler(nn.Module):
    def __init__(self, d_model, latent_dim, variational):
        super().__init__()
        self.variational = variational
        self.fc_mu = nn.Linear(d_model, latent_dim)
        self.fc_log_var = nn.Linear(d_model, latent_dim)
        self.eps_mode = "device"   # "cpu": draw eps from the CPU torch generator (parity runs)
        self.eps_override = None   # tests: a fixed eps tensor [B, L, latent]

    def forward(self, x, eps=None):
        if eps is None and self.eps_override is not None:
            eps = self.eps_override.to(x.device)
        if eps is None and self.variational and self.eps_mode == "cpu":
            # reference: torch.randn_like(std) on the global generator (sublayers.py:17);
            # bit-equal to torch.randn(shape) on CPU, uploaded for exact-stream parity runs
            eps = torch.randn(x.size(0), x.size(1), self.fc_mu.out_features).to(x.device)
        return engine.SamplerFn.apply(x, self.fc_mu.weight, self.fc_mu.bias,
                                      self.fc_log_var.weight, self.fc_log_var.bias, eps,
                                      self.variational)


class MultiHeadAttention(nn.Module):
    def __init__(self, heads, d_model, dropout=0.1, get_attn=False):
        super().__init__()
        self.d_model = d_model
        self.d_k = d_model // heads
        self.h = heads
        self.get_attn = get_attn
        self.q_linear = nn.Linear(d_model, d_model)
        self.v_linear = nn.Linear(d_model, d_model)
        self.k_linear = nn.Linear(d_model, d_model)
        self.p = dropout
        self.out = nn.Linear(d_model, d_model)

    def forward(self, q, k, v, mask=None):
        """Standalone call (inside the trunks the block is fused with its residual).  k and v
        must be the same tensor, as at every call site of the reference (layers.py:26,61,69)."""
        if k is not v:
            raise ValueError("gct_plus_amd MultiHeadAttention needs k is v (fused K/V projection)")
        run = engine.Run(self.p, self.training)
        outs = engine.MhaFn.apply(self, run, q, None if k is q else k, ops.to_mask_u8(mask),
                                  self.get_attn, *self.parameters())
        return outs if self.get_attn else outs[0]


class FeedForward(nn.Module):
    def __init__(self, d_model, d_ff=2048, dropout=0.1):
        super().__init__()
        self.linear_1 = nn.Linear(d_model, d_ff)
        self.p = dropout
        self.linear_2 = nn.Linear(d_ff, d_model)

    def forward(self, x):
        run = engine.Run(self.p, self.training)
        return engine.FfnFn.apply(self, run, x, *self.parameters())
