"""Norm / Embeddings / PositionalEncoding parameter holders and the mask builders.

Mirrors the names, constructor order and state_dict keys of the reference's
Model/modules.py (Norm :80-95, Embeddings :101-110, PositionalEncoding :116-144, masks
:17-58, get_clones :73-74); the arithmetic runs in the HIP kernels (K1/K2) via
gct_plus_amd.engine -- these classes only hold parameters and dispatch.
"""
import copy
import math

import torch
import torch.nn as nn

from .. import engine, ops


def get_clones(layer, N):
    """N deep copies => identical initial values in every layer until reset_parameters
    (reference Model/modules.py:73-74)."""
    return nn.ModuleList([copy.deepcopy(layer) for _ in range(N)])


# --------------------------------------------------------------------------------- masks
def nopeak_mask(trg_size, use_cond2dec, pad_idx, cond_dim=0, device=None):
    """Causal 'may attend' pattern times pad_idx (int64, reference modules.py:17-30), built
    directly on `device` (the reference builds it with numpy on the host every step)."""
    allow = torch.ones(trg_size, trg_size, dtype=torch.bool, device=device).tril_()
    if use_cond2dec:
        n = cond_dim + trg_size
        full = torch.zeros(n, n, dtype=torch.bool, device=device)
        full[:cond_dim, :cond_dim] = True
        full[:cond_dim, cond_dim] = True
        full[cond_dim:, :cond_dim] = True
        full[cond_dim:, cond_dim:] = allow
        allow = full
    return allow.unsqueeze(0) * pad_idx


def get_cond_mask(conditions):
    return torch.ones_like(conditions.unsqueeze(-2), dtype=torch.bool)


def get_src_mask(src, pad_idx, conditions=None):
    """(bs,1,nc+len) key-padding mask (reference modules.py:38-44)."""
    m = (src != pad_idx).unsqueeze(-2)
    if conditions is not None:
        m = torch.cat([get_cond_mask(conditions).to(m.device), m], dim=2)
    return m


def get_trg_mask(target, pad_id, use_cond2dec, conditions=None):
    """pad mask & no-peek mask (reference modules.py:47-58); works on any device."""
    m = (target != pad_id).unsqueeze(-2)
    if use_cond2dec:
        m = torch.cat([get_cond_mask(conditions).to(m.device), m], dim=2)
    cond_dim = 0 if conditions is None else conditions.size(-1)
    return m & nopeak_mask(target.size(1), use_cond2dec, pad_id, cond_dim, device=target.device)


def get_masks(source, target, conditions, pad_idx, use_cond2dec=True):
    return (get_src_mask(source, pad_idx, conditions),
            get_trg_mask(target, pad_idx, use_cond2dec, conditions))


# ------------------------------------------------------------------------------- modules
class Norm(nn.Module):
    """alpha*(x-mean)/(std_unbiased+eps)+bias with eps=1e-6 on the STD (K2, norm.hip)."""

    def __init__(self, d_model, eps=1e-6):
        super().__init__()
        self.size = d_model
        self.alpha = nn.Parameter(torch.ones(self.size))
        self.bias = nn.Parameter(torch.zeros(self.size))
        self.eps = eps

    def forward(self, x):
        return engine.NormFn.apply(x, self.alpha, self.bias, self.eps)


class Embeddings(nn.Module):
    """Plain lookup; the sqrt(d) scale lives in PositionalEncoding (reference :108-110)."""

    def __init__(self, d_model, vocab):
        super().__init__()
        self.embed = nn.Embedding(vocab, d_model)
        self.d_model = d_model

    def forward(self, x):
        return engine.EmbedFn.apply(x, self.embed.weight)


def pe_table(d_model, max_seq_len=200):
    """pe[pos,i]=sin(pos/10000^(2i/d)), pe[pos,i+1]=cos(pos/10000^(2(i+1)/d)) for even i --
    the reference's non-Vaswani exponents (modules.py:123-131), evaluated in Python floats
    exactly like the reference so the fp32 buffer is bit-identical."""
    pe = torch.zeros(max_seq_len, d_model)
    for pos in range(max_seq_len):
        for i in range(0, d_model, 2):
            pe[pos, i] = math.sin(pos / (10000 ** ((2 * i) / d_model)))
            pe[pos, i + 1] = math.cos(pos / (10000 ** ((2 * (i + 1)) / d_model)))
    return pe.unsqueeze(0)


class PositionalEncoding(nn.Module):
    """Holds the `pe` buffer (part of the checkpoint) and the dropout rate.  Inside the
    encoder/decoder trunks scale+pe+dropout is fused into the embedding kernel (K1)."""

    def __init__(self, d_model, max_seq_len=200, dropout=0.1):
        super().__init__()
        self.d_model = d_model
        self.p = dropout
        self.register_buffer("pe", pe_table(d_model, max_seq_len))

    def forward(self, x):
        return engine.PosEncFn.apply(x, self.pe, math.sqrt(self.d_model),
                                     engine.Run(self.p, self.training))
