"""EncoderLayer / DecoderLayer parameter holders (reference Model/layers.py:8-38, 41-82)."""
import torch.nn as nn

from .. import engine, ops
from .modules import Norm
from .sublayers import FeedForward, MultiHeadAttention


class EncoderLayer(nn.Module):
    """x = norm_1(x); x = x + drop(attn(x)); x = norm_2(x); x = x + drop(ff(x)) -- the
    residual branches start from the NORMALISED tensor (reference layers.py:20-38)."""

    def __init__(self, heads, d_model, dff, dropout, get_attn=False):
        super().__init__()
        self.get_attn = get_attn
        self.norm_1 = Norm(d_model)
        self.attn = MultiHeadAttention(heads, d_model, dropout, get_attn)
        self.norm_2 = Norm(d_model)
        self.ff = FeedForward(d_model, dff, dropout)
        self.p = dropout

    def forward(self, x, mask):
        run = engine.Run(self.p, self.training)
        outs = engine.EncLayerFn.apply(self, run, x, ops.to_mask_u8(mask), self.get_attn,
                                       *self.parameters())
        return outs if self.get_attn else outs[0]


class DecoderLayer(nn.Module):
    """Standard pre-norm decoder layer (reference layers.py:56-82)."""

    def __init__(self, heads, d_model, dff, dropout, get_attn=False):
        super().__init__()
        self.get_attn = get_attn
        self.norm_1 = Norm(d_model)
        self.attn_1 = MultiHeadAttention(heads, d_model, dropout, get_attn)
        self.norm_2 = Norm(d_model)
        self.attn_2 = MultiHeadAttention(heads, d_model, dropout, get_attn)
        self.norm_3 = Norm(d_model)
        self.ff = FeedForward(d_model, dff, dropout)
        self.p = dropout

    def forward(self, x, e_outputs, src_mask, trg_mask):
        run = engine.Run(self.p, self.training)
        outs = engine.DecLayerFn.apply(self, run, x, e_outputs, ops.to_mask_u8(src_mask),
                                       ops.to_mask_u8(trg_mask), self.get_attn, *self.parameters())
        return outs if self.get_attn else outs[0]
