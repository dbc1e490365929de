"""Vaetf: unconditioned Transformer-VAE (reference Model/vaetf.py:14-182) on the MI355X
engine.  Same constructor signature, sub-module names, registration order (=> identical RNG
consumption at init, identical state_dict keys and Adam state order) and call contract:

    forward(src, trg, src_mask, trg_mask, econds=None, dconds=None)
        -> (output_prop, output_mol, mu, log_var, z)        [+ attention lists with get_attn]
    encode(src, src_mask, econds=None) -> (z, mu, log_var)
    decode(trg, z, src_mask, trg_mask, dconds=None) -> logits
"""
import torch
import torch.nn as nn

from .. import engine, ops
from ..flat import FlatModelMixin, planes_scope
from .layers import DecoderLayer, EncoderLayer
from .modules import Embeddings, Norm, PositionalEncoding, get_clones
from .sublayers import Sampler


class _TrunkParams:
    """Caches the parameter tuple a trunk hands to its autograd.Function."""

    def trunk_params(self):
        ps = self.__dict__.get("_trunk_params_cache")
        if ps is None:
            ps = tuple(p for n, p in self.named_parameters() if not n.startswith(self._dead_prefixes))
            self.__dict__["_trunk_params_cache"] = ps
        return ps


class Encoder(nn.Module, _TrunkParams):
    _dead_prefixes = ("fc_mu", "fc_log_var")  # owned but never used by Vaetf (vaetf.py:26-27)

    def __init__(self, vocab_size, d_model, N, h, dff, latent_dim, nconds, dropout,
                 variational=True, get_attn=False):
        super().__init__()
        self.N, self.nconds, self.variational, self.get_attn = N, nconds, variational, get_attn
        self.d_model, self.p = d_model, dropout
        self.embed_sentence = Embeddings(d_model, vocab_size)
        self.norm = Norm(d_model)
        self.pe = PositionalEncoding(d_model, dropout=dropout)
        self.layers = get_clones(EncoderLayer(h, d_model, dff, dropout, get_attn), N)
        self.fc_mu = nn.Linear(d_model, latent_dim)
        self.fc_log_var = nn.Linear(d_model, latent_dim)
        if nconds > 0:
            self.embed_cond2enc = nn.Linear(nconds, d_model * nconds)

    def trunk(self, src, src_mask, econds, _keys=None):
        run = engine.Run(self.p, self.training)
        outs = engine.EncoderFn.apply(self, run, src.contiguous(), ops.to_mask_u8(src_mask), econds,
                                      self.get_attn, _keys, *self.trunk_params())
        if self.get_attn:
            return outs[0], list(outs[1:])
        return outs, None

    def forward(self, src, src_mask, econds):
        x, attn = self.trunk(src, src_mask, econds)
        return (x, attn) if self.get_attn else x


class Decoder(nn.Module, _TrunkParams):
    _dead_prefixes = ("\0",)

    def __init__(self, vocab_size, d_model, N, h, dff, latent_dim, nconds, dropout, use_cond2dec,
                 use_cond2lat, get_attn=False):
        super().__init__()
        self.N, self.nconds, self.d_model, self.get_attn = N, nconds, d_model, get_attn
        self.use_cond2dec, self.use_cond2lat, self.p = use_cond2dec, use_cond2lat, dropout
        self.embed = Embeddings(d_model, vocab_size)
        self.pe = PositionalEncoding(d_model, dropout=dropout)
        self.fc_z = nn.Linear(latent_dim, d_model)
        self.layers = get_clones(DecoderLayer(h, d_model, dff, dropout, get_attn), N)
        self.norm = Norm(d_model)
        if use_cond2dec and nconds > 0:
            self.embed_cond2dec = nn.Linear(nconds, d_model * nconds)
        if use_cond2lat and nconds > 0:
            self.embed_cond2lat = nn.Linear(nconds, d_model * nconds)

    def forward(self, trg, z, src_mask, trg_mask, dconds, loss_rows=None, _compact_out=False, _plan=None):
        """loss_rows (bool / uint8 [B, T], optional -- an extension of this build): the rows whose output reaches the
        loss; the others are not computed: they come back as zeros, except the (at most 3 per sample) padded rows that
        share an aligned group of four rows with a live one, which hold arbitrary finite values (engine.decoder_trunk_fwd)."""
        run = engine.Run(self.p, self.training)
        if loss_rows is not None:
            loss_rows = loss_rows.to(torch.uint8).contiguous()
        self._gct_live_out = None
        outs = engine.DecoderFn.apply(self, run, trg.contiguous(), z, ops.to_mask_u8(src_mask),
                                      ops.to_mask_u8(trg_mask), dconds, self.get_attn, loss_rows, _plan,
                                      *self.trunk_params())
        if self._gct_live_out is not None and not _compact_out:
            # the trunk ran on the loss rows only and returned them compact [Mc, d]: a caller of the decoder alone gets
            # every row (zeros where nothing was computed); Vaetf / Cvaetf.forward keep the compact rows through the
            # vocabulary head and scatter the logits instead
            live, self._gct_live_out = self._gct_live_out, None
            outs = engine.ScatterRowsFn.apply(outs, live, trg.size(0), trg.size(1))
        if self.get_attn:
            n = self.N
            return outs[0], list(outs[1:1 + n]), list(outs[1 + n:1 + 2 * n])
        return outs


def _row_plan(model, src_mask, trg_mask, loss_rows, trg):
    """engine.RowPlan of a training-style forward: the row maps of both trunks, read back in ONE synchronisation at the
    start of the step (see engine.RowPlan).  Works on the masks exactly as the caller passed them (Model/modules.py)."""
    return _row_plan_launch(model, src_mask, trg_mask, loss_rows, trg).finish()


def _row_plan_launch(model, src_mask, trg_mask, loss_rows, trg):
    """The first half of _row_plan (engine.RowPlan.launch): kernels and the read-back are queued, .finish() gives the plan."""
    if model.get_attn or (model.use_cond2dec and model.nconds > 0):
        loss_rows = None
    if model.get_attn or src_mask is None or not src_mask.is_cuda:
        return engine._PendingPlan(None, None, None, None, 0, 0, 0)
    dec = model.decoder
    sm = ops.to_mask_u8(src_mask)
    B, T = trg.shape
    if sm.dim() != 3 or sm.shape[0] != B or sm.shape[1] != 1:          # not the reference's [B, 1, L] key-padding mask
        return engine._PendingPlan(None, None, None, None, 0, 0, 0)
    Le = sm.shape[2]
    c2d = dec.use_cond2dec and dec.nconds > 0
    nc_lat = dec.nconds if (not c2d and dec.use_cond2lat and dec.nconds > 0) else 0
    lr = None if loss_rows is None else loss_rows.to(torch.uint8).contiguous()
    tm = None if trg_mask is None else ops.to_mask_u8(trg_mask)
    return engine.RowPlan.launch(sm.view(B, Le), tm, lr, B, Le, T if not c2d else T + dec.nconds, nc_lat,
                                 len(model.encoder.layers), len(dec.layers))


class Linear(nn.Linear):
    """nn.Linear parameter holder whose forward runs on the MFMA GEMM (`out`, `prop_fc`)."""

    def forward(self, x):
        return engine.LinearFn.apply(x, self.weight, self.bias)


class Vaetf(FlatModelMixin, nn.Module):
    def __init__(self, src_vocab, trg_vocab, N=6, d_model=256, dff=2048, h=8, latent_dim=64,
                 dropout=0.1, nconds=3, use_cond2dec=False, use_cond2lat=False, variational=True,
                 get_attn=False):
        super().__init__()
        self.nconds, self.get_attn = nconds, get_attn
        self.use_cond2dec, self.use_cond2lat = use_cond2dec, use_cond2lat
        self.encoder = Encoder(src_vocab, d_model, N, h, dff, latent_dim, nconds, dropout,
                               variational, get_attn)
        self.decoder = Decoder(trg_vocab, d_model, N, h, dff, latent_dim, nconds, dropout,
                               use_cond2dec, use_cond2lat, get_attn)
        self.sampler = Sampler(d_model, latent_dim, variational)
        self.out = Linear(d_model, trg_vocab)
        if use_cond2dec and nconds > 0:
            self.prop_fc = Linear(trg_vocab, 1)
        self.reset_parameters()

    def reset_parameters(self):
        for _, p in self.named_parameters():        # reference vaetf.py:140-143
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    @planes_scope
    def encode(self, src, src_mask, econds=None):
        x, _ = self.encoder.trunk(src, src_mask, econds)
        return self.sampler(x)

    @planes_scope
    def decode(self, trg, z, src_mask, trg_mask, dconds=None):
        x = self.decoder(trg, z, src_mask, trg_mask, dconds)
        if self.get_attn:
            x = x[0]
        return self.out(x)

    def plan_ahead(self, src_mask, trg_mask, loss_rows, trg):
        """Queue the row maps of a batch that a later forward(..., _plan_ahead=<the returned object>) will use (the
        trainer: the NEXT batch's, between this step's forward and its backward -- Model/forward_propagation1.prefetch)."""
        return _row_plan_launch(self, src_mask, trg_mask, loss_rows, trg)

    @planes_scope
    def forward(self, src, trg, src_mask, trg_mask, econds=None, dconds=None, *, loss_rows=None, _plan_ahead=None):
        """Reference signature (Model/vaetf.py:154) plus one keyword-only extension: loss_rows (bool [B, T]) names the
        decoder rows whose logits reach the loss -- the trainer passes `ys != pad` (Model/forward_propagation1.py); the
        other rows are then not computed at all: their logits come back as zeros -- or, for the few padded rows that share an
        aligned group of four with a live row, as arbitrary finite values -- NOT as the reference's values.  Default (None): every row, as the reference."""
        if self.get_attn or (self.use_cond2dec and self.nconds > 0):
            loss_rows = None
        plan = _plan_ahead.finish() if _plan_ahead is not None else _row_plan(self, src_mask, trg_mask, loss_rows, trg)
        x, enc_attn = self.encoder.trunk(src, src_mask, econds, _keys=plan.enc_keys)
        z, mu, log_var = self.sampler(x)
        d = self.decoder(trg, z, src_mask, trg_mask, dconds, loss_rows, _compact_out=True, _plan=plan)
        if self.get_attn:
            d, dec_attn_1, dec_attn_2 = d
        output = self.out(d)
        live = self.decoder._gct_live_out
        if live is not None:            # the decoder ran on the loss rows only: d and the logits are compact [Mc, .]
            self.decoder._gct_live_out = None
            output = engine.ScatterRowsFn.apply(output, live, trg.size(0), trg.size(1))
        if self.use_cond2dec:
            output_prop = self.prop_fc(output[:, :self.nconds, :])
            output_mol = output[:, self.nconds:, :]
        else:
            output_prop = torch.zeros(output.size(0), self.nconds, 1)   # CPU, as the reference
            output_mol = output
        if self.get_attn:
            return output_prop, output_mol, mu, log_var, z, enc_attn, dec_attn_1, dec_attn_2
        return output_prop, output_mol, mu, log_var, z
