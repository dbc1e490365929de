"""Cvaetf: pvaetf / scavaetf / pscavaetf (reference Model/cvaetf.py:14-193).  Differences
from Vaetf that matter for the checkpoint and the Adam state order: embed_cond2enc is
registered right after the embedding, embed_cond2dec/lat right after the decoder embedding,
mu/log_var/sampling live inside the Encoder, prop_fc precedes out."""
import torch
import torch.nn as nn

from .. import engine, ops
from ..flat import FlatModelMixin, planes_scope
from .layers import DecoderLayer, EncoderLayer
from .modules import Embeddings, Norm, PositionalEncoding, get_clones
from .vaetf import Linear, _TrunkParams, _row_plan, _row_plan_launch


class Encoder(nn.Module, _TrunkParams):
    _dead_prefixes = ("fc_mu", "fc_log_var")  # used by the sampler head, not by the trunk

    def __init__(self, vocab_size, d_model, N, h, dff, latent_dim, nconds, dropout,
                 variational=True, get_attn=False):
        super().__init__()
        self.N, self.nconds, self.variational, self.get_attn = N, nconds, variational, get_attn
        self.d_model, self.p = d_model, dropout
        self.embed_sentence = Embeddings(d_model, vocab_size)
        if nconds > 0:
            self.embed_cond2enc = nn.Linear(nconds, d_model * nconds)
        self.norm = Norm(d_model)
        self.pe = PositionalEncoding(d_model, dropout=dropout)
        self.layers = get_clones(EncoderLayer(h, d_model, dff, dropout, get_attn), N)
        self.fc_mu = nn.Linear(d_model, latent_dim)
        self.fc_log_var = nn.Linear(d_model, latent_dim)
        self.eps_mode = "device"
        self.eps_override = None

    def forward(self, src, src_mask, econds=None, eps=None, _keys=None):
        run = engine.Run(self.p, self.training)
        outs = engine.EncoderFn.apply(self, run, src.contiguous(), ops.to_mask_u8(src_mask), econds,
                                      self.get_attn, _keys, *self.trunk_params())
        x = outs[0] if self.get_attn else outs
        if eps is None and self.eps_override is not None:
            eps = self.eps_override.to(x.device)
        if eps is None and self.variational and self.eps_mode == "cpu":
            eps = torch.randn(x.size(0), x.size(1), self.fc_mu.out_features).to(x.device)
        z, mu, log_var = engine.SamplerFn.apply(x, self.fc_mu.weight, self.fc_mu.bias,
                                                self.fc_log_var.weight, self.fc_log_var.bias, eps,
                                                self.variational)
        if self.get_attn:
            return z, mu, log_var, list(outs[1:])
        return z, mu, log_var


class Decoder(nn.Module, _TrunkParams):
    _dead_prefixes = ("\0",)

    def __init__(self, vocab_size, d_model, N, h, dff, latent_dim, nconds, dropout, use_cond2dec,
                 use_cond2lat, get_attn=False):
        super().__init__()
        self.N, self.nconds, self.d_model, self.get_attn = N, nconds, d_model, get_attn
        self.use_cond2dec, self.use_cond2lat, self.p = use_cond2dec, use_cond2lat, dropout
        self.embed = Embeddings(d_model, vocab_size)
        if self.use_cond2dec and nconds > 0:
            self.embed_cond2dec = nn.Linear(nconds, d_model * nconds)
        if self.use_cond2lat and nconds > 0:
            self.embed_cond2lat = nn.Linear(nconds, d_model * nconds)
        self.pe = PositionalEncoding(d_model, dropout=dropout)
        self.fc_z = nn.Linear(latent_dim, d_model)
        self.layers = get_clones(DecoderLayer(h, d_model, dff, dropout, get_attn), N)
        self.norm = Norm(d_model)

    def forward(self, trg, z, src_mask, trg_mask, dconds=None, loss_rows=None, _compact_out=False, _plan=None):
        """loss_rows: see Model/vaetf.py Decoder.forward (an extension of this build)."""
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


class Cvaetf(FlatModelMixin, nn.Module):
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
        if self.use_cond2dec and nconds > 0:
            self.prop_fc = Linear(trg_vocab, 1)
        self.out = Linear(d_model, trg_vocab)
        self.reset_parameters()

    def reset_parameters(self):
        for _, p in self.named_parameters():        # reference cvaetf.py:162-165
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    @planes_scope
    def encode(self, src, src_mask, econds=None):
        return self.encoder(src, src_mask, econds)[:3]

    @planes_scope
    def decode(self, trg, z, src_mask, trg_mask, dconds=None):
        x = self.decoder(trg, z, src_mask, trg_mask, dconds)
        if self.get_attn:
            x = x[0]
        return self.out(x)

    def plan_ahead(self, src_mask, trg_mask, loss_rows, trg):
        """As Vaetf.plan_ahead."""
        return _row_plan_launch(self, src_mask, trg_mask, loss_rows, trg)

    @planes_scope
    def forward(self, src, trg, src_mask, trg_mask, econds=None, dconds=None, *, loss_rows=None, _plan_ahead=None):
        """Reference signature (Model/cvaetf.py:179) plus the keyword-only loss_rows extension of Vaetf.forward."""
        if self.get_attn or (self.use_cond2dec and self.nconds > 0):
            loss_rows = None
        plan = _plan_ahead.finish() if _plan_ahead is not None else _row_plan(self, src_mask, trg_mask, loss_rows, trg)
        z, mu, log_var = self.encoder(src, src_mask, econds, _keys=plan.enc_keys)[:3]
        d_output = self.decoder(trg, z, src_mask, trg_mask, dconds, loss_rows, _compact_out=True, _plan=plan)
        if self.get_attn:
            d_output = d_output[0]
        output = self.out(d_output)
        live = self.decoder._gct_live_out
        if live is not None:            # the decoder ran on the loss rows only: d_output and the logits are compact [Mc, .]
            self.decoder._gct_live_out = None
            output = engine.ScatterRowsFn.apply(output, live, trg.size(0), trg.size(1))
        if self.use_cond2dec and self.nconds > 0:
            output_prop = self.prop_fc(output[:, :self.nconds, :])
            output_mol = output[:, self.nconds:, :]
        elif self.nconds > 0:
            output_prop = torch.zeros(output.size(0), self.nconds, 1)
            output_mol = output
        else:
            output_prop = None
            output_mol = output
        return output_prop, output_mol, mu, log_var, z
