"""Hand-written forward/backward passes of the GCT-Plus Transformer-VAE on top of the HIP
kernels (gct_plus_amd.ops).  No PyTorch arithmetic runs here: torch is used for device
memory (caching allocator), streams and the autograd *boundary* -- one autograd.Function per
trunk (encoder, sampler, decoder, linear, losses); inside a trunk the backward is explicit,
so residual-gradient joins, the 6-way accumulation of the encoder-memory gradient and the
dropout masks are handled by kernels/epilogues instead of autograd nodes.

Reference semantics (file:line relative to /root/reference):
  encoder layer   Model/layers.py:20-38   (residuals start from the NORMALISED x)
  decoder layer   Model/layers.py:56-82   (standard pre-norm)
  MHA             Model/sublayers.py:61-74, attention() :29-41
  FFN             Model/sublayers.py:85-89
  encoder/decoder Model/vaetf.py:32-54, 79-114 ; Model/cvaetf.py:35-61, 93-133
  sampler         Model/sublayers.py:14-26
"""
from __future__ import annotations

import math
import os
from typing import List, Optional

import torch

from . import ops
from .flat import HANDED_SLOTS

_SEED = {"base": None, "ctr": 0}

# Compacted decoder backward (decoder_trunk_bwd): on by default, GCT_COMPACT_BWD=0 keeps the dense path.  Taken only
# when the compact rows are at most this fraction of all rows (the gathers cost ~1 ms per step).
COMPACT_BWD = os.environ.get("GCT_COMPACT_BWD", "1") != "0"
COMPACT_MAX_FRACTION = 0.85
# Cross-attention over the visible rows of the encoder memory only (decoder_trunk_fwd): GCT_COMPACT_KV=0 disables.
COMPACT_KV = os.environ.get("GCT_COMPACT_KV", "1") != "0"
# Decoder FORWARD over the rows that reach the loss only (decoder_trunk_fwd(loss_rows=...)): GCT_COMPACT_FWD=0 disables.
COMPACT_FWD = os.environ.get("GCT_COMPACT_FWD", "1") != "0"
# K | V projections of the ENCODER self-attention over the visible rows only (mha_fwd): GCT_COMPACT_ENC_KV=0 disables.
COMPACT_ENC_KV = os.environ.get("GCT_COMPACT_ENC_KV", "1") != "0"


# Data parallelism (dp.FlatDataParallel) sets this while a backward pass is running: called as
# GRAD_NOTIFY(params, grads) when a trunk backward has finished writing the gradients of a layer, so that the layer's
# bucket of the flat gradient buffer can be exchanged while the layers underneath are still in their backward pass
# (autograd's post-accumulate hooks only fire when the whole trunk Function returns).
GRAD_NOTIFY = None


def _grads_done(module, G: "GradSink"):
    if GRAD_NOTIFY is not None:
        # with GCT_SIDE_STREAM=1 the layer's weight gradients were written on the side stream: the exchange that the
        # notifier may launch is ordered behind the CURRENT stream only, so the side stream joins first
        if ops.SIDE_ENABLED:
            ops.join_side()
        ps = list(module.parameters())
        GRAD_NOTIFY(ps, [G.out.get(p) for p in ps])


class RowPlan:
    """The row maps of one training step, built and read back TOGETHER at the start of model.forward (one device->host
    synchronisation per step, taken while the GPU is still finishing the previous step -- the host then runs ahead for the
    whole step instead of stalling at the decoder's entry):
      enc_keys  ops.KeyRows of the encoder's key-padding mask  -> K | V of the encoder self-attention on visible rows
      dec_keys  ops.KeyRows of the decoder's memory mask       -> cross-attention K | V on visible rows
      live      ops.LiveRows of the rows that reach the loss   -> decoder forward / backward on those rows
    Each is None when its shortcut does not apply (switch off, mask not a prefix, too little to gain)."""

    def __init__(self, enc_keys=None, dec_keys=None, live=None):
        self.enc_keys, self.dec_keys, self.live = enc_keys, dec_keys, live

    @staticmethod
    def usable_keys(kr, rows):
        if kr is None:
            return None
        h = kr.host()
        ok = h["nonprefix"] == 0 and h["empty"] == 0 and 0 < h["padded"] <= COMPACT_MAX_FRACTION * rows
        return kr if ok else None

    @staticmethod
    def usable_live(lr, rows):
        if lr is None:
            return None
        h = lr.host()
        ok = h["violations"] == 0 and h["nonprefix"] == 0 and 0 < h["padded"] <= COMPACT_MAX_FRACTION * rows
        return lr if ok else None

    @classmethod
    def build(cls, src_mask_u8, trg_mask_u8, loss_rows, B, Le, T, nc_lat, n_layers_enc, n_layers_dec):
        """src_mask_u8 [B, Le] (encoder keys, condition rows included); the decoder's memory mask is the same with nc_lat
        visible condition rows in front (use_cond2lat); loss_rows uint8 [B, T] or None."""
        return cls.launch(src_mask_u8, trg_mask_u8, loss_rows, B, Le, T, nc_lat, n_layers_enc, n_layers_dec).finish()

    @classmethod
    def launch(cls, src_mask_u8, trg_mask_u8, loss_rows, B, Le, T, nc_lat, n_layers_enc, n_layers_dec):
        """build() in two halves: the map kernels and ONE asynchronous read-back are queued here, .finish() of the
        returned object waits for that read-back alone and returns the plan.  Queued a step ahead (between a step's
        forward and its backward: Model/forward_propagation1.prefetch), the maps of the NEXT batch reach the host while
        this step's backward runs, and the next forward starts without the host ever waiting for the device."""
        if torch.cuda.is_current_stream_capturing() or src_mask_u8 is None or src_mask_u8.numel() != B * Le:
            return _PendingPlan(None, None, None, None, 0, 0, 0)
        sm = src_mask_u8.view(B, Le)
        ek = ops.KeyRows(sm, B, Le) if (COMPACT_ENC_KV and n_layers_enc > 0) else None
        dk = None
        if COMPACT_KV and n_layers_dec > 0:
            if nc_lat > 0:
                ones = torch.ones(B, nc_lat, dtype=torch.uint8, device=sm.device)
                dk = ops.KeyRows(torch.cat([ones, sm], dim=1).contiguous(), B, Le + nc_lat)
            elif ek is not None:
                dk = ek                                  # the same mask: one map serves both trunks
            else:
                dk = ops.KeyRows(sm, B, Le)
        lr = None
        if (COMPACT_FWD and loss_rows is not None and trg_mask_u8 is not None and T <= 96 and Le + nc_lat <= 96
                and n_layers_dec > 0):
            lr = ops.LiveRows.from_rows(loss_rows.reshape(B, T), B, T, trg_mask_u8)
        return _PendingPlan(ek, dk, lr, ops.PendingReadBack(ek, dk, lr),   # ONE read-back for all three
                            B * Le, B * (Le + nc_lat), B * T)


class _PendingPlan:
    def __init__(self, ek, dk, lr, pending, rows_e, rows_d, rows_t):
        self.ek, self.dk, self.lr, self.pending = ek, dk, lr, pending
        self.rows = (rows_e, rows_d, rows_t)

    def finish(self) -> "RowPlan":
        if self.pending is None:
            return RowPlan()
        self.pending.finish()                            # the step's ONE host synchronisation
        plan = RowPlan(RowPlan.usable_keys(self.ek, self.rows[0]), RowPlan.usable_keys(self.dk, self.rows[1]),
                       RowPlan.usable_live(self.lr, self.rows[2]))
        if plan.live is not None:
            plan.live.fwd = True
        return plan


def next_seed() -> int:
    """Fresh 64-bit dropout/eps seed per trunk call, derived from torch.manual_seed()."""
    if _SEED["base"] != torch.initial_seed():
        _SEED["base"] = torch.initial_seed()
        _SEED["ctr"] = 0
    _SEED["ctr"] += 1
    return (_SEED["base"] * 0x9E3779B97F4A7C15 + _SEED["ctr"] * 0xD1B54A32D192ED03) & ((1 << 64) - 1)


class Run:
    """Per-call dropout context: probability, seed and a site counter (every dropout
    application in the trunk gets its own Philox stream)."""

    def __init__(self, p: float, training: bool):
        self.p = float(p) if training else 0.0
        self.seed = next_seed() if self.p > 0 else 0
        self._site = 0
        # decoder backward only: (list, count) of the 32-row token tiles whose incoming gradient is not
        # identically zero -- the weight-gradient GEMMs over decoder rows reduce over these tiles only
        self.kt = None

    def site(self) -> int:
        self._site += 1
        return self._site


class GradSink:
    """Where parameter gradients are written.  Parameters of a flattened model carry
    `_gct_gview` (a view into the model's flat gradient buffer); the kernels write straight
    into it and autograd adopts the view as .grad (zero copies).  If .grad already aliases
    that view (no zero_grad(set_to_none=True) since the last backward), or the slot has already been handed
    to another Function of the same backward pass (model used twice in one graph: flat.HANDED_SLOTS), a
    temporary is used so accumulation semantics stay correct."""

    def __init__(self):
        self.out = {}

    def __call__(self, p: torch.nn.Parameter) -> torch.Tensor:
        t = self.out.get(p)
        if t is None:
            v = getattr(p, "_gct_gview", None)
            if v is not None and id(p) not in HANDED_SLOTS and \
                    (p.grad is None or p.grad.data_ptr() != v.data_ptr()):
                # a FRESH view object: AccumulateGrad only adopts (instead of cloning) a
                # gradient tensor nobody else holds a reference to
                t = v.view(v.shape)
                HANDED_SLOTS.add(id(p))
            else:
                t = torch.empty_like(p)
            self.out[p] = t
        return t

    def collect(self, params):
        ops.join_side()      # weight gradients were produced on the side stream (ops.linear_wgrad)
        return tuple(self.out.get(p) for p in params)


def _empty(rows, cols, like):
    return torch.empty(rows, cols, dtype=torch.float32, device=like.device)


# ------------------------------------------------------------------------------------- MHA
def mha_fwd(run: Run, m, xq, xkv, B, Lq, Lk, mask_u8, resid, want_probs=False, keys=None, live=None):
    """m: MultiHeadAttention module (q_linear, k_linear, v_linear, out).  xq [B*Lq,d],
    xkv [B*Lk,d] (the same tensor object for self-attention).  With `resid` the output
    projection fuses  resid + dropout(.)  (the layer's dropout_k + residual add).
    live (ops.LiveRows with .fwd): xq / resid / the result hold the compact query rows only (and xkv too for
    self-attention: the live rows of a sample are a prefix, so they are also the only keys a live query can see)."""
    d, H = m.d_model, m.h
    dk = d // H
    self_attn = xkv is xq
    Mq = xq.shape[0]
    # Self-attention under a key-padding mask (the encoder): the K / V projections of masked rows are never read and
    # their dK / dV are exactly zero, so K | V run on the VISIBLE rows only (keys: ops.KeyRows of the mask; 56 % fewer
    # rows at MOSES-like lengths) while Q -- every row's output feeds the KL term -- runs on all of them.  From here on
    # it is the cross-attention path with xkv = the gathered visible rows of xq.
    self_split = self_attn and keys is not None and live is None and not want_probs
    if self_split:
        xkv = keys.gather(xq)
    if self_attn and not self_split:
        qkv = _empty(Mq, 3 * d, xq)
        ops.linear_fwd(xq, [m.q_linear.weight, m.k_linear.weight, m.v_linear.weight],
                       [m.q_linear.bias, m.k_linear.bias, m.v_linear.bias],
                       [qkv, qkv[:, d:], qkv[:, 2 * d:]], 3 * d)
        q, k, v, ldq, ldkv = qkv, qkv[:, d:], qkv[:, 2 * d:], 3 * d, 3 * d
        qb, kvb = qkv, None
        if live is not None:
            keys = live                                 # K / V rows of sample b: cstart[b] .. + n_b[b]
    else:
        qb = _empty(Mq, d, xq)
        ops.linear_fwd(xq, [m.q_linear.weight], [m.q_linear.bias], [qb], d)
        # keys (ops.KeyRows): xkv holds the VISIBLE rows of the encoder memory only (quad-compacted)
        kvb = _empty(B * Lk, 2 * d, xkv) if keys is None else keys.empty(2 * d)
        ops.linear_fwd(xkv, [m.k_linear.weight, m.v_linear.weight],
                       [m.k_linear.bias, m.v_linear.bias], [kvb, kvb[:, d:]], 2 * d)
        q, k, v, ldq, ldkv = qb, kvb, kvb[:, d:], d, 2 * d
    site_p = run.site()
    o, lse, probs = ops.attn_fwd(q, k, v, ldq, ldkv, ldkv, mask_u8, B, H, Lq, Lk, dk, run.p,
                                 run.seed, site_p, want_probs=want_probs, keys=keys, live=live)
    y = _empty(Mq, d, xq)
    site_o = run.site()
    if resid is not None:
        ops.linear_fwd(o, [m.out.weight], [m.out.bias], [y], d, epi=ops.EPI_DROP_RESID,
                       resid=resid, p=run.p, seed=run.seed, site=site_o, live=live)
    else:
        ops.linear_fwd(o, [m.out.weight], [m.out.bias], [y], d)
    saved = (xq, xkv, qb, kvb, o, lse, mask_u8, B, Lq, Lk, site_p, site_o, resid is not None,
             None if (self_attn and live is not None) else keys, self_split)
    return y, saved, probs


def mha_bwd(run: Run, m, saved, dy, G: GradSink, dxq_out, depi_q, dxkv_out=None,
            depi_kv=ops.DEPI_STORE, live=None, gdrop=None):
    """dy: gradient w.r.t. the block output (resid + dropout(out(o)) or out(o)); gdrop: dropout_bwd(dy) already
    computed by the producer of dy (ops.norm_bwd(drop=...)).
    Writes dW/db through G, d(xq) into dxq_out (epilogue depi_q) and, for cross-attention,
    d(xkv) into dxkv_out (epilogue depi_kv).  The identity path to `resid` is the caller's.
    live (ops.LiveRows): the QUERY-side rows (dy, dxq_out) are quad-compacted; saved forward tensors are
    gathered on the way in, key/value-side gradients of cross-attention stay in the encoder's row space."""
    xq, xkv, qb, kvb, o, lse, mask_u8, B, Lq, Lk, site_p, site_o, fused, keys, self_split = saved
    d, H = m.d_model, m.h
    dk = d // H
    Mq, Mk = B * Lq, xkv.shape[0]
    kt = run.kt
    new = lambda cols: _empty(Mq, cols, dy)                                      # noqa: E731
    if live is not None:
        Mq, kt = live.Mc, None
        # rows of a live quad that belong to no sample's live prefix are never written by the attention kernel
        new = live.empty_zero_gaps
        # (a forward that ran on the compact rows saved compact activations: nothing to gather)
        o_in, xq_in = (o, xq) if live.fwd else (live.gather(o), live.gather(xq))
    else:
        o_in, xq_in = o, xq
    if gdrop is not None:
        g = gdrop
    else:
        g = ops.dropout_bwd(dy, run.p, run.seed, site_o, live=live) if (fused and run.p > 0) else dy
    ops.linear_wgrad([g], d, o_in, [G(m.out.weight)], [G(m.out.bias)], kt=kt)
    do = _empty(Mq, d, dy) if live is None else live.empty(d)
    ops.linear_dgrad([g], d, Mq, [m.out.weight], do)
    if kvb is None:  # self-attention: fused [q|k|v]
        dqkv = new(3 * d)
        fwdc = live is not None and live.fwd             # q | k | v and o are compact too: address K / V like dK / dV
        ops.attn_bwd(qb, qb[:, d:], qb[:, 2 * d:], 3 * d, 3 * d, 3 * d, mask_u8, o, do, lse, dqkv,
                     dqkv[:, d:], dqkv[:, 2 * d:], 3 * d, 3 * d, 3 * d, B, H, Lq, Lk, dk, run.p,
                     run.seed, site_p, live=live, kv_compact=live is not None and not fwdc,
                     keys=live if fwdc else None)
        segs = [dqkv, dqkv[:, d:], dqkv[:, 2 * d:]]
        ops.linear_wgrad(segs, 3 * d, xq_in,
                         [G(m.q_linear.weight), G(m.k_linear.weight), G(m.v_linear.weight)],
                         [G(m.q_linear.bias), G(m.k_linear.bias), G(m.v_linear.bias)], kt=kt)
        ops.linear_dgrad(segs, 3 * d, Mq,
                         [m.q_linear.weight, m.k_linear.weight, m.v_linear.weight], dxq_out,
                         depi=depi_q)
    else:
        dq = new(d)
        if keys is None:
            dkv = _empty(Mk, 2 * d, dy)
        else:      # compacted keys: the rows that pad a quad belong to no sample and are never written
            dkv = keys.empty_zero_gaps(2 * d)
        ops.attn_bwd(qb, kvb, kvb[:, d:], d, 2 * d, 2 * d, mask_u8, o, do, lse, dq, dkv, dkv[:, d:],
                     d, 2 * d, 2 * d, B, H, Lq, Lk, dk, run.p, run.seed, site_p, live=live, keys=keys)
        ops.linear_wgrad([dq], d, xq_in, [G(m.q_linear.weight)], [G(m.q_linear.bias)], kt=kt)   # query rows
        ops.linear_wgrad([dkv, dkv[:, d:]], 2 * d, xkv,
                         [G(m.k_linear.weight), G(m.v_linear.weight)],
                         [G(m.k_linear.bias), G(m.v_linear.bias)])
        ops.linear_dgrad([dq], d, Mq, [m.q_linear.weight], dxq_out, depi=depi_q)
        if self_split:       # K | V came from the visible rows of xq itself: their input gradient goes back into d(xq)
            dxk = keys.empty(d)
            ops.linear_dgrad([dkv, dkv[:, d:]], 2 * d, Mk, [m.k_linear.weight, m.v_linear.weight], dxk)
            keys.scatter_add(dxk, dxq_out)
        elif dxkv_out is not None:
            ops.linear_dgrad([dkv, dkv[:, d:]], 2 * d, Mk, [m.k_linear.weight, m.v_linear.weight],
                             dxkv_out, depi=depi_kv)


# ------------------------------------------------------------------------------------- FFN
def ffn_fwd(run: Run, ff, x, resid, live=None):
    M, d = x.shape
    dff = ff.linear_1.weight.shape[0]
    pre = _empty(M, dff, x)
    hdn = _empty(M, dff, x)
    site_h = run.site()
    ops.linear_fwd(x, [ff.linear_1.weight], [ff.linear_1.bias], [hdn], dff, epi=ops.EPI_GELU_DROP,
                   pre=pre, p=run.p, seed=run.seed, site=site_h, live=live)
    y = _empty(M, d, x)
    site_o = run.site()
    if resid is not None:
        ops.linear_fwd(hdn, [ff.linear_2.weight], [ff.linear_2.bias], [y], d,
                       epi=ops.EPI_DROP_RESID, resid=resid, p=run.p, seed=run.seed, site=site_o, live=live)
    else:
        ops.linear_fwd(hdn, [ff.linear_2.weight], [ff.linear_2.bias], [y], d)
    return y, (x, pre, hdn, site_h, site_o, resid is not None)


def ffn_bwd(run: Run, ff, saved, dy, G: GradSink, dx_out, depi, live=None, gdrop=None):
    x, pre, hdn, site_h, site_o, fused = saved
    M, d = x.shape
    dff = pre.shape[1]
    kt = run.kt
    if live is not None:          # quad-compacted rows: gather what the forward saved
        M, kt = live.Mc, None
        if not live.fwd:                              # (a compact forward saved compact x / pre / hdn)
            x, hdn = live.gather(x), live.gather(hdn)     # (pre stays in place: the GELU-backward epilogue reads it
                                                          #  through the quad map)
    if gdrop is not None:
        g = gdrop
    else:
        g = ops.dropout_bwd(dy, run.p, run.seed, site_o, live=live) if (fused and run.p > 0) else dy
    ops.linear_wgrad([g], d, hdn, [G(ff.linear_2.weight)], [G(ff.linear_2.bias)], kt=kt)
    dpre = _empty(M, dff, dy) if live is None else live.empty(dff)
    ops.linear_dgrad([g], d, M, [ff.linear_2.weight], dpre, depi=ops.DEPI_GELU_BWD, pre=pre,
                     p=run.p, seed=run.seed, site=site_h, live=live, pre_full=live is not None and not live.fwd)
    ops.linear_wgrad([dpre], dff, x, [G(ff.linear_1.weight)], [G(ff.linear_1.bias)], kt=kt)
    ops.linear_dgrad([dpre], dff, M, [ff.linear_1.weight], dx_out, depi=depi)


# ---------------------------------------------------------------------------------- layers
def enc_layer_fwd(run: Run, layer, x_in, B, L, mask_u8, want_probs=False, keys=None):
    n1, m1, r1 = ops.norm_fwd(x_in, layer.norm_1.alpha, layer.norm_1.bias, layer.norm_1.eps)
    a, sv_a, probs = mha_fwd(run, layer.attn, n1, n1, B, L, L, mask_u8, n1, want_probs, keys=keys)
    n2, m2, r2 = ops.norm_fwd(a, layer.norm_2.alpha, layer.norm_2.bias, layer.norm_2.eps)
    out, sv_f = ffn_fwd(run, layer.ff, n2, n2)
    return out, (x_in, m1, r1, sv_a, a, m2, r2, sv_f), probs


def _mha_drop(run: Run, sv, buf):
    """ops.norm_bwd(drop=...) request for the attention block saved in sv: the Norm backward that produces the
    gradient of that block's output also writes dropout_bwd of it (the block's own first step) into buf."""
    return (buf, run.p, run.seed, sv[11]) if (buf is not None and sv[12] and run.p > 0) else None


def _ffn_drop(run: Run, sv, buf):
    return (buf, run.p, run.seed, sv[4]) if (buf is not None and sv is not None and sv[5] and run.p > 0) else None


def enc_layer_bwd(run: Run, layer, saved, g, G: GradSink, gd=None, gdbuf=None, below=None):
    """g: mutable [M,d] gradient buffer w.r.t. the layer output; returns d(x_in) in g.
    gd: dropout_bwd(g) for this layer's FFN if the caller's Norm backward already produced it; gdbuf: scratch
    [M,d] for the fused dropout_bwd outputs; below: saved FFN state of the layer underneath (its dropout_bwd is
    written by this layer's last Norm backward) -- returns (g, that buffer or None)."""
    x_in, m1, r1, sv_a, a, m2, r2, sv_f = saved
    with ops.deferred_reductions():      # the layer's 6 weight / bias / Norm slab reductions: one launch at the end
        # out = n2 + drop(ffn(n2))  =>  d(n2) = g + ffn'(g): the W1 dgrad accumulates into g
        ffn_bwd(run, layer.ff, sv_f, g, G, g, ops.DEPI_ACCUM, gdrop=gd)
        dr = _mha_drop(run, sv_a, gdbuf)
        ops.norm_bwd(g, a, layer.norm_2.alpha, m2, r2, G(layer.norm_2.alpha), G(layer.norm_2.bias),
                     out=g, eps=layer.norm_2.eps, drop=dr)
        # a = n1 + drop(attn(n1))
        mha_bwd(run, layer.attn, sv_a, g, G, g, ops.DEPI_ACCUM, gdrop=None if dr is None else gdbuf)
        dr = _ffn_drop(run, below, gdbuf)
        ops.norm_bwd(g, x_in, layer.norm_1.alpha, m1, r1, G(layer.norm_1.alpha), G(layer.norm_1.bias),
                     out=g, eps=layer.norm_1.eps, drop=dr)
    _grads_done(layer, G)
    return g, (None if dr is None else gdbuf)


def dec_layer_fwd(run: Run, layer, x, e, B, T, Lk, src_mask_u8, trg_mask_u8, want_probs=False, keys=None, live=None):
    """live (ops.LiveRows with .fwd): x and everything row-wise of this layer hold the compact live rows only."""
    x2, m1, r1 = ops.norm_fwd(x, layer.norm_1.alpha, layer.norm_1.bias, layer.norm_1.eps)
    xa, sv1, p1 = mha_fwd(run, layer.attn_1, x2, x2, B, T, T, trg_mask_u8, x, want_probs, live=live)
    x2, m2, r2 = ops.norm_fwd(xa, layer.norm_2.alpha, layer.norm_2.bias, layer.norm_2.eps)
    xb, sv2, p2 = mha_fwd(run, layer.attn_2, x2, e, B, T, Lk, src_mask_u8, xa, want_probs, keys=keys, live=live)
    x2, m3, r3 = ops.norm_fwd(xb, layer.norm_3.alpha, layer.norm_3.bias, layer.norm_3.eps)
    xc, svf = ffn_fwd(run, layer.ff, x2, xb, live=live)
    return xc, (x, m1, r1, sv1, xa, m2, r2, sv2, xb, m3, r3, svf), p1, p2


def dec_layer_bwd(run: Run, layer, saved, g, de, first_de, G: GradSink, live=None, gd=None, gdbuf=None, below=None):
    """live (ops.LiveRows): g and every row-wise gradient of this layer are quad-compacted [live.Mc, d].
    gd / gdbuf / below as in enc_layer_bwd; returns (g, dropout_bwd(g) for the FFN of the layer underneath or None)."""
    x, m1, r1, sv1, xa, m2, r2, sv2, xb, m3, r3, svf = saved
    t = torch.empty_like(g) if live is None else live.empty(g.shape[1])
    with ops.deferred_reductions():      # the layer's 9 weight / bias / Norm slab reductions: one launch at the end
        ffn_bwd(run, layer.ff, svf, g, G, t, ops.DEPI_STORE, live=live, gdrop=gd)
        dr = _mha_drop(run, sv2, gdbuf)
        ops.norm_bwd(t, xb, layer.norm_3.alpha, m3, r3, G(layer.norm_3.alpha), G(layer.norm_3.bias),
                     dres=g, out=g, eps=layer.norm_3.eps, live=live, drop=dr)
        mha_bwd(run, layer.attn_2, sv2, g, G, t, ops.DEPI_STORE, de,
                ops.DEPI_STORE if first_de else ops.DEPI_ACCUM, live=live, gdrop=None if dr is None else gdbuf)
        dr = _mha_drop(run, sv1, gdbuf)
        ops.norm_bwd(t, xa, layer.norm_2.alpha, m2, r2, G(layer.norm_2.alpha), G(layer.norm_2.bias),
                     dres=g, out=g, eps=layer.norm_2.eps, live=live, drop=dr)
        mha_bwd(run, layer.attn_1, sv1, g, G, t, ops.DEPI_STORE, live=live, gdrop=None if dr is None else gdbuf)
        dr = _ffn_drop(run, below, gdbuf)
        ops.norm_bwd(t, x, layer.norm_1.alpha, m1, r1, G(layer.norm_1.alpha), G(layer.norm_1.bias),
                     dres=g, out=g, eps=layer.norm_1.eps, live=live, drop=dr)
    _grads_done(layer, G)
    return g, (None if dr is None else gdbuf)


# ---------------------------------------------------------------------------------- trunks
def _pe2d(pe_mod, L):
    pe = pe_mod.pe
    if L > pe.shape[1]:
        raise ValueError(f"sequence length {L} exceeds the positional table ({pe.shape[1]})")
    return pe[0]


def encoder_trunk_fwd(enc, run: Run, src, mask_u8, econds, want_probs=False, keys=None):
    """Model/vaetf.py:32-54 (up to the final Norm).  Returns x [B, n_c+S, d] and saved state.
    keys (ops.KeyRows of the key-padding mask, usable(): checked by the caller): the K | V projections of every layer
    run on the visible rows only (mha_fwd)."""
    B, S = src.shape
    d, nc = enc.d_model, enc.nconds
    cond = None
    if nc > 0:
        if econds is None:
            raise ValueError("econds required when nconds > 0")          # vaetf.py:36
        cond = ops.small_linear_fwd(econds.contiguous(), enc.embed_cond2enc.weight,
                                    enc.embed_cond2enc.bias)
    L = S + nc
    mask_u8 = ops.pack_mask(mask_u8, B, L, L)           # bits: once per call, shared by all layers / heads / backward
    site_pe = run.site()
    x = ops.embed_pe_fwd(src, enc.embed_sentence.embed.weight, cond, _pe2d(enc.pe, L), nc,
                         math.sqrt(d), run.p, run.seed, site_pe)
    lsv, probs = [], []
    for layer in enc.layers:
        x, sv, pr = enc_layer_fwd(run, layer, x, B, L, mask_u8, want_probs, keys=keys)
        lsv.append(sv)
        probs.append(pr)
    y, mean, rstd = ops.norm_fwd(x, enc.norm.alpha, enc.norm.bias, enc.norm.eps)
    return y.view(B, L, d), (src, econds, site_pe, lsv, x, mean, rstd, B, L), probs


def encoder_trunk_bwd(enc, run: Run, saved, dy, G: GradSink):
    src, econds, site_pe, lsv, x_last, mean, rstd, B, L = saved
    d, nc = enc.d_model, enc.nconds
    g = dy.reshape(B * L, d).clone()
    # every Norm backward also writes dropout_bwd of its result for the sub-layer that consumes it next (gdbuf)
    gdbuf = torch.empty_like(g) if (run.p > 0 and len(lsv) > 0) else None
    dr = _ffn_drop(run, lsv[-1][-1] if lsv else None, gdbuf)
    ops.norm_bwd(g, x_last, enc.norm.alpha, mean, rstd, G(enc.norm.alpha), G(enc.norm.bias), out=g,
                 eps=enc.norm.eps, drop=dr)
    gd = None if dr is None else gdbuf
    for i in range(len(lsv) - 1, -1, -1):
        g, gd = enc_layer_bwd(run, enc.layers[i], lsv[i], g, G, gd=gd, gdbuf=gdbuf,
                              below=lsv[i - 1][-1] if i > 0 else None)
    dcond = torch.empty(B, nc * d, dtype=torch.float32, device=g.device) if nc > 0 else None
    ops.embed_pe_bwd(g, src, G(enc.embed_sentence.embed.weight), dcond, nc, math.sqrt(d), run.p,
                     run.seed, site_pe)
    if nc > 0:
        ops.small_linear_bwd(dcond, econds.contiguous(), G(enc.embed_cond2enc.weight),
                             G(enc.embed_cond2enc.bias))


def decoder_trunk_fwd(dec, run: Run, trg, z, src_mask_u8, trg_mask_u8, dconds, want_probs=False, loss_rows=None,
                      plan=None):
    """Model/vaetf.py:79-114.  z [B, L_e, latent].
    loss_rows (uint8 [B, T], optional): the decoder rows whose output reaches the loss (the reference's cross-entropy
    ignores the rows of padded targets, Train/trainer1.py:21-22: 56 % of the rows at MOSES-like lengths).  A row outside
    the set influences a row inside it only as a KEY of self-attention, so when no live query can see a dead row under
    THIS call's trg_mask (checked on the device, as for the compacted backward) the whole trunk runs on the quad-
    compacted live rows: every GEMM, Norm and attention launch sees ~half the rows, every dropout site draws the bits of
    the original coordinates, the backward finds its activations compact already.  When the shortcut is taken the trunk
    RETURNS THE COMPACT ROWS [Mc, d] and the map (saved[-1]); whoever scatters them back (ScatterRowsFn: Decoder.forward
    for a caller of the decoder alone, Vaetf / Cvaetf.forward behind the vocabulary head) produces zeros on the skipped
    rows -- or, for the few padded rows that share an aligned group of four rows with a live one, finite values without
    meaning -- not the reference's values (nothing downstream of an ignore_index loss reads them), and reports a
    gradient that arrives on a skipped row as an error (ops.LiveRows.check_grad)."""
    B, T0 = trg.shape
    d, nc = dec.d_model, dec.nconds
    Le, lat = z.shape[1], z.shape[2]
    z2 = z.reshape(B * Le, lat)
    c2d = dec.use_cond2dec and nc > 0
    c2l = (not c2d) and dec.use_cond2lat and nc > 0
    cond_x = None
    if c2d:
        cond_x = ops.small_linear_fwd(dconds.contiguous(), dec.embed_cond2dec.weight,
                                      dec.embed_cond2dec.bias)
    T = T0 + (nc if c2d else 0)
    site_pe = run.site()
    x = ops.embed_pe_fwd(trg, dec.embed.embed.weight, cond_x, _pe2d(dec.pe, T), nc if c2d else 0,
                         math.sqrt(d), run.p, run.seed, site_pe)
    ez = _empty(B * Le, d, z2)
    ops.linear_fwd(z2, [dec.fc_z.weight], [dec.fc_z.bias], [ez], d)
    Lk = Le
    e = ez
    if c2l:
        Lk = Le + nc
        cl = ops.small_linear_fwd(dconds.contiguous(), dec.embed_cond2lat.weight,
                                  dec.embed_cond2lat.bias)                     # [B, nc*d]
        e = _empty(B * Lk, d, z2)
        ops.copy_rows(cl, nc, 0, e, Lk, 0, B * nc, nc, d)
        ops.copy_rows(ez, Le, 0, e, Lk, nc, B * Le, Le, d)
        if src_mask_u8 is not None:                                            # vaetf.py:95-98
            ones = torch.ones(B, nc, dtype=torch.uint8, device=src_mask_u8.device)
            src_mask_u8 = torch.cat([ones, src_mask_u8.view(B, Le)], dim=1).contiguous()
    lsv, p1s, p2s = [], [], []
    capturing = torch.cuda.is_current_stream_capturing()
    # Row maps.  `plan` (engine.RowPlan, built and read back at the start of model.forward): use what it validated.
    # Without one (the decoder called on its own, decode prefill) they are built here -- their kernels are queued before
    # either is read, so both come back in one synchronisation:
    #   live  the rows that reach the loss (loss_rows): forward and backward of the trunk run on them only;
    #   keys  the visible rows of the encoder memory: padded rows are masked keys of every cross-attention, their K / V
    #         projections are never used and their dK / dV are zero, so the six K|V GEMMs, their weight gradients and
    #         the gradient w.r.t. the memory run on the visible rows only (quad-compacted, ops.KeyRows).
    live, keys = None, None
    fwd_ok = (not c2d and not want_probs and T <= 96 and Lk <= 96 and len(dec.layers) > 0 and not capturing
              and trg_mask_u8 is not None)
    if plan is not None:
        live = plan.live if fwd_ok else None
        keys = plan.dec_keys if (src_mask_u8 is not None and src_mask_u8.numel() == B * Lk) else None
    else:
        lr = kr = None
        if COMPACT_FWD and loss_rows is not None and fwd_ok:
            tm_u8 = trg_mask_u8.u8 if isinstance(trg_mask_u8, ops.MaskBits) else trg_mask_u8
            lr = ops.LiveRows.from_rows(loss_rows.reshape(B, T), B, T, tm_u8)
        if (COMPACT_KV and src_mask_u8 is not None and src_mask_u8.numel() == B * Lk and len(dec.layers) > 0
                and not capturing):
            kr = ops.KeyRows(src_mask_u8.view(B, Lk), B, Lk)
        ops.read_back(lr, kr)
        live = RowPlan.usable_live(lr, B * T)
        keys = RowPlan.usable_keys(kr, B * Lk)
        if live is not None:
            live.fwd = True
    if live is not None:
        x = live.gather(x)
    if keys is not None:
        e = keys.gather(e)
    src_m = ops.pack_mask(src_mask_u8, B, T, Lk)
    trg_m = ops.pack_mask(trg_mask_u8, B, T, T)
    for layer in dec.layers:
        x, sv, p1, p2 = dec_layer_fwd(run, layer, x, e, B, T, Lk, src_m, trg_m, want_probs, keys=keys, live=live)
        lsv.append(sv)
        p1s.append(p1)
        p2s.append(p2)
    y, mean, rstd = ops.norm_fwd(x, dec.norm.alpha, dec.norm.bias, dec.norm.eps)
    saved = (trg, z2, dconds, site_pe, lsv, x, mean, rstd, B, T, Le, Lk, c2d, c2l,
             trg_mask_u8.u8 if isinstance(trg_mask_u8, ops.MaskBits) else trg_mask_u8, keys, live)
    if live is not None:
        return y, saved, p1s, p2s        # [Mc, d] COMPACT rows: the caller owns `saved[-1]` (the map) and scatters what it needs
    return y.view(B, T, d), saved, p1s, p2s


def decoder_trunk_bwd(dec, run: Run, saved, dy, G: GradSink, need_dz=True):
    trg, z2, dconds, site_pe, lsv, x_last, mean, rstd, B, T, Le, Lk, c2d, c2l, trg_mask_u8, keys, live_fwd = saved
    new_de = (lambda: _empty(B * Lk, d, dy)) if keys is None else (lambda: keys.empty(d))     # d(memory), maybe compact
    d, nc = dec.d_model, dec.nconds
    # A decoder row whose incoming gradient is zero (padded target positions under the ignore_index loss: 56 % of
    # the rows at MOSES-like lengths) keeps a zero gradient through every layer below -- norm, linear, GELU and
    # dropout backward map a zero row to a zero row, attention gives zero dQ rows for zero dO rows -- PROVIDED no live
    # query attends to it (a dead row that is a visible key receives dK / dV).  ops.LiveRows derives the live rows
    # from the gradient and checks that proviso on the device against the trg_mask this call used
    # (csrc/liverows.hip).  When it holds (and pays), the whole decoder backward runs on the QUAD-COMPACTED live
    # rows: every GEMM, norm and dropout backward sees ~half the rows; otherwise the dense path runs, with the
    # weight-gradient GEMMs reducing over the live token tiles (the list names every tile when the check fails).
    live = None
    if live_fwd is not None:
        # the forward ran on these rows only and handed out COMPACT rows: dy arrives compact (ScatterRowsFn gathers the
        # gradient of what it scattered and reports gradient rows that fell on skipped rows)
        live = live_fwd
        lr = None
        g = dy.reshape(live.Mc, d)
    else:
        g = dy.reshape(B * T, d)
        lr = ops.LiveRows(g, B, T, trg_mask_u8) if (COMPACT_BWD or (B * T) % 32 == 0) else None
    if live is None and COMPACT_BWD and lr is not None and len(dec.layers) > 0 and not torch.cuda.is_current_stream_capturing():
        h = lr.host()                                   # one 32-byte read-back per step
        if h["violations"] == 0 and h["nonprefix"] == 0 and 0 < h["padded"] <= COMPACT_MAX_FRACTION * B * T:
            live = lr
    if live is not None:
        gc = g.clone() if live_fwd is not None else live.gather(g)
        gdbuf = live.empty(d) if run.p > 0 else None
        dr = _ffn_drop(run, lsv[-1][-1], gdbuf)
        ops.norm_bwd(gc, x_last, dec.norm.alpha, mean, rstd, G(dec.norm.alpha), G(dec.norm.bias), out=gc,
                     eps=dec.norm.eps, live=live, drop=dr)
        gd = None if dr is None else gdbuf
        de = new_de()
        for i in range(len(lsv) - 1, -1, -1):
            gc, gd = dec_layer_bwd(run, dec.layers[i], lsv[i], gc, de, i == len(lsv) - 1, G, live=live, gd=gd,
                                   gdbuf=gdbuf, below=lsv[i - 1][-1] if i > 0 else None)
        g = live.scatter(gc)                            # back to [B*T, d]: zero rows where nothing was live
    else:
        g = g.clone()
        run.kt = lr.kt if (lr is not None and (B * T) % 32 == 0) else None
        gdbuf = torch.empty_like(g) if (run.p > 0 and len(lsv) > 0) else None
        dr = _ffn_drop(run, lsv[-1][-1] if lsv else None, gdbuf)
        ops.norm_bwd(g, x_last, dec.norm.alpha, mean, rstd, G(dec.norm.alpha), G(dec.norm.bias), out=g,
                     eps=dec.norm.eps, drop=dr)
        gd = None if dr is None else gdbuf
        de = new_de()
        for i in range(len(lsv) - 1, -1, -1):
            g, gd = dec_layer_bwd(run, dec.layers[i], lsv[i], g, de, i == len(lsv) - 1, G, gd=gd, gdbuf=gdbuf,
                                  below=lsv[i - 1][-1] if i > 0 else None)
        run.kt = None
    if len(dec.layers) == 0:
        de.zero_()
    if keys is not None:
        de = keys.scatter(de)                            # back to [B*Lk, d]: masked keys get no gradient
    # embedding side
    dcx = torch.empty(B, nc * d, dtype=torch.float32, device=g.device) if c2d else None
    ops.embed_pe_bwd(g, trg, G(dec.embed.embed.weight), dcx, nc if c2d else 0, math.sqrt(d), run.p,
                     run.seed, site_pe)
    if c2d:
        ops.small_linear_bwd(dcx, dconds.contiguous(), G(dec.embed_cond2dec.weight),
                             G(dec.embed_cond2dec.bias))
    # memory side
    dez = de
    if c2l:
        dcl = torch.empty(B, nc * d, dtype=torch.float32, device=g.device)
        ops.copy_rows(de, Lk, 0, dcl, nc, 0, B * nc, nc, d)
        ops.small_linear_bwd(dcl, dconds.contiguous(), G(dec.embed_cond2lat.weight),
                             G(dec.embed_cond2lat.bias))
        dez = _empty(B * Le, d, g)
        ops.copy_rows(de, Lk, nc, dez, Le, 0, B * Le, Le, d)
    ops.linear_wgrad([dez], d, z2, [G(dec.fc_z.weight)], [G(dec.fc_z.bias)])
    dz = None
    if need_dz:
        dz = _empty(B * Le, z2.shape[1], g)
        ops.linear_dgrad([dez], d, B * Le, [dec.fc_z.weight], dz)
        dz = dz.view(B, Le, -1)
    return dz


# ------------------------------------------------------------------------ autograd boundary
def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


class EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, run, src, mask_u8, econds, want_probs, keys, *params):
        y, saved, probs = encoder_trunk_fwd(enc, run, src, mask_u8, econds, want_probs, keys=keys)
        ctx.enc, ctx.run, ctx.saved, ctx.params = enc, run, saved, params
        if want_probs:
            for p in probs:
                ctx.mark_non_differentiable(p)
            return (y, *probs)
        return y

    @staticmethod
    def backward(ctx, dy, *unused):
        G = GradSink()
        encoder_trunk_bwd(ctx.enc, ctx.run, ctx.saved, _f32c(dy), G)
        ctx.saved = None
        return (None,) * 7 + G.collect(ctx.params)


class DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dec, run, trg, z, src_mask_u8, trg_mask_u8, dconds, want_probs, loss_rows, plan, *params):
        y, saved, p1, p2 = decoder_trunk_fwd(dec, run, trg, _f32c(z), src_mask_u8, trg_mask_u8,
                                             dconds, want_probs, loss_rows=loss_rows, plan=plan)
        ctx.dec, ctx.run, ctx.saved, ctx.params = dec, run, saved, params
        dec._gct_live_out = saved[-1]          # not None: y holds the COMPACT live rows [Mc, d] (Decoder.forward hands
                                               # the map to its caller, who scatters what it needs: ScatterRowsFn)
        ctx.need_dz = z.requires_grad
        if want_probs:
            for p in p1 + p2:
                ctx.mark_non_differentiable(p)
            return (y, *p1, *p2)
        return y

    @staticmethod
    def backward(ctx, dy, *unused):
        G = GradSink()
        dz = decoder_trunk_bwd(ctx.dec, ctx.run, ctx.saved, _f32c(dy), G, ctx.need_dz)
        ctx.saved = None
        return (None, None, None, dz, None, None, None, None, None, None) + G.collect(ctx.params)


class ScatterRowsFn(torch.autograd.Function):
    """Compact rows [Mc, cols] of a decoder forward that ran on the loss rows only -> all [B, T, cols] rows (zeros where
    nothing was computed).  Backward: the gradient's compact rows; gradient rows that fall on SKIPPED rows cannot be
    honoured -- they are counted on the device and raised at the next read-back (ops.LiveRows.check_grad)."""

    @staticmethod
    def forward(ctx, xc, live, B, T):
        ctx.live, ctx.shape = live, xc.shape
        return live.scatter(_f32c(xc)).view(B, T, xc.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        live = ctx.live
        g = _f32c(dy).reshape(live.M, dy.shape[-1])
        live.check_grad(g)
        return live.gather(g).view(ctx.shape), None, None, None


class SamplerFn(torch.autograd.Function):
    """mu = fc_mu(x), log_var = fc_log_var(x) as ONE segmented GEMM, then the
    reparameterisation (Model/sublayers.py:22-26, Model/cvaetf.py:55-69)."""

    @staticmethod
    def forward(ctx, x, w_mu, b_mu, w_lv, b_lv, eps, variational):
        B, L, d = x.shape
        lat = w_mu.shape[0]
        x2 = _f32c(x).view(B * L, d)
        mu = _empty(B * L, lat, x2)
        lv = _empty(B * L, lat, x2)
        ops.linear_fwd(x2, [w_mu, w_lv], [b_mu, b_lv], [mu, lv], lat)
        if variational:
            z, eps_used = ops.reparam_fwd(mu, lv, None if eps is None else _f32c(eps).view(B * L, lat),
                                          next_seed(), 0)
        else:
            z, eps_used = mu, None
        ctx.save_for_backward(x2, w_mu, w_lv, lv, eps_used if eps_used is not None else lv)
        ctx.variational = variational
        ctx.shape = (B, L, d, lat)
        ctx.params = (w_mu, b_mu, w_lv, b_lv)
        shp = (B, L, lat)
        if variational:
            return z.view(shp), mu.view(shp), lv.view(shp)
        return mu.view(shp), mu.view(shp).clone(), lv.view(shp)

    @staticmethod
    def backward(ctx, dz, dmu, dlv):
        x2, w_mu, w_lv, lv, eps = ctx.saved_tensors
        B, L, d, lat = ctx.shape
        M = B * L
        zeros = None
        def flat(t):
            return None if t is None else _f32c(t).view(M, lat)
        dz, dmu, dlv = flat(dz), flat(dmu), flat(dlv)
        gmu = _empty(M, lat, x2)
        glv = _empty(M, lat, x2)
        if ctx.variational:
            if dz is None:
                dz = torch.zeros(M, lat, dtype=torch.float32, device=x2.device)
            ops.reparam_bwd(dz, lv, eps, dmu, dlv, gmu, glv)
        else:
            z0 = torch.zeros(M, lat, dtype=torch.float32, device=x2.device)
            a = dz if dz is not None else z0
            b = dmu if dmu is not None else z0
            ops.add(a, b, out=gmu)
            glv = dlv if dlv is not None else z0
        G = GradSink()
        w_mu_p, b_mu_p, w_lv_p, b_lv_p = ctx.params
        ops.linear_wgrad([gmu, glv], lat, x2, [G(w_mu_p), G(w_lv_p)], [G(b_mu_p), G(b_lv_p)])
        dx = _empty(M, d, x2)
        ops.linear_dgrad([gmu, glv], lat, M, [w_mu, w_lv], dx)
        return (dx.view(B, L, d),) + G.collect(ctx.params) + (None, None)


class LinearFn(torch.autograd.Function):
    """y = x W^T + b on the MFMA GEMM (Model/vaetf.py:169 `out`, :172 `prop_fc`)."""

    @staticmethod
    def forward(ctx, x, w, b):
        shp = x.shape
        x2 = _f32c(x).reshape(-1, shp[-1])
        y = _empty(x2.shape[0], w.shape[0], x2)
        ops.linear_fwd(x2, [w], [b], [y], w.shape[0])
        ctx.save_for_backward(x2, w)
        ctx.params = (w, b)
        ctx.shp = shp
        return y.view(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        N = w.shape[0]
        dy2 = _f32c(dy).reshape(-1, N)
        G = GradSink()
        wp, bp = ctx.params
        ops.linear_wgrad([dy2], N, x2, [G(wp)], [G(bp) if bp is not None else None])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _empty(x2.shape[0], x2.shape[1], x2)
            ops.linear_dgrad([dy2], N, x2.shape[0], [w], dx)
            dx = dx.view(ctx.shp)
        ops.join_side()
        return dx, G.out.get(wp), G.out.get(bp) if bp is not None else None


class CrossEntropySumFn(torch.autograd.Function):
    """Train/trainer1.py:21-22."""

    @staticmethod
    def forward(ctx, logits, target, pad_id):
        l2 = _f32c(logits).reshape(-1, logits.shape[-1])
        t = target.reshape(-1).contiguous()
        ctx.save_for_backward(l2, t)
        ctx.pad_id, ctx.shp = int(pad_id), logits.shape
        return ops.ce_fwd(l2, t, int(pad_id))

    @staticmethod
    def backward(ctx, g):
        l2, t = ctx.saved_tensors
        return ops.ce_bwd(l2, t, _f32c(g), ctx.pad_id).view(ctx.shp), None, None


class KldFn(torch.autograd.Function):
    """Train/trainer1.py:23."""

    @staticmethod
    def forward(ctx, mu, log_var):
        m, l = _f32c(mu), _f32c(log_var)
        ctx.save_for_backward(m, l)
        return ops.kld_fwd(m, l)

    @staticmethod
    def backward(ctx, g):
        m, l = ctx.saved_tensors
        dmu, dlv = ops.kld_bwd(m, l, _f32c(g))
        return dmu, dlv


class NormFn(torch.autograd.Function):
    """Standalone Norm module call (Model/modules.py:92-95)."""

    @staticmethod
    def forward(ctx, x, alpha, bias, eps):
        x2 = _f32c(x).reshape(-1, x.shape[-1])
        y, mean, rstd = ops.norm_fwd(x2, alpha, bias, eps)
        ctx.save_for_backward(x2, alpha, mean, rstd)
        ctx.params, ctx.eps, ctx.shp = (alpha, bias), eps, x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, alpha, mean, rstd = ctx.saved_tensors
        G = GradSink()
        a, b = ctx.params
        dx = ops.norm_bwd(_f32c(dy).reshape(x2.shape), x2, alpha, mean, rstd, G(a), G(b), eps=ctx.eps)
        return dx.view(ctx.shp), G.out[a], G.out[b], None


# ----------------------------------------------------- standalone sub-module calls (rare)
class EmbedFn(torch.autograd.Function):
    """Embeddings.forward alone: the K1 kernel with scale 1 and a zero positional table."""

    @staticmethod
    def forward(ctx, tok, table):
        B, S = tok.shape
        d = table.shape[1]
        zeros = torch.zeros(S, d, dtype=torch.float32, device=table.device)
        out = ops.embed_pe_fwd(tok.contiguous(), table, None, zeros, 0, 1.0, 0.0, 0, 0)
        ctx.tok, ctx.param = tok.contiguous(), table
        return out.view(B, S, d)

    @staticmethod
    def backward(ctx, dy):
        G = GradSink()
        d = ctx.param.shape[1]
        ops.embed_pe_bwd(_f32c(dy).reshape(-1, d), ctx.tok, G(ctx.param), None, 0, 1.0, 0.0, 0, 0)
        return None, G.out[ctx.param]


class PosEncFn(torch.autograd.Function):
    """PositionalEncoding.forward alone: x*sqrt(d) + pe[:L], dropout -- the K1 kernel with
    zero token columns (every row is a 'cond' row)."""

    @staticmethod
    def forward(ctx, x, pe, scale, run):
        B, L, d = x.shape
        x2 = _f32c(x).view(B, L * d)
        site = run.site()
        dummy_tok = torch.zeros(B, 0, dtype=torch.int64, device=x.device)
        out = ops.embed_pe_fwd(dummy_tok, None, x2, _pe2d_raw(pe, L), L, scale, run.p, run.seed,
                               site, d=d)
        ctx.cfg = (B, L, d, scale, run, site)
        return out.view(B, L, d)

    @staticmethod
    def backward(ctx, dy):
        B, L, d, scale, run, site = ctx.cfg
        dx = torch.empty(B, L * d, dtype=torch.float32, device=dy.device)
        dummy_tok = torch.zeros(B, 0, dtype=torch.int64, device=dy.device)
        ops.embed_pe_bwd(_f32c(dy).reshape(B * L, d), dummy_tok, None, dx, L, scale, run.p,
                         run.seed, site, d=d)
        return dx.view(B, L, d), None, None, None


def _pe2d_raw(pe, L):
    if L > pe.shape[1]:
        raise ValueError(f"sequence length {L} exceeds the positional table ({pe.shape[1]})")
    return pe[0]


def _flat3(x):
    B, L, d = x.shape
    return _f32c(x).view(B * L, d), B, L, d


class MhaFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m, run, q, kv, mask_u8, want_probs, *params):
        xq, B, Lq, d = _flat3(q)
        if kv is None:
            xkv, Lk = xq, Lq
        else:
            xkv, _, Lk, _ = _flat3(kv)
        y, saved, probs = mha_fwd(run, m, xq, xkv, B, Lq, Lk, mask_u8, None, want_probs)
        ctx.m, ctx.run, ctx.saved, ctx.params = m, run, saved, params
        ctx.shapes = (B, Lq, Lk, d, kv is not None)
        if want_probs:
            ctx.mark_non_differentiable(probs)
            return y.view(B, Lq, d), probs
        return (y.view(B, Lq, d),)

    @staticmethod
    def backward(ctx, dy, *unused):
        B, Lq, Lk, d, cross = ctx.shapes
        G = GradSink()
        dxq = torch.empty(B * Lq, d, dtype=torch.float32, device=dy.device)
        dxkv = torch.empty(B * Lk, d, dtype=torch.float32, device=dy.device) if cross else None
        mha_bwd(ctx.run, ctx.m, ctx.saved, _f32c(dy).reshape(B * Lq, d), G, dxq, ops.DEPI_STORE,
                dxkv, ops.DEPI_STORE)
        ctx.saved = None
        return (None, None, dxq.view(B, Lq, d), dxkv.view(B, Lk, d) if cross else None, None,
                None) + G.collect(ctx.params)


class FfnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ff, run, x, *params):
        x2 = _f32c(x).reshape(-1, x.shape[-1])
        y, saved = ffn_fwd(run, ff, x2, None)
        ctx.ff, ctx.run, ctx.saved, ctx.params, ctx.shp = ff, run, saved, params, x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        G = GradSink()
        dy2 = _f32c(dy).reshape(-1, dy.shape[-1])
        dx = torch.empty_like(dy2)
        ffn_bwd(ctx.run, ctx.ff, ctx.saved, dy2, G, dx, ops.DEPI_STORE)
        ctx.saved = None
        return (None, None, dx.view(ctx.shp)) + G.collect(ctx.params)


class EncLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, layer, run, x, mask_u8, want_probs, *params):
        x2, B, L, d = _flat3(x)
        y, saved, probs = enc_layer_fwd(run, layer, x2, B, L, mask_u8, want_probs)
        ctx.layer, ctx.run, ctx.saved, ctx.params, ctx.shp = layer, run, saved, params, (B, L, d)
        if want_probs:
            ctx.mark_non_differentiable(probs)
            return y.view(B, L, d), probs
        return (y.view(B, L, d),)

    @staticmethod
    def backward(ctx, dy, *unused):
        B, L, d = ctx.shp
        G = GradSink()
        g, _ = enc_layer_bwd(ctx.run, ctx.layer, ctx.saved, _f32c(dy).reshape(B * L, d).clone(), G)
        ctx.saved = None
        return (None, None, g.view(B, L, d), None, None) + G.collect(ctx.params)


class DecLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, layer, run, x, e, src_mask_u8, trg_mask_u8, want_probs, *params):
        x2, B, T, d = _flat3(x)
        e2, _, Lk, _ = _flat3(e)
        y, saved, p1, p2 = dec_layer_fwd(run, layer, x2, e2, B, T, Lk, src_mask_u8, trg_mask_u8,
                                         want_probs)
        ctx.layer, ctx.run, ctx.saved, ctx.params = layer, run, saved, params
        ctx.shp = (B, T, Lk, d)
        if want_probs:
            ctx.mark_non_differentiable(p1, p2)
            return y.view(B, T, d), p1, p2
        return (y.view(B, T, d),)

    @staticmethod
    def backward(ctx, dy, *unused):
        B, T, Lk, d = ctx.shp
        G = GradSink()
        de = torch.empty(B * Lk, d, dtype=torch.float32, device=dy.device)
        g, _ = dec_layer_bwd(ctx.run, ctx.layer, ctx.saved, _f32c(dy).reshape(B * T, d).clone(), de,
                             True, G)
        ctx.saved = None
        return (None, None, g.view(B, T, d), de.view(B, Lk, d), None, None, None) + \
            G.collect(ctx.params)
