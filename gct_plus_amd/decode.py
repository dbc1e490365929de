"""KV-cached autoregressive decoding (BASELINE config 5).

The reference's Sampling.decode (Inference/sampling_tool.py:140-184) re-runs the WHOLE decoder
on ys[:, :i+1] for every generated token -- fc_z(z), all six cross-attention K/V projections
and every earlier position are recomputed 79 times, with one device->host sync per step.
Because the decoder is causal, position j's hidden states depend only on tokens 0..j, so the
same token ids come out of
  * `start`:   once per sequence, the cross-attention operands: the key / value projections of the memory fc_z(z) are
               FOLDED into the query / output projections (`_fold_cross`), so a step attends over the latent rows z
               themselves (gct_attn_decode_z) and no per-layer K / V of the memory is built -- or, with a latent wider
               than 2 d_model / H or `GCT_DECODE_ZATTN=0`, the K/V of all layers projected once;
  * `prefill`: the prefix (<sos>, or <sos> scaffold <sep>; with use_cond2dec the n_c condition tokens in
               front of it, which see each other and the first token -- Model/modules.py:19-26) through ONE ordinary
               decoder forward, whose per-layer self-attention K/V fill the caches;
  * `step`:    a single-token chain per generated token -- embedding row, 6 x [norm, fused QKV GEMM, single-query
               attention that APPENDS the new key/value to the cache, out-proj+residual, cross-attention, FFN],
               norm, vocabulary GEMM, fused softmax + argmax / multinomial that appends the token and updates the
               key-valid and finished flags on the device.
The step reads its position from a DEVICE-side counter (csrc/decode.hip), so one captured graph serves every step
(`use_graphs=True`): no host round trip, no per-position capture.  Work per generated token drops from O(T) decoder
passes to O(1).
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch

from . import engine, ops
from ._lib import check


# rows per step below which the step's GEMMs take the skinny fp32 kernels (panel / 64x64 split-K) instead of the general
# path (bf16x6).  Measured on MI355X: n = 1100 / 1536 / 2048 / 3000 take 2.18 / 2.32 / 2.39 / 3.04 ms per token on the
# general path against 2.27 / 2.46 / 2.58 / 3.74 on the skinny kernels; `GCT_DECODE_SKINNY_BELOW` overrides
SKINNY_BELOW = int(os.environ.get("GCT_DECODE_SKINNY_BELOW", "1024"))
# cross-attention of a step over the latent rows themselves (gct_attn_decode_z) instead of per-layer K / V projections of
# the memory; `GCT_DECODE_ZATTN=0` keeps the projected K / V (the A/B switch; also taken when the latent is too wide)
ZATTN = os.environ.get("GCT_DECODE_ZATTN", "1") != "0"
# replay guard of use_graphs=True (KVDecoder._replay_is_fast): replay must not be slower than REPLAY_SLOW_FACTOR x the
# eager launches of the same step on this box; `GCT_DECODE_GRAPH_GUARD=0` switches the guard off (always replay)
REPLAY_GUARD = os.environ.get("GCT_DECODE_GRAPH_GUARD", "1") != "0"
REPLAY_SLOW_FACTOR = 1.3


class KVDecoder:
    def __init__(self, model, pad_id: int, sos_id: int, eos_id: int):
        dec = model.decoder
        self.model, self.dec = model, dec
        self.pad_id, self.sos_id, self.eos_id = int(pad_id), int(sos_id), int(eos_id)
        self.d = dec.d_model
        self.H = dec.layers[0].attn_1.h
        self.dk = self.d // self.H
        self.c2d = bool(dec.use_cond2dec and dec.nconds > 0)
        self.off = dec.nconds if self.c2d else 0          # cache / positional index of token 0
        self.graphs = {}
        self.graph_replay = True
        self.replay_probe = None                          # numbers of the replay guard (after the first capture)
        self._fold_key = None                             # what the cached folded projections were computed from
        self._shape = None

    # -------------------------------------------------------------------------------------
    @torch.no_grad()
    def start(self, z, src_mask, dconds=None, max_total_len=208, refold=False):
        """z [n, L_e, latent]; src_mask bool [n,1,L_e] (as the reference builds it); max_total_len = longest
        token sequence (prefix + generated) this call may reach.
        refold=True recomputes the folded cross-attention projections even if `model.weights_token()` has not changed:
        the token follows torch in-place operations on the parameters / the flat buffer and FusedAdam's kernel, but NOT
        writes through `p.data` (p.data.copy_ / mul_: a separate version counter) or raw kernels of the caller's own --
        such writers call model.invalidate_weight_planes() (which bumps the token) or pass refold=True here."""
        if refold:
            self._fold_key = None
        dec, d = self.dec, self.d
        dev = z.device
        n, Le, lat = z.shape
        if hasattr(self.model, "refresh_weight_planes"):
            self.model.refresh_weight_planes()         # the prefill GEMMs may take the bf16x6 path
        nc = dec.nconds
        c2l = (not self.c2d) and dec.use_cond2lat and nc > 0
        self.z, self.src_mask_in, self.dconds = z, src_mask, dconds
        self.zattn = bool(ZATTN and lat % 4 == 0 and lat <= 128 and Le <= 256 and self.H * lat <= 2 * d)
        z2 = z.reshape(n * Le, lat).float().contiguous()
        Lk, e = Le, None
        if not self.zattn:
            e = ez = torch.empty(n * Le, d, device=dev)
            ops.linear_fwd(z2, [dec.fc_z.weight], [dec.fc_z.bias], [ez], d)
        sv = ops.to_mask_u8(src_mask).view(n, Le)
        if c2l:
            Lk = Le + nc
            cl = ops.small_linear_fwd(dconds.float().contiguous(), dec.embed_cond2lat.weight,
                                      dec.embed_cond2lat.bias)
            if not self.zattn:
                e = torch.empty(n * Lk, d, device=dev)
                ops.copy_rows(cl, nc, 0, e, Lk, 0, n * nc, nc, d)
                ops.copy_rows(ez, Le, 0, e, Lk, nc, n * Le, Le, d)
            sv = torch.cat([torch.ones(n, nc, dtype=torch.uint8, device=dev), sv], dim=1)
        T = int(max_total_len) + self.off              # cache rows: condition tokens (cond2dec) + tokens
        if T > 256 or Lk > 256:
            raise ValueError("decode lengths above 256 are not supported by gct_attn_decode")
        pe_rows = dec.pe.pe.shape[1]
        if T > pe_rows:
            # the positional table has pe_rows rows (Model/modules.py:116-144: 200); the reference fails loudly past it
            raise ValueError(f"decode: {max_total_len} tokens + {self.off} condition rows exceed the {pe_rows}-row "
                             "positional table")
        shape = (n, Lk, T, str(dev), self.zattn, lat)
        if shape != self._shape:
            # new geometry: new buffers, and the graphs captured against the old ones are dropped with them
            # (they hold raw pointers: replaying them after a reallocation would write freed memory)
            self.graphs = {}
            self.graph_replay = True
            self._shape = shape
            self.n, self.Lk, self.T = n, Lk, T
            self.nz = self.H * lat                     # width of the folded query / latent context (zattn)
            self.nq = (d if c2l else 0) + self.nz
            if self.zattn:
                # folded projections of all layers in ONE flat buffer (its bf16 planes serve the bf16x6 GEMMs)
                per = 2 * self.nq * d
                self.zflat = torch.empty(len(dec.layers) * per, device=dev)
                self._fold_key = None                      # new buffers: nothing folded in them yet
                self.zq_w = [self.zflat[i * per:i * per + self.nq * d].view(self.nq, d) for i in range(len(dec.layers))]
                self.zo_w = [self.zflat[i * per + self.nq * d:(i + 1) * per].view(d, self.nq) for i in range(len(dec.layers))]
                self.zq_b = [torch.empty(self.nq, device=dev) for _ in dec.layers]
                self.zo_b = [torch.empty(d, device=dev) for _ in dec.layers]
                self.ckv = [torch.empty(n * nc, 2 * d, device=dev) if c2l else None for _ in dec.layers]
                self.z3 = torch.empty(n, Le, lat, device=dev)     # captured graphs read the latent rows from here
                self.zplanes = None
                self.cross_kv = None
            else:
                self.cross_kv = [torch.empty(n * Lk, 2 * d, device=dev) for _ in dec.layers]
            self.kc = [torch.empty(n, T, d, device=dev) for _ in dec.layers]
            self.vc = [torch.empty(n, T, d, device=dev) for _ in dec.layers]
            self.valid = torch.zeros(n, T, dtype=torch.uint8, device=dev)
            self.done = torch.zeros(n, dtype=torch.uint8, device=dev)
            self.ys = torch.full((n, T), self.pad_id, dtype=torch.int64, device=dev)
            self.src_valid = torch.empty(n, Lk, dtype=torch.uint8, device=dev)
            self.src_klen = torch.empty(n, dtype=torch.int32, device=dev)  # leading memory rows the cross-attention reads
            self.pos = torch.zeros(1, dtype=torch.int32, device=dev)       # token index the next step consumes
            self.seed = torch.zeros(1, dtype=torch.int64, device=dev)      # multinomial seed of this generate()
            dff = dec.layers[0].ff.linear_1.weight.shape[0]
            V = self.model.out.weight.shape[0]
            need = ops._L().gct_linear_fwd_ws_bytes
            shapes = [(dff, d), (d, 3 * d), (d, dff), (d, d), (d, V)]           # (K, N) of every GEMM of a step
            if self.zattn:
                shapes += [(d, self.nq), (self.nq, d)]                         # the folded cross-attention projections
            wsb = max(need(n, k_, n_) for k_, n_ in shapes)
            self.ws = torch.empty(wsb // 4 + 64, device=dev)           # split-K / tail slabs of the step's GEMMs
            # few rows: the skinny split-K kernels; many rows (n >= 1024): the general path, i.e. the bf16x6 kernels
            self.gemm_kw = dict(splitk_ws=self.ws) if n < SKINNY_BELOW else dict(ws=self.ws)
            f = lambda *sh: torch.empty(*sh, device=dev)                # noqa: E731
            qw = self.nq if self.zattn else d
            self.buf = dict(x=f(n, d), x2=f(n, d), qkv=f(n, 3 * d), o=f(n, d), xa=f(n, d), q2=f(n, qw), o2=f(n, qw),
                            xb=f(n, d), pre=f(n, dff), hdn=f(n, dff), xc=[f(n, d), f(n, d)], y=f(n, d),
                            logits=f(n, V))
        self.src_valid.copy_(sv)
        # visible memory rows that form a non-empty prefix (the padding masks of Inference/*_sampling.py): the masked rows
        # behind them weigh exactly 0, gct_attn_decode does not read them; any other mask keeps all Lk rows
        cnt = sv.sum(1, dtype=torch.int32)
        prefix = (sv[:, :-1] >= sv[:, 1:]).all(1) if Lk > 1 else torch.ones(n, dtype=torch.bool, device=dev)
        self.src_klen.copy_(torch.where(prefix & (cnt > 0), cnt, torch.full_like(cnt, Lk)))
        if self.zattn:
            self.z3.copy_(z2.view(n, Le, lat))
            self._fold_cross(cl.view(n * nc, d) if c2l else None)
            return
        for li, layer in enumerate(dec.layers):                    # cross K/V: once per sequence
            kv, a = self.cross_kv[li], layer.attn_2
            ops.linear_fwd(e, [a.k_linear.weight, a.v_linear.weight], [a.k_linear.bias, a.v_linear.bias],
                           [kv, kv[:, d:]], 2 * d)

    def _fold_cross(self, cl):
        """Once per sequence (the n_c condition rows) on top of a weights-only part that is CACHED across start()
        calls while the weights stay the same (`model.weights_token()`): the cross-attention of layer l over the memory
        e = fc_z(z) is rewritten over z itself (gct_attn_decode_z).  With G = W_k W_z, Hm = W_v W_z (d x latent) and
        the block-diagonal BD[h*lat + c, h*dk + r] = G[h*dk + r, c]:
            folded query    q' = BD (W_q x + b_q)            -> zq_w = [W_q ; BD W_q], zq_b = [b_q ; BD b_q]
            folded output   y  = W_o (o_cond + BDH ctx) + b_o + W_o d   with d = W_v b_z + b_v
        (the key-side constant c = W_k b_z + b_k shifts every score of a row equally and drops out of the softmax; the
        condition rows keep explicit keys / values, shifted by -c / -d so that they share the softmax and the bias)."""
        dec, d = self.dec, self.d
        token = self.model.weights_token() if hasattr(self.model, "weights_token") else None
        key = (token, ops.gemm_get_mode(), self.nq, self.nz, self.zflat.data_ptr())
        if token is None or self._fold_key != key:
            self._fold_weights()
            self._fold_key = key
        off = self.nq - self.nz
        if off:                                                     # condition rows: k - c | v - d
            for li, layer in enumerate(dec.layers):
                a, kv = layer.attn_2, self.ckv[li]
                ops.linear_fwd(cl, [a.k_linear.weight, a.v_linear.weight], [a.k_linear.bias, a.v_linear.bias],
                               [kv, kv[:, d:]], 2 * d)
                kv.view(-1, 2, d).sub_(self.zcv[li].view(1, 2, d))

    def _fold_weights(self):
        """The weights-only part of _fold_cross (43 ms of small GEMMs and index fills for six layers: 15 % of a whole
        n = 4096 decode before it was cached)."""
        dec, d, H, dk, dev = self.dec, self.d, self.H, self.dk, self.zflat.device
        lat = self.nz // H
        Wz, bz = dec.fc_z.weight, dec.fc_z.bias
        WzT = Wz.t().contiguous()                                  # [lat, d]
        hh = torch.arange(H, device=dev)
        off = self.nq - self.nz                                    # d with condition rows, else 0

        def blockdiag(M):                                          # M [d, lat] -> [H*lat, d]
            bd = torch.zeros(H, lat, H, dk, device=dev)
            bd[hh, :, hh, :] = M.view(H, dk, lat).transpose(1, 2)
            return bd.view(H * lat, d)

        self.zcv = []
        for li, layer in enumerate(dec.layers):
            a = layer.attn_2
            G, Hm = torch.empty(d, lat, device=dev), torch.empty(d, lat, device=dev)
            ops.linear_fwd(a.k_linear.weight, [WzT], [None], [G], lat)             # W_k W_z
            ops.linear_fwd(a.v_linear.weight, [WzT], [None], [Hm], lat)            # W_v W_z
            cv = torch.empty(2, d, device=dev)                                      # c = W_k b_z + b_k ; d = W_v b_z + b_v
            ops.linear_fwd(bz.view(1, d), [a.k_linear.weight], [a.k_linear.bias], [cv[0:1]], d)
            ops.linear_fwd(bz.view(1, d), [a.v_linear.weight], [a.v_linear.bias], [cv[1:2]], d)
            self.zcv.append(cv)
            BD, BDH = blockdiag(G), blockdiag(Hm)
            qw, ow = self.zq_w[li], self.zo_w[li]
            ops.linear_fwd(BD, [a.q_linear.weight.t().contiguous()], [None], [qw[off:]], d)          # BD W_q
            ops.linear_fwd(BD, [a.q_linear.bias.view(1, d)], [None], [self.zq_b[li][off:].view(-1, 1)], 1)
            ops.linear_fwd(a.out.weight, [BDH], [None], [ow[:, off:]], self.nq)    # W_o BDH^T^T: rows of BDH are its N
            ops.linear_fwd(cv[1:2], [a.out.weight], [a.out.bias], [self.zo_b[li].view(1, d)], d)     # W_o d + b_o
            if off:
                qw[:off].copy_(a.q_linear.weight)
                self.zq_b[li][:off].copy_(a.q_linear.bias)
                ow[:, :off].copy_(a.out.weight)
        if ops.gemm_get_mode() == ops.GEMM_BF16X6:
            self.zplanes = ops.split_planes(self.zflat, self.zplanes)
            ops.register_planes(self.zflat, self.zplanes)

    # -------------------------------------------------------------------------------------
    @torch.no_grad()
    def prefill(self, ys0):
        """The prefix ys0 [n, t0] (and, with use_cond2dec, the condition tokens in front of it) through one decoder
        forward; fills the caches for positions < off + t0 and returns the logits of the last prefix position."""
        from .Model.modules import get_trg_mask
        dec, d, n = self.dec, self.d, self.n
        t0 = ys0.shape[1]
        ys0 = ys0.to(self.ys.device)
        self.ys.fill_(self.pad_id)
        self.ys[:, :t0] = ys0
        self.valid.zero_()
        self.valid[:, :self.off] = 1
        self.valid[:, self.off:self.off + t0] = (ys0 != self.pad_id).to(torch.uint8)
        self.done.zero_()
        trg_mask = get_trg_mask(ys0, self.pad_id, self.c2d, self.dconds if dec.nconds > 0 else None)
        run = engine.Run(0.0, False)
        y, saved, _, _ = engine.decoder_trunk_fwd(dec, run, ys0.contiguous(), engine._f32c(self.z),
                                                  ops.to_mask_u8(self.src_mask_in), ops.to_mask_u8(trg_mask),
                                                  self.dconds)
        Tp = self.off + t0
        for li, sv in enumerate(saved[4]):                          # per layer: (x, m1, r1, sv1, ...); sv1[2] = q|k|v
            qkv = sv[3][2].view(n, Tp, 3 * d)
            self.kc[li][:, :Tp].copy_(qkv[:, :, d:2 * d])
            self.vc[li][:, :Tp].copy_(qkv[:, :, 2 * d:])
        out = self.model.out
        ops.linear_fwd(y[:, -1].contiguous(), [out.weight], [out.bias], [self.buf["logits"]], out.weight.shape[0])
        self.pos.fill_(t0 - 1)                                      # "token t0-1 has been consumed"
        return self.buf["logits"]

    # -------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self):
        """Consume token ys[:, p] with p = *pos + 1 ... see _chain: ONE generated token, position on the device."""
        dec, d, n, T, B = self.dec, self.d, self.n, self.T, self.buf
        L = ops._L()
        st = ops._st()
        check(L.gct_decode_advance(self.pos.data_ptr(), st), "gct_decode_advance")   # pos = index of the token consumed now
        check(L.gct_decode_embed(self.ys.data_ptr(), self.ys.stride(0), self.pos.data_ptr(), self.off,
                                 dec.embed.embed.weight.data_ptr(), dec.embed.embed.weight.shape[0],
                                 dec.pe.pe.data_ptr(), B["x"].data_ptr(), n, d, math.sqrt(d), st), "gct_decode_embed")
        x = B["x"]
        for li, layer in enumerate(dec.layers):
            a1, a2, ff = layer.attn_1, layer.attn_2, layer.ff
            ops.norm_fwd(x, layer.norm_1.alpha, layer.norm_1.bias, layer.norm_1.eps, out=B["x2"])
            qkv = B["qkv"]
            ops.linear_fwd(B["x2"], [a1.q_linear.weight, a1.k_linear.weight, a1.v_linear.weight],
                           [a1.q_linear.bias, a1.k_linear.bias, a1.v_linear.bias],
                           [qkv, qkv[:, d:], qkv[:, 2 * d:]], 3 * d, **self.gemm_kw)
            ops.attn_decode(qkv, 3 * d, self.kc[li], self.vc[li], d, T * d, self.valid, T, B["o"], n,
                            self.H, 0, self.dk, pos=self.pos, cache_off=self.off, knew=qkv[:, d:],
                            vnew=qkv[:, 2 * d:], ldn=3 * d)
            ops.linear_fwd(B["o"], [a1.out.weight], [a1.out.bias], [B["xa"]], d,
                           epi=ops.EPI_DROP_RESID, resid=x, **self.gemm_kw)
            ops.norm_fwd(B["xa"], layer.norm_2.alpha, layer.norm_2.bias, layer.norm_2.eps, out=B["x2"])
            if self.zattn:
                off = self.nq - self.nz
                ops.linear_fwd(B["x2"], [self.zq_w[li]], [self.zq_b[li]], [B["q2"]], self.nq, **self.gemm_kw)
                ops.attn_decode_z(B["q2"], off, self.z3, self.ckv[li], self.Lk - self.z3.shape[1], self.src_valid,
                                  B["o2"], off, n, self.H, self.dk, klen=self.src_klen)
                ops.linear_fwd(B["o2"], [self.zo_w[li]], [self.zo_b[li]], [B["xb"]], d,
                               epi=ops.EPI_DROP_RESID, resid=B["xa"], **self.gemm_kw)
            else:
                ops.linear_fwd(B["x2"], [a2.q_linear.weight], [a2.q_linear.bias], [B["q2"]], d, **self.gemm_kw)
                kv = self.cross_kv[li]
                ops.attn_decode(B["q2"], d, kv, kv[:, d:], 2 * d, self.Lk * 2 * d, self.src_valid, self.Lk,
                                B["o2"], n, self.H, self.Lk, self.dk, klen=self.src_klen)
                ops.linear_fwd(B["o2"], [a2.out.weight], [a2.out.bias], [B["xb"]], d,
                               epi=ops.EPI_DROP_RESID, resid=B["xa"], **self.gemm_kw)
            ops.norm_fwd(B["xb"], layer.norm_3.alpha, layer.norm_3.bias, layer.norm_3.eps, out=B["x2"])
            ops.linear_fwd(B["x2"], [ff.linear_1.weight], [ff.linear_1.bias], [B["hdn"]],
                           B["hdn"].shape[1], epi=ops.EPI_GELU_DROP, pre=B["pre"], **self.gemm_kw)
            xc = B["xc"][li & 1]
            ops.linear_fwd(B["hdn"], [ff.linear_2.weight], [ff.linear_2.bias], [xc], d,
                           epi=ops.EPI_DROP_RESID, resid=B["xb"], **self.gemm_kw)
            x = xc
        ops.norm_fwd(x, dec.norm.alpha, dec.norm.bias, dec.norm.eps, out=B["y"])
        out = self.model.out
        ops.linear_fwd(B["y"], [out.weight], [out.bias], [B["logits"]], out.weight.shape[0], ws=self.ws)
        return B["logits"]

    def _select(self, mode):
        """softmax + choice of the next token from buf['logits']; written at ys[:, *pos + 1] (device position)."""
        ops.select_token(self.buf["logits"], self.ys, 0, self.valid, self.done, mode, self.pad_id, self.eos_id,
                         pos_dev=self.pos, valid_off=self.off, seed_dev=self.seed)

    # -------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, ys0, max_strlen=80, algo="greedy", seed=0, check_every=8, use_graphs=False):
        """Mirror of Sampling.decode: appends max_strlen-1 tokens to the prefix ys0 [n, t0]
        (stops early once every sample has produced <eos>, like the reference's break)."""
        n, t0 = ys0.shape
        steps = max_strlen - 1
        if self.off + t0 + steps > self.T:
            raise ValueError(f"prefix {t0} + {steps} steps exceeds the cache length {self.T - self.off}")
        mode = {"greedy": 0, "multinomial": 1}[algo]
        self.seed.fill_(int(seed) & 0x7FFFFFFFFFFFFFFF)
        self.prefill(ys0)
        self._select(mode)                                         # token t0 from the prefill's last position
        last = t0 + steps
        for i in range(1, steps):
            self._run_step(mode, use_graphs)                       # consumes token t0+i-1, writes token t0+i
            if check_every and (i + 1) % check_every == 0 and bool(self.done.all()):
                last = t0 + i + 1
                break
        ys = self.ys[:, :last]
        gen = ys[:, t0:]
        is_eos = gen == self.eos_id
        if bool(is_eos.any(dim=1).all()):                          # reference break point
            first = torch.where(is_eos, torch.arange(gen.size(1), device=gen.device)[None, :],
                                gen.size(1)).min(dim=1).values
            ys = ys[:, :t0 + int(first.max().item()) + 1]
        return ys.clone()

    def _run_step(self, mode, use_graphs):
        if not use_graphs:
            self.step()
            self._select(mode)
            return
        g = self.graphs.get(mode)
        if g is False:                                  # no usable graph for these buffers (capture failed, or replay is
            self.step()                                 # the slower launch mode on this box): same kernels, eagerly
            self._select(mode)
            return
        if g is None:
            g = self._capture(mode)
            if g is None:
                return                                  # _capture ran the step eagerly
        g.replay()

    def _state(self):
        return (self.pos.clone(), self.ys.clone(), self.valid.clone(), self.done.clone())

    def _restore(self, keep):
        self.pos.copy_(keep[0]); self.ys.copy_(keep[1]); self.valid.copy_(keep[2]); self.done.copy_(keep[3])

    def _capture(self, mode):
        """Capture step + select into one graph; returns it, or None after running the step eagerly (capture failed, or
        the replay guard found replay slower than eager launches on this box)."""
        # Warm-up run on a side stream (lazy LDS opt-ins, allocator), then capture.  The warm-up really executes a
        # step (it advances the device position and writes a token), so the state it touches is restored before
        # the capture; a capture itself executes nothing.
        keep = self._state()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.step()
            self._select(mode)
        torch.cuda.current_stream().wait_stream(s)
        # (the key / value row the warm-up appended is rewritten with the same values by the replay below)
        self._restore(keep)
        try:
            g = torch.cuda.CUDAGraph(keep_graph=True)   # the hipGraph_t stays readable (graphdiag.census)
        except TypeError:
            g = torch.cuda.CUDAGraph()
        try:
            # thread_local: another thread's runtime calls (the RCCL watchdog of a data-parallel job queries
            # events) must not invalidate this thread's capture
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self.step()
                self._select(mode)
        except RuntimeError as exc:                     # no graph for these buffers: same kernels, launched eagerly
            import warnings
            warnings.warn(f"KVDecoder: graph capture failed ({exc}); decoding without graph replay")
            torch.cuda.synchronize()
            self._restore(keep)
            self.graphs[mode] = False
            self.graph_replay = False
            self.step()
            self._select(mode)
            return None
        self.graphs[mode] = g
        if REPLAY_GUARD and not self._replay_is_fast(mode, g, keep):
            self.graphs[mode] = False
            self.graph_replay = False
            self.step()
            self._select(mode)
            return None
        return g

    def _replay_is_fast(self, mode, g, keep):
        """Replay guard: a few steps launched eagerly and a few replayed, timed on the device, state restored after
        each.  Replay is the faster way to issue the ~70 launches of a step everywhere it behaves; on boxes where it is
        clearly the slower one (round 2: 3-13x) the decoder keeps launching eagerly, says so once, and leaves the
        numbers and the graph's census in `self.replay_probe` for the caller (bench.py prints them)."""
        room = self.T - self.off - (int(keep[0].item()) + 1) - 1         # steps the caches still have room for
        k = min(4, room)
        if k < 1:
            return True

        def timed(fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(k):
                fn()
            e1.record()
            e1.synchronize()
            self._restore(keep)
            return e0.elapsed_time(e1) / k

        def eager():
            self.step()
            self._select(mode)

        g.replay()                                                       # first replay (instantiation, upload): untimed
        self._restore(keep)
        t_graph = min(timed(g.replay), timed(g.replay))
        t_eager = min(timed(eager), timed(eager))
        torch.cuda.synchronize()
        try:                                   # the census is evidence for a report, never a reason to fail a decode
            from . import graphdiag
            cen = graphdiag.census(g)
        except Exception as exc:               # noqa: BLE001 -- diagnostics library missing / hipGraphGetNodes refused
            cen = {"error": repr(exc)}
        self.replay_probe = {"ms_per_step_graph": round(t_graph, 4), "ms_per_step_eager": round(t_eager, 4),
                             "steps_timed": k, "rows": self.n, "census": cen}
        if t_graph <= REPLAY_SLOW_FACTOR * t_eager:
            return True
        import warnings
        warnings.warn(f"KVDecoder: replaying the captured step takes {t_graph:.3f} ms against {t_eager:.3f} ms for the "
                      f"same kernels launched eagerly on this box; decoding with eager launches "
                      f"(GCT_DECODE_GRAPH_GUARD=0 forces replay; python tools/graph_probe.py prints the diagnosis)")
        return False


@torch.no_grad()
def reference_style_decode(model, z, src_mask, dconds, ys0, pad_id, eos_id, max_strlen=80):
    """The reference's loop (sampling_tool.py:140-184, greedy) on the un-cached model.decode --
    used by tests/benchmarks as the baseline the KV-cached path must match token for token."""
    from .Model.modules import get_trg_mask
    ys = ys0.clone()
    done = torch.zeros(ys.size(0), dtype=torch.bool, device=ys.device)
    for _ in range(max_strlen - 1):
        trg_mask = get_trg_mask(ys, pad_id, False, dconds)
        logits = model.decode(ys, z, src_mask, trg_mask, dconds)
        nxt = logits[:, -1].argmax(-1)
        ys = torch.cat([ys, nxt[:, None]], dim=1)
        done |= nxt == eos_id
        if bool(done.all()):
            break
    return ys
