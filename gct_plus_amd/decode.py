"""KV-cached autoregressive decoding (BASELINE config 5).

The reference's Sampling.decode (Inference/sampling_tool.py:140-184) re-runs the WHOLE decoder
on ys[:, :i+1] for every generated token -- fc_z(z), all six cross-attention K/V projections
and every earlier position are recomputed 79 times, with one device->host sync per step.
Because the decoder is causal, position j's hidden states depend only on tokens 0..j, so the
same token ids come out of a single-token step that
  * projects z and the cross-attention K/V of all layers ONCE (`start`),
  * keeps per-layer self-attention q/k/v caches [n, T, d] -- the fused QKV GEMM writes its three
    segments straight into the cache slots of position `pos` (segmented-output addressing),
  * runs one query row per (sample, head) against the caches (gct_attn_decode),
  * picks the next token, updates key-valid flags and the finished mask on the device
    (gct_select_token) -- no host round trip inside a step, so a step is graph-capturable.
Work per generated token drops from O(T) decoder passes to O(1).
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import ops


class KVDecoder:
    def __init__(self, model, pad_id: int, sos_id: int, eos_id: int):
        dec = model.decoder
        if dec.use_cond2dec and dec.nconds > 0:
            raise NotImplementedError("KV-cached decode with use_cond2dec is not implemented; "
                                      "use model.decode (no shipped script sets -use_cond2dec)")
        self.model, self.dec = model, dec
        self.pad_id, self.sos_id, self.eos_id = int(pad_id), int(sos_id), int(eos_id)
        self.d = dec.d_model
        self.H = dec.layers[0].attn_1.h
        self.dk = self.d // self.H
        self.graphs = {}

    # -------------------------------------------------------------------------------------
    @torch.no_grad()
    def start(self, z, src_mask, dconds=None, max_total_len=208):
        """z [n, L_e, latent]; src_mask bool [n,1,L_e] (as the reference builds it)."""
        dec, d = self.dec, self.d
        dev = z.device
        n, Le, lat = z.shape
        if hasattr(self.model, "refresh_weight_planes"):
            self.model.refresh_weight_planes()         # the prefill GEMMs may take the bf16x6 path
        nc = dec.nconds
        c2l = dec.use_cond2lat and nc > 0
        z2 = z.reshape(n * Le, lat).float().contiguous()
        ez = torch.empty(n * Le, d, device=dev)
        ops.linear_fwd(z2, [dec.fc_z.weight], [dec.fc_z.bias], [ez], d)
        Lk, e = Le, ez
        sv = ops.to_mask_u8(src_mask).view(n, Le)
        if c2l:
            Lk = Le + nc
            cl = ops.small_linear_fwd(dconds.float().contiguous(), dec.embed_cond2lat.weight,
                                      dec.embed_cond2lat.bias)
            e = torch.empty(n * Lk, d, device=dev)
            ops.copy_rows(cl, nc, 0, e, Lk, 0, n * nc, nc, d)
            ops.copy_rows(ez, Le, 0, e, Lk, nc, n * Le, Le, d)
            sv = torch.cat([torch.ones(n, nc, dtype=torch.uint8, device=dev), sv], dim=1)
        self.src_valid = sv.contiguous()
        self.n, self.Lk, self.T = n, Lk, int(max_total_len)
        if self.T > 256 or Lk > 256:
            raise ValueError("decode lengths above 256 are not supported by gct_attn_decode")
        self.cross_kv = []
        for layer in dec.layers:                                   # cross K/V: once per sequence
            kv = torch.empty(n * Lk, 2 * d, device=dev)
            a = layer.attn_2
            ops.linear_fwd(e, [a.k_linear.weight, a.v_linear.weight], [a.k_linear.bias, a.v_linear.bias],
                           [kv, kv[:, d:]], 2 * d)
            self.cross_kv.append(kv)
        T = self.T
        self.qc = [torch.empty(n, T, d, device=dev) for _ in dec.layers]
        self.kc = [torch.empty(n, T, d, device=dev) for _ in dec.layers]
        self.vc = [torch.empty(n, T, d, device=dev) for _ in dec.layers]
        self.valid = torch.zeros(n, T, dtype=torch.uint8, device=dev)
        self.done = torch.zeros(n, dtype=torch.uint8, device=dev)
        self.ys = torch.full((n, T), self.pad_id, dtype=torch.int64, device=dev)
        # per-step scratch (fixed addresses => graph friendly)
        dff = dec.layers[0].ff.linear_1.weight.shape[0]
        V = self.model.out.weight.shape[0]
        wsb = max(ops._L().gct_linear_fwd_ws_bytes(n, dff, d), ops._L().gct_linear_fwd_ws_bytes(n, d, 3 * d),
                  ops._L().gct_linear_fwd_ws_bytes(n, d, dff))
        self.ws = torch.empty(wsb // 4 + 64, device=dev)           # split-K slabs of the skinny GEMMs
        f = lambda *s: torch.empty(*s, device=dev)
        self.buf = dict(x2=f(n, d), o=f(n, d), xa=f(n, d), q2=f(n, d), o2=f(n, d), xb=f(n, d),
                        pre=f(n, dff), hdn=f(n, dff), xc=[f(n, d), f(n, d)], y=f(n, d),
                        logits=f(n, V), tok=torch.empty(n, 1, dtype=torch.int64, device=dev))

    # -------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, pos: int):
        """Consume token ys[:, pos]; returns logits [n, V] for position pos+1."""
        dec, d, n, T, B = self.dec, self.d, self.n, self.T, self.buf
        B["tok"].copy_(self.ys[:, pos:pos + 1])
        x = ops.embed_pe_fwd(B["tok"], dec.embed.embed.weight, None, dec.pe.pe[0, pos:], 0,
                             math.sqrt(d), 0.0, 0, 0)
        for li, layer in enumerate(dec.layers):
            a1, a2, ff = layer.attn_1, layer.attn_2, layer.ff
            ops.norm_fwd(x, layer.norm_1.alpha, layer.norm_1.bias, layer.norm_1.eps, out=B["x2"])
            qs, ks, vs = self.qc[li][:, pos], self.kc[li][:, pos], self.vc[li][:, pos]
            ops.linear_fwd(B["x2"], [a1.q_linear.weight, a1.k_linear.weight, a1.v_linear.weight],
                           [a1.q_linear.bias, a1.k_linear.bias, a1.v_linear.bias], [qs, ks, vs], T * d, splitk_ws=self.ws)
            ops.attn_decode(qs, T * d, self.kc[li], self.vc[li], d, T * d, self.valid, T, B["o"], n,
                            self.H, pos + 1, self.dk)
            ops.linear_fwd(B["o"], [a1.out.weight], [a1.out.bias], [B["xa"]], d,
                           epi=ops.EPI_DROP_RESID, resid=x, splitk_ws=self.ws)
            ops.norm_fwd(B["xa"], layer.norm_2.alpha, layer.norm_2.bias, layer.norm_2.eps, out=B["x2"])
            ops.linear_fwd(B["x2"], [a2.q_linear.weight], [a2.q_linear.bias], [B["q2"]], d, splitk_ws=self.ws)
            kv = self.cross_kv[li]
            ops.attn_decode(B["q2"], d, kv, kv[:, d:], 2 * d, self.Lk * 2 * d, self.src_valid, self.Lk,
                            B["o2"], n, self.H, self.Lk, self.dk)
            ops.linear_fwd(B["o2"], [a2.out.weight], [a2.out.bias], [B["xb"]], d,
                           epi=ops.EPI_DROP_RESID, resid=B["xa"], splitk_ws=self.ws)
            ops.norm_fwd(B["xb"], layer.norm_3.alpha, layer.norm_3.bias, layer.norm_3.eps, out=B["x2"])
            ops.linear_fwd(B["x2"], [ff.linear_1.weight], [ff.linear_1.bias], [B["hdn"]],
                           B["hdn"].shape[1], epi=ops.EPI_GELU_DROP, pre=B["pre"], splitk_ws=self.ws)
            xc = B["xc"][li & 1]
            ops.linear_fwd(B["hdn"], [ff.linear_2.weight], [ff.linear_2.bias], [xc], d,
                           epi=ops.EPI_DROP_RESID, resid=B["xb"], splitk_ws=self.ws)
            x = xc
        ops.norm_fwd(x, dec.norm.alpha, dec.norm.bias, dec.norm.eps, out=B["y"])
        out = self.model.out
        ops.linear_fwd(B["y"], [out.weight], [out.bias], [B["logits"]], out.weight.shape[0])
        return B["logits"]

    # -------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, ys0, max_strlen=80, algo="greedy", seed=0, check_every=8, use_graphs=False):
        """Mirror of Sampling.decode: appends max_strlen-1 tokens to the prefix ys0 [n, t0]
        (stops early once every sample has produced <eos>, like the reference's break)."""
        n, t0 = ys0.shape
        steps = max_strlen - 1
        if t0 + steps > self.T:
            raise ValueError(f"prefix {t0} + {steps} steps exceeds the cache length {self.T}")
        mode = {"greedy": 0, "multinomial": 1}[algo]
        self.ys[:, :t0] = ys0.to(self.ys.device)
        self.valid[:, :t0] = (self.ys[:, :t0] != self.pad_id).to(torch.uint8)
        self.done.zero_()
        for pos in range(t0 - 1):                                  # prefix tokens fill the caches
            self._run_step(pos, None, use_graphs)
        last = t0 + steps
        for i in range(steps):
            pos = t0 - 1 + i
            self._run_step(pos, (mode, seed), use_graphs)
            if check_every and (i + 1) % check_every == 0 and bool(self.done.all()):
                last = pos + 2
                break
        ys = self.ys[:, :last]
        gen = ys[:, t0:]
        is_eos = gen == self.eos_id
        if bool(is_eos.any(dim=1).all()):                          # reference break point
            first = torch.where(is_eos, torch.arange(gen.size(1), device=gen.device)[None, :],
                                gen.size(1)).min(dim=1).values
            ys = ys[:, :t0 + int(first.max().item()) + 1]
        return ys.clone()

    def _run_step(self, pos, select, use_graphs):
        if not use_graphs:
            logits = self.step(pos)
            if select is not None:
                ops.select_token(logits, self.ys, pos + 1, self.valid, self.done, select[0], self.pad_id,
                                 self.eos_id, select[1])
            return
        key = (pos, select)
        g = self.graphs.get(key)
        if g is None:
            # warm-up run on a side stream (lazy LDS opt-ins, allocator), then capture
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self.step(pos)
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                logits = self.step(pos)
                if select is not None:
                    ops.select_token(logits, self.ys, pos + 1, self.valid, self.done, select[0],
                                     self.pad_id, self.eos_id, select[1])
            self.graphs[key] = g
        g.replay()


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
