#!/usr/bin/env python3
"""Micro-benchmarks of the hot kernels at the BASELINE shapes (B=512): per-shape time and
TFLOP/s (GEMMs) or GB/s (bandwidth kernels), HIP-event timed, interleaved rounds."""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gct_plus_amd import ops  # noqa: E402


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def gemm_suite(reps, only=None):
    dev = "cuda"
    shapes = [("qkv", 40960, 512, 512, 3), ("out", 40960, 512, 512, 1), ("ffn1", 40960, 512, 2048, 1),
              ("ffn2", 40960, 2048, 512, 1), ("kv", 40960, 512, 512, 2), ("mulv", 40960, 512, 128, 2),
              ("sq4k", 4096, 4096, 4096, 1)]
    for name, M, K, nper, nseg in shapes:
        if only and name not in only:
            continue
        N = nper * nseg
        x = torch.randn(M, K, device=dev)
        wflat = torch.randn(nseg * nper * K, device=dev) * K ** -0.5      # one buffer, as flat.py lays weights out
        ws = [wflat[s * nper * K:(s + 1) * nper * K].view(nper, K) for s in range(nseg)]
        if ops.gemm_get_mode() == ops.GEMM_BF16X6:
            ops.register_planes(wflat, ops.split_planes(wflat))
        bs = [torch.randn(nper, device=dev) for _ in range(nseg)]
        y = torch.empty(M, N, device=dev)
        outs = [y[:, s * nper:] for s in range(nseg)]
        dy = torch.randn(M, N, device=dev)
        dys = [dy[:, s * nper:] for s in range(nseg)]
        dx = torch.empty(M, K, device=dev)
        dws = [torch.empty(nper, K, device=dev) for _ in range(nseg)]
        dbs = [torch.empty(nper, device=dev) for _ in range(nseg)]
        fl = 2.0 * M * K * N
        for kind, fn in (("fwd", lambda: ops.linear_fwd(x, ws, bs, outs, N)),
                         ("dgrad", lambda: ops.linear_dgrad(dys, N, M, ws, dx)),
                         ("wgrad", lambda: ops.linear_wgrad(dys, N, x, dws, dbs))):
            med, best = timeit(fn, reps)
            print(f"gemm {name:5s} {kind:5s} M={M} K={K} N={N}: {med*1e6:8.1f} us  {fl/med/1e12:6.1f} TF  (best {fl/best/1e12:6.1f})",
                  flush=True)


def decgemm_suite(reps):
    """Forward GEMMs of a KV-cached decode step (n rows) and of small training batches."""
    dev = "cuda"
    for M in (1024, 2048, 4096, 5184):
        for name, K, N in (("out", 512, 512), ("q'", 512, 1024), ("out'", 1024, 512), ("qkv", 512, 1536),
                           ("ffn1", 512, 2048), ("ffn2", 2048, 512)):
            x = torch.randn(M, K, device=dev)
            w = torch.randn(N * K, device=dev) * K ** -0.5
            if ops.gemm_get_mode() == ops.GEMM_BF16X6:
                ops.register_planes(w, ops.split_planes(w))
            b, y = torch.randn(N, device=dev), torch.empty(M, N, device=dev)
            wsb = torch.empty(int(ops._L().gct_linear_fwd_ws_bytes(M, K, N)) // 4 + 64, device=dev)
            med, best = timeit(lambda: ops.linear_fwd(x, [w.view(N, K)], [b], [y], N, ws=wsb), reps)
            fl = 2.0 * M * K * N
            print(f"gemm {name:5s} fwd M={M} K={K} N={N}: {med*1e6:8.1f} us  {fl/med/1e12:6.1f} TF  (best {fl/best/1e12:6.1f})", flush=True)


def epi_suite(reps):
    """What the fused epilogues cost per forward / dgrad GEMM at the training shapes: bias only vs GELU (+ dropout) vs
    dropout + residual, p = 0 against p = 0.1 (the Philox share)."""
    dev = "cuda"
    M = 40960
    for name, K, N in (("out", 512, 512), ("ffn1", 512, 2048), ("ffn2", 2048, 512)):
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N * K, device=dev) * K ** -0.5
        if ops.gemm_get_mode() == ops.GEMM_BF16X6:
            ops.register_planes(w, ops.split_planes(w))
        W = w.view(N, K)
        b, y = torch.randn(N, device=dev), torch.empty(M, N, device=dev)
        resid, pre = torch.randn(M, N, device=dev), torch.empty(M, N, device=dev)
        fl = 2.0 * M * K * N
        cases = [("bias", dict())]
        for p in (0.0, 0.1):
            cases.append((f"gelu_drop p={p}", dict(epi=ops.EPI_GELU_DROP, pre=pre, p=p, seed=3, site=1)))
            cases.append((f"drop_resid p={p}", dict(epi=ops.EPI_DROP_RESID, resid=resid, p=p, seed=3, site=1)))
        for label, kw in cases:
            med, best = timeit(lambda: ops.linear_fwd(x, [W], [b], [y], N, **kw), reps)
            print(f"epi {name:5s} fwd   {label:18s}: {med*1e6:8.1f} us  {fl/med/1e12:6.1f} TF", flush=True)
        dy, dx = torch.randn(M, N, device=dev), torch.empty(M, K, device=dev)
        prek = torch.randn(M, K, device=dev)
        cases = [("store", dict())] + [(f"gelu_bwd p={p}", dict(depi=ops.DEPI_GELU_BWD, pre=prek, p=p, seed=3, site=1))
                                       for p in (0.0, 0.1)]
        for label, kw in cases:
            med, best = timeit(lambda: ops.linear_dgrad([dy], N, M, [W], dx, **kw), reps)
            print(f"epi {name:5s} dgrad {label:18s}: {med*1e6:8.1f} us  {fl/med/1e12:6.1f} TF", flush=True)


def attn_suite(reps, fixed=False):
    dev = "cuda"
    shapes = [("enc", 512, 8, 80, 80, 64, False), ("dec", 512, 8, 81, 81, 64, True), ("cross", 512, 8, 81, 80, 64, False)]
    if os.environ.get("ATTN_LAYOUT_PROBE"):
        # same work per (batch, head) pair, but every pair's q / k / v slice is CONTIGUOUS (H = 1): what the
        # head-sliced 256-B-per-row access of the [M, 3d] projection buffer costs
        shapes = [("enc", 512, 8, 80, 80, 64, False), ("encH1", 4096, 1, 80, 80, 64, False)]
    for name, B, H, Lq, Lk, dk, causal in shapes:
        d = H * dk
        qkv = torch.randn(B * max(Lq, Lk), 3 * d, device=dev)
        # MOSES-like ragged lengths: l ~ N(35,8) clipped to [15, L]
        lens = torch.clamp(torch.round(torch.randn(B, device=dev) * 8 + 35), 15, Lk).long()
        lens[0] = Lk
        if fixed:
            lens[:] = Lk
        pad = (torch.arange(Lk, device=dev)[None, :] < lens[:, None])
        if causal:
            mask = (pad[:, None, :] & torch.ones(Lq, Lk, dtype=torch.bool, device=dev).tril_()[None]).to(torch.uint8).contiguous()
        else:
            mask = pad.to(torch.uint8).contiguous()
        q, k, v = qkv, qkv[:, d:], qkv[:, 2 * d:]
        ld = 3 * d
        if name == "encH1":
            q, k, v = (torch.randn(B * Lq, d, device=dev) for _ in range(3))
            ld = d
        mask = ops.pack_mask(mask, B, Lq, Lk)        # packed once per forward in the engine, not per call
        o, lse, _ = ops.attn_fwd(q, k, v, ld, ld, ld, mask, B, H, Lq, Lk, dk, 0.1, 1, 1)
        do = torch.randn_like(o)
        dqkv = torch.empty_like(qkv)
        dq_, dk__, dv_ = (dqkv, dqkv[:, d:], dqkv[:, 2 * d:]) if name != "encH1" else (torch.empty_like(q), torch.empty_like(k), torch.empty_like(v))
        fl = 4.0 * B * H * Lq * Lk * dk
        med, best = timeit(lambda: ops.attn_fwd(q, k, v, ld, ld, ld, mask, B, H, Lq, Lk, dk, 0.1, 1, 1), reps)
        gb = 4.0 * B * max(Lq, Lk) * d * 4
        print(f"attn {name:5s} fwd{' fixed' if fixed else ''}: {med*1e6:8.1f} us  {fl/med/1e12:6.2f} TF  {gb/med/1e9:7.0f} GB/s", flush=True)
        med, best = timeit(lambda: ops.attn_bwd(q, k, v, ld, ld, ld, mask, o, do, lse, dq_, dk__, dv_, ld, ld, ld,
                                                B, H, Lq, Lk, dk, 0.1, 1, 1), reps)
        print(f"attn {name:5s} bwd{' fixed' if fixed else ''}: {med*1e6:8.1f} us  {2.5*fl/med/1e12:6.2f} TF  {2*gb/med/1e9:7.0f} GB/s", flush=True)


def bw_suite(reps):
    dev = "cuda"
    M, d = 40960, 512
    x = torch.randn(M, d, device=dev)
    a, b = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    y, mean, rstd = ops.norm_fwd(x, a, b)
    med, _ = timeit(lambda: ops.norm_fwd(x, a, b, out=y), reps)
    print(f"norm fwd: {med*1e6:7.1f} us {2*M*d*4/med/1e9:7.0f} GB/s", flush=True)
    da, db = torch.empty(d, device=dev), torch.empty(d, device=dev)
    dx = torch.empty_like(x)
    med, _ = timeit(lambda: ops.norm_bwd(y, x, a, mean, rstd, da, db, dres=x, out=dx), reps)
    print(f"norm bwd: {med*1e6:7.1f} us {4*M*d*4/med/1e9:7.0f} GB/s", flush=True)
    n = 44_514_334
    p, g, m, v = (torch.randn(n, device=dev) for _ in range(4))
    v.abs_()
    med, _ = timeit(lambda: ops.adam_step(p, g, m, v, 1e-4, 0.9, 0.98, 1e-9, 3), reps)
    print(f"adam    : {med*1e6:7.1f} us {28*n/med/1e9:7.0f} GB/s", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--suite", default="gemm,attn,bw")
    ap.add_argument("--only", default="")
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    if "gemm" in a.suite.split(","):
        gemm_suite(a.reps, a.only.split(",") if a.only else None)
    if "decgemm" in a.suite:
        decgemm_suite(a.reps)
    if "epi" in a.suite.split(","):
        epi_suite(a.reps)
    if "attn" in a.suite:
        attn_suite(a.reps)
        attn_suite(a.reps, fixed=True)
    if "bw" in a.suite:
        bw_suite(a.reps)
