"""Randomised parity sweep of the attention kernels against fp64 (dropout 0) and, with --dump / --compare, between the
two kernel families with dropout on (same Philox bits => same results to fp32 rounding):
  python tools/attn_fuzz.py --cases 80
  GCT_ATTN_FWD_LDS=1 GCT_ATTN_BWD_LDS=1 python tools/attn_fuzz.py --cases 40 --dropout 0.1 --dump /tmp/lds.pt
  python tools/attn_fuzz.py --cases 40 --dropout 0.1 --compare /tmp/lds.pt"""
import argparse, math, sys, torch
sys.path.insert(0, ".")
from gct_plus_amd import ops
DEV = "cuda"


def sweep(cases=60, dropout=0.0, dump="", compare=""):
    a = argparse.Namespace(cases=cases, dropout=dropout, dump=dump, compare=compare)
    g = torch.Generator().manual_seed(1234)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))          # noqa: E731
    outs, worst = [], 0.0
    for case in range(a.cases):
        B, H, dk = ri(1, 5), [1, 2, 4, 8][ri(0, 3)], [16, 32, 64][ri(0, 2)]
        Lq, Lk = ri(1, 130), ri(1, 96)
        mode = ["none", "pad", "causal", "holes", "rows_empty"][ri(0, 4)]
        d = H * dk
        q2 = torch.randn(B * Lq, d, generator=g)
        kv = torch.randn(B * Lk, 2 * d, generator=g)
        do = torch.randn(B * Lq, d, generator=g)
        if ri(0, 3) == 0:
            do[torch.rand(B * Lq, generator=g) < 0.5] = 0          # zero gradient rows (dead query tiles)
        if mode == "none":
            mask = None
        elif mode == "pad":
            lens = torch.randint(1, Lk + 1, (B,), generator=g)
            mask = (torch.arange(Lk)[None, :] < lens[:, None]).to(torch.uint8)
        elif mode == "causal":
            mask = torch.tril(torch.ones(Lq, Lk, dtype=torch.uint8), diagonal=ri(0, 3))[None].repeat(B, 1, 1)
            mask[:, :, 0] = 1
        else:
            mask = (torch.rand(B, Lq, Lk, generator=g) < 0.5).to(torch.uint8)
            if mode == "rows_empty":
                mask[:, ::2, :] = 0
        mfull = None if mask is None else (mask[:, None, None, :] if mask.dim() == 2 else mask[:, None])
        qg, kvg = q2.to(DEV), kv.to(DEV)
        mg = None if mask is None else mask.to(DEV)
        seed, site = 77 + case, 5
        o, lse, _ = ops.attn_fwd(qg, kvg, kvg[:, d:], d, 2 * d, 2 * d, mg, B, H, Lq, Lk, dk, a.dropout, seed, site)
        dq = torch.empty(B * Lq, d, device=DEV)
        dkv = torch.empty(B * Lk, 2 * d, device=DEV)
        ops.attn_bwd(qg, kvg, kvg[:, d:], d, 2 * d, 2 * d, mg, o, do.to(DEV), lse, dq, dkv, dkv[:, d:], d, 2 * d, 2 * d,
                     B, H, Lq, Lk, dk, a.dropout, seed, site)
        res = [t.cpu() for t in (o, lse, dq, dkv)]
        if a.dropout == 0.0:
            qd = q2.double().view(B, Lq, H, dk).transpose(1, 2).requires_grad_()
            kd = kv[:, :d].double().reshape(B, Lk, H, dk).transpose(1, 2).requires_grad_()
            vd = kv[:, d:].double().reshape(B, Lk, H, dk).transpose(1, 2).requires_grad_()
            s = qd @ kd.transpose(-1, -2) / math.sqrt(dk)
            if mfull is not None:
                s = s.masked_fill(mfull == 0, -1e9)
            oref = s.softmax(-1) @ vd
            oref.backward(do.double().view(B, Lq, H, dk).transpose(1, 2))
            chk = [(res[0].view(B, Lq, H, dk).transpose(1, 2), oref.detach(), "o"),
                   (res[2].view(B, Lq, H, dk).transpose(1, 2), qd.grad, "dq"),
                   (res[3][:, :d].reshape(B, Lk, H, dk).transpose(1, 2), kd.grad, "dk"),
                   (res[3][:, d:].reshape(B, Lk, H, dk).transpose(1, 2), vd.grad, "dv")]
            for got, ref, what in chk:
                err = float((got.double() - ref).abs().max())
                scale = float(ref.abs().max()) + 1e-6
                worst = max(worst, err / (3e-5 + 2e-4 * scale))
                assert err <= 3e-5 + 2e-4 * scale, (case, (B, H, Lq, Lk, dk, mode), what, err, scale)
        outs.append(res)
    if a.dump:
        torch.save(outs, a.dump)
    if a.compare:
        other = torch.load(a.compare)
        for i, (x, y) in enumerate(zip(outs, other)):
            for t, u, what in zip(x, y, ("o", "lse", "dq", "dkv")):
                err = float((t - u).abs().max())
                scale = float(u.abs().max()) + 1e-6
                worst = max(worst, err / (1e-5 + 1e-4 * scale))
                assert err <= 1e-5 + 1e-4 * scale, (i, what, err, scale)
    print(f"{a.cases} cases ok, worst error / tolerance {worst:.2f}")
    return worst


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--dropout", type=float, default=0.0)
    ap.add_argument("--dump", default="")
    ap.add_argument("--compare", default="")
    b = ap.parse_args()
    sweep(b.cases, b.dropout, b.dump, b.compare)
