"""Randomised model-level parity sweep: the HIP training step (forward, loss, backward) against the CPU oracle on random
small configurations -- model type, depth, width / heads (head dims 16 / 32 / 64), dff, latent size, batch size, padded
length, length distribution (ragged, all full, very short, one long sample among short ones), cond2dec on the
conditional types.  Dropout 0 (the oracle's masks are torch's, not Philox's).  Tolerances of tests/test_model_gpu.py.
  python tools/model_fuzz.py --cases 40 [--seed 1]
Needs the oracle (test infrastructure): run from a checkout that has oracle/."""
import argparse, sys, torch
sys.path.insert(0, ".")
from gct_plus_amd import synthetic
PAD = synthetic.PAD_ID


def _dataset(mtype, B, S, kind, g):
    ds = synthetic.make_dataset(B, max_len=S, model_type=mtype, seed=int(torch.randint(0, 10 ** 6, (1,), generator=g)),
                                fixed_len=(kind == "full"))
    if kind in ("short", "one_long"):
        # re-pad: every sample (but sample 0 in "one_long") keeps 1..4 tokens
        src, trg = ds["src"], ds["trg"]
        for i in range(0 if kind == "short" else 1, B):
            ln = int(torch.randint(1, min(4, S) + 1, (1,), generator=g))
            src[i, ln:] = PAD
            trg[i, ln + 1] = synthetic.EOS_ID
            trg[i, ln + 2:] = PAD
    return ds


def sweep(cases=30, seed=1, verbose=True):
    from gct_plus_amd.Model import forward_propagation, model_dict
    from gct_plus_amd.Train.trainer1 import loss_function
    from oracle import gct_oracle as O
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))          # noqa: E731
    bad, worst = [], 0.0
    for case in range(cases):
        mtype = ["vaetf", "pvaetf", "scavaetf", "pscavaetf"][ri(0, 3)]
        d_model, h = [(64, 4), (64, 2), (128, 8), (128, 4), (128, 2), (256, 4), (32, 2)][ri(0, 6)]
        kw = dict(N=ri(1, 3), d_model=d_model, dff=[64, 128, 256, 512][ri(0, 3)], h=h, latent_dim=[8, 16, 32, 64][ri(0, 3)])
        B, S = ri(1, 20), [ri(3, 30), ri(30, 100), ri(100, 140)][min(ri(0, 5), 2) if ri(0, 1) else 0]
        kind = ["ragged", "ragged", "full", "short", "one_long"][ri(0, 4)]
        nc = synthetic.n_conds(mtype)
        c2d = bool(nc and ri(0, 2) == 0)
        vs, vt = synthetic.vocab_sizes(mtype)
        torch.manual_seed(1000 + case)
        model = model_dict[mtype](vs, vt, dropout=0.0, nconds=nc, use_cond2dec=c2d, use_cond2lat=not c2d if nc else True,
                                  **kw).cuda().train()
        cfg = O.make_cfg(mtype, vs, vt, dropout=0.0, nconds=nc, use_cond2dec=c2d, use_cond2lat=not c2d if nc else True, **kw)
        P = O.make_leaves({k: v.detach().cpu() for k, v in model.state_dict().items()})
        ds = _dataset(mtype, B, S, kind, g)
        Le = S + nc
        eps = torch.randn(B, Le, kw["latent_dim"], generator=g)
        (model.sampler if hasattr(model, "sampler") else model.encoder).eps_override = eps
        b = {k: v.cuda() for k, v in ds.items()}
        beta = 0.05
        skip = bool(ri(0, 1))           # the trainer's mode: decoder rows of padded targets are not computed
        prop, mol, mu, lv, z = forward_propagation[mtype](model, b, PAD, c2d, skip_ignored=skip)
        ys = b["trg"][:, 1:].contiguous().view(-1)
        ys_cond = b["dconds"].unsqueeze(2).contiguous().view(-1, nc, 1) if nc else None
        loss, rce, _, kld = loss_function(beta, prop, mol, ys_cond, ys, mu, lv, c2d, PAD)
        loss.backward()
        sm, tm, trg_in = O.batch_masks(cfg, ds, PAD)
        oprop, omol, omu, olv, oz = O.forward(P, cfg, ds["src"], trg_in, sm, tm, ds.get("econds"), ds.get("dconds"),
                                             eps=eps, train=True)
        oys_cond = ds["dconds"].unsqueeze(2).contiguous().view(-1, nc, 1) if nc else None
        oloss = O.loss_function(beta, oprop, omol, oys_cond, ds["trg"][:, 1:].contiguous().view(-1), omu, olv, c2d, PAD)[0]
        oloss.backward()

        def rel(got, ref, atol, rtol):
            got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
            if not torch.isfinite(got).all():
                return float("inf")
            return float(((got - ref).abs() / (atol + rtol * ref.abs())).max()) if got.numel() else 0.0
        if skip and not c2d:            # logits are only promised on the rows the loss reads
            keep = (ds["trg"][:, 1:] != PAD)
            mol, omol = mol.detach().cpu()[keep], omol.detach()[keep]
        e = {"logits": rel(mol, omol, 1e-4, 1e-4), "mu": rel(mu, omu, 1e-4, 1e-4), "z": rel(z, oz, 1e-4, 1e-4),
             "loss": abs(loss.item() - oloss.item()) / (2e-5 * abs(oloss.item()) + 1e-6)}
        if c2d:
            e["prop"] = rel(prop, oprop, 1e-4, 1e-4)
        floor = 2e-6 * max(float(v.grad.abs().max()) for v in P.values() if v.grad is not None)
        gw, gname = 0.0, ""
        for name, p in model.named_parameters():
            if P[name].grad is None:
                if p.grad is not None:
                    gw, gname = float("inf"), name + " (gradient where the oracle has none)"
                continue
            if p.grad is None:
                gw, gname = float("inf"), name + " (no gradient)"
                continue
            eg = P[name].grad
            r = rel(p.grad, eg, 1e-5 * float(eg.abs().max()) + floor, 1e-3)
            if r > gw:
                gw, gname = r, name
        e["grads"] = gw
        w_case = max(e.values())
        worst = max(worst, w_case) if w_case != float("inf") else float("inf")
        ok = w_case <= 1.0
        line = (f"case {case:3d} {mtype:9s} N={kw['N']} d={d_model} h={h} dff={kw['dff']} lat={kw['latent_dim']} B={B:2d} "
                f"S={S:3d} {kind:8s} cond2dec={int(c2d)} skip={int(skip)} worst={w_case:.3f} (grads {gw:.3f} {gname})")
        if verbose or not ok:
            print(line + ("" if ok else "   <-- FAIL " + repr(e)), flush=True)
        if not ok:
            bad.append(line)
        del model
    print(f"{cases} cases, worst error / tolerance = {worst:.3f}, failures: {len(bad)}", flush=True)
    return worst, bad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    sys.exit(1 if sweep(a.cases, a.seed)[1] else 0)
