"""Randomised parity sweep of the nn.Linear kernels (forward / dgrad / wgrad, all fused epilogues, both arithmetic modes)
against fp64 on the GPU.  Shapes are drawn around the places where the host side changes route: tile counts next to
multiples of the 256 CUs (tail balancing), few tiles (skinny split-K, panel kernel, split-K over the whole problem),
ragged M / N, K from 32 to 4096, 1-3 weight segments, quad-mapped GELU-backward, weight gradients over a tile list.
  python tools/gemm_fuzz.py --cases 80 [--seed 1]
Prints one line per case; exit code 1 when a case is outside the tolerance (the tolerances of tests/test_kernels_gpu.py)."""
import argparse, math, sys, types, torch
sys.path.insert(0, ".")
from gct_plus_amd import ops
DEV = "cuda"


def _err(a, ref, atol, rtol):
    """max over elements of |a - ref| / (atol + rtol |ref|): <= 1 passes"""
    a, ref = a.double(), ref.double()
    if not torch.isfinite(a).all():
        return float("inf")
    return float(((a - ref).abs() / (atol + rtol * ref.abs())).max()) if a.numel() else 0.0


def _shape(ri):
    kind = ri(0, 5)
    if kind == 0:      # around k * 256 tiles of 128 x 256
        tn = ri(1, 8)
        tm = max(1, (256 * ri(1, 3) + ri(-6, 6)) // tn)
        M, N = tm * 128 - ri(0, 127), tn * 256 - 4 * ri(0, 40)
    elif kind == 1:    # decode-like: few rows
        M, N = ri(1, 1100), [64, 128, 512, 1536, 2048][ri(0, 4)]
    elif kind == 2:    # vocabulary head: very few columns
        M, N = ri(100, 9000), ri(5, 40)
    elif kind == 3:    # ragged everything
        M, N = ri(1, 5000), 4 * ri(1, 300)
    elif kind == 4:    # the training shapes at smaller batch
        M, N = 80 * ri(8, 140), [512, 1024, 1536, 2048][ri(0, 3)]
    else:
        M, N = 128 * ri(1, 40), 256 * ri(1, 6)
    K = [32, 64, 96, 256, 512, 512, 512, 768, 1024, 2048, 2048, 4096][ri(0, 11)]
    if ri(0, 9) == 0:
        K = ri(1, 200)                # not a multiple of 32: fp32 kernels
    if M * N * K > 6e10:
        K = 512
    return max(M, 1), max(N, 1), K


def sweep(cases=60, seed=1, verbose=True):
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))          # noqa: E731
    gd = torch.Generator(device=DEV).manual_seed(seed)
    rn = lambda *s, scale=1.0: torch.randn(*s, device=DEV, generator=gd) * scale     # noqa: E731
    worst, bad = 0.0, []
    for case in range(cases):
        M, N, K = _shape(ri)
        nseg = [1, 1, 2, 3][ri(0, 3)]
        if N % nseg:
            nseg = 1
        nper = N // nseg
        p = [0.0, 0.1][ri(0, 1)]
        seed_d, site = 1000 + case, ri(1, 9)
        x = rn(M, K)
        ws = [rn(nper, K, scale=max(K, 64) ** -0.5) for _ in range(nseg)]
        bs = [rn(nper) if ri(0, 5) else None for _ in range(nseg)]
        if any(b is None for b in bs):
            bs = [None] * nseg
        resid, dy = rn(M, N), rn(M, N)
        if ri(0, 3) == 0:
            dy[torch.rand(M, device=DEV, generator=gd) < 0.6] = 0
        flat = torch.cat([w.reshape(-1) for w in ws])
        pad = (-flat.numel()) % 4
        if pad:
            flat = torch.cat([flat, flat.new_zeros(pad)])
        flat = flat.contiguous()
        wv, o = [], 0
        for w in ws:
            wv.append(flat[o:o + w.numel()].view(w.shape))
            o += w.numel()
        use_planes = flat.data_ptr() % 16 == 0 and all(v.data_ptr() % 16 == 0 for v in wv)
        if use_planes:
            ops.register_planes(flat, ops.split_planes(flat))
        W = torch.cat(ws).double()
        bias = torch.zeros(N, device=DEV, dtype=torch.float64) if bs[0] is None else torch.cat(bs).double()
        u = x.double() @ W.t() + bias
        dys = lambda t: [t[:, s * nper:] for s in range(nseg)]                    # noqa: E731
        fwd_epi = ri(0, 2) if nseg == 1 else 0
        dg_epi = ri(0, 2) if nseg == 1 else ri(0, 1)
        # quad-mapped GELU backward: dy2 [Mc, N] -> dpre [Mc, K2] with pre [M2, K2] read through the map
        quads = None
        if dg_epi == 2 and M % 4 == 0 and ri(0, 1):
            nq_full = M // 4 + ri(1, 400)
            quads = torch.randperm(nq_full, generator=g)[:M // 4].sort().values.to(torch.int32).to(DEV)
        errs, res = {}, {}
        k0 = ops._L().gct_gemm_x6_kernel_launches()
        try:
            for mode in (ops.GEMM_F32, ops.GEMM_BF16X6):
                ops.gemm_set_mode(mode)
                y, pre = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
                if fwd_epi == 0:
                    ops.linear_fwd(x, wv, bs, dys(y), N)
                    ref_y = u
                elif fwd_epi == 1:
                    ops.linear_fwd(x, wv, bs, [y], N, epi=ops.EPI_GELU_DROP, pre=pre, p=p, seed=seed_d, site=site)
                    ref_y = torch.nn.functional.gelu(u)
                else:
                    ops.linear_fwd(x, wv, bs, [y], N, epi=ops.EPI_DROP_RESID, resid=resid, p=p, seed=seed_d, site=site)
                    ref_y = u
                dx = rn(M, K) if dg_epi == 1 else torch.empty(M, K, device=DEV)
                base = dx.clone() if dg_epi == 1 else None
                pre_in = rn(M, K) if dg_epi == 2 else None
                if dg_epi == 2 and quads is not None:
                    # the pre-activation stays in a larger row space and is read through the quad map
                    pre_big = rn(4 * nq_full, K)
                    rows = (quads.long()[:, None] * 4 + torch.arange(4, device=DEV)[None, :]).reshape(-1)
                    pre_in = pre_big[rows].contiguous()
                    live = types.SimpleNamespace(quad_list=quads, Mc=M)
                    ops.linear_dgrad(dys(dy), N, M, wv, dx, depi=ops.DEPI_GELU_BWD, pre=pre_big, p=0.0, seed=seed_d,
                                     site=site, live=live, pre_full=True)
                elif dg_epi == 2:
                    ops.linear_dgrad(dys(dy), N, M, wv, dx, depi=ops.DEPI_GELU_BWD, pre=pre_in, p=0.0, seed=seed_d, site=site)
                else:
                    ops.linear_dgrad(dys(dy), N, M, wv, dx, depi=dg_epi)
                dws = [torch.empty(nper, K, device=DEV) for _ in range(nseg)]
                dbs = [torch.empty(nper, device=DEV) for _ in range(nseg)]
                kt = ops.nonzero_row_tiles(dy) if (M % 32 == 0 and N % 4 == 0 and ri(0, 1)) else None
                ops.linear_wgrad(dys(dy), N, x, dws, dbs, kt=kt)
                res[mode] = (y, pre if fwd_epi == 1 else None, dx, torch.cat(dws), torch.cat(dbs))
                # references
                e = {}
                if fwd_epi == 0:
                    e["fwd"] = _err(y, ref_y, 2e-5, 2e-5)
                elif fwd_epi == 1:
                    keep = (y != 0) | (ref_y == 0)
                    e["fwd"] = _err(torch.where(keep, y.double() * (1 - p), ref_y), ref_y, 3e-5, 3e-5)
                    e["pre"] = _err(pre, u, 2e-5, 2e-5)
                    e["rate"] = abs(float(keep.double().mean()) - (1 - p)) / (0.02 + 3.0 / math.sqrt(M * N)) if p else 0.0
                else:
                    kept = (y != resid)
                    e["fwd"] = _err(torch.where(kept, (y.double() - resid.double()) * (1 - p), u), u, 3e-5, 3e-5)
                    e["rate"] = abs(float(kept.double().mean()) - (1 - p)) / (0.02 + 3.0 / math.sqrt(M * N)) if p else 0.0
                gx = dy.double() @ W
                if dg_epi == 1:
                    gx = gx + base.double()
                elif dg_epi == 2:
                    pd = pre_in.double().requires_grad_()
                    (torch.nn.functional.gelu(pd) * gx).sum().backward()
                    gx = pd.grad
                e["dgrad"] = _err(dx, gx, 5e-5, 1e-4)
                tw = 1.5e-4 * math.sqrt(M / 100 + 1)      # max over up to 1e7 outputs of an fp32 chain of M terms
                e["wgrad"] = _err(res[mode][3], dy.double().t() @ x.double(), tw, 1e-4)
                e["dbias"] = _err(res[mode][4], dy.double().sum(0), tw, 1e-4)
                errs[mode] = e
            # same dropout mask in both modes
            a, c = res[ops.GEMM_F32], res[ops.GEMM_BF16X6]
            same_mask = True
            # (an element whose kept value is below the rounding of what it is added to cannot be told from a dropped one)
            if p and fwd_epi == 1:
                same_mask = not bool((((a[0] == 0) != (c[0] == 0)) & (ref_y.abs() > 1e-4)).any())
            if p and fwd_epi == 2:
                same_mask = not bool((((a[0] == resid) != (c[0] == resid)) & (u.abs() > 1e-4)).any())
        finally:
            ops.gemm_set_mode(ops.GEMM_BF16X6)
            if use_planes:
                ops.unregister_planes(flat)
        w_case = max(max(e.values()) for e in errs.values())
        ok = w_case <= 1.0 and same_mask
        worst = max(worst, w_case)
        line = (f"case {case:3d} M={M:6d} K={K:5d} N={N:5d} nseg={nseg} fwd_epi={fwd_epi} dgrad_epi={dg_epi} p={p} "
                f"planes={int(use_planes)} qmap={int(quads is not None)} x6_launches={ops._L().gct_gemm_x6_kernel_launches() - k0} worst={w_case:.3f} mask_equal={same_mask}")
        if verbose or not ok:
            print(line + ("" if ok else "   <-- FAIL " + repr(errs)), flush=True)
        if not ok:
            bad.append(line)
    print(f"{cases} cases, worst error / tolerance = {worst:.3f}, failures: {len(bad)}", flush=True)
    return worst, bad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    worst, bad = sweep(a.cases, a.seed)
    sys.exit(1 if bad else 0)
