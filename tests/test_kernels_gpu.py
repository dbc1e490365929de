"""Per-kernel parity: every C-ABI entry point (through gct_plus_amd.ops) against an fp64
PyTorch-CPU restatement of the same op.  Tolerances are stated per test; fp32 MFMA is an
exact-fp32 fma chain so 1e-5-level agreement with fp64 is expected at these sizes."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from gct_plus_amd import ops as _ops
    _ops._L()
    return _ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


def close(got, ref, atol, rtol, what=""):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    assert not bad.any(), f"{what}: max err {err.max().item():.3e} (ref scale {ref.abs().max().item():.3e}), {int(bad.sum())} bad"


def ref_norm(x, a, b, eps=1e-6):
    return a * (x - x.mean(-1, keepdim=True)) / (x.std(-1, keepdim=True) + eps) + b


@pytest.mark.parametrize("rows,d", [(7, 64), (1000, 512), (33, 2048), (5, 12)])
def test_norm(ops, rows, d):
    x, a, b, dy, dres = rnd(rows, d, seed=1), rnd(d, seed=2) + 1, rnd(d, seed=3), rnd(rows, d, seed=4), rnd(rows, d, seed=5)
    y, mean, rstd = ops.norm_fwd(x.to(DEV), a.to(DEV), b.to(DEV))
    xd, ad, bd = x.double().requires_grad_(), a.double().requires_grad_(), b.double().requires_grad_()
    yr = ref_norm(xd, ad, bd)
    close(y, yr, 1e-5, 1e-5, "norm fwd")
    yr.backward(dy.double())
    da, db = torch.empty(d, device=DEV), torch.empty(d, device=DEV)
    dx = ops.norm_bwd(dy.to(DEV), x.to(DEV), a.to(DEV), mean, rstd, da, db, dres=dres.to(DEV))
    close(dx, xd.grad + dres.double(), 2e-5, 1e-4, "norm dx")
    close(da, ad.grad, 1e-4 * math.sqrt(rows), 1e-4, "norm dalpha")
    close(db, bd.grad, 1e-4 * math.sqrt(rows), 1e-4, "norm dbias")


def test_norm_known_answer(ops):
    # Norm(4)([1,2,3,4])
    y, _, _ = ops.norm_fwd(torch.tensor([[1.0, 2.0, 3.0, 4.0]], device=DEV), torch.ones(4, device=DEV), torch.zeros(4, device=DEV))
    close(y, torch.tensor([[-1.161894, -0.387298, 0.387298, 1.161894]]), 1e-6, 0, "norm known answer")


@pytest.mark.parametrize("M,K,nper,nseg", [(300, 512, 512, 3), (129, 64, 64, 3), (5184, 512, 30, 1),
                                           (260, 128, 512, 1), (1000, 2048, 512, 1), (77, 512, 128, 2)])
def test_linear_fwd_dgrad_wgrad(ops, M, K, nper, nseg):
    x = rnd(M, K, seed=1)
    ws = [rnd(nper, K, seed=10 + s, scale=K ** -0.5) for s in range(nseg)]
    bs = [rnd(nper, seed=20 + s) for s in range(nseg)]
    N = nper * nseg
    y = torch.empty(M, N, device=DEV)
    outs = [y[:, s * nper:] for s in range(nseg)]
    xg = x.to(DEV)
    wg = [w.to(DEV) for w in ws]
    bg = [b.to(DEV) for b in bs]
    ops.linear_fwd(xg, wg, bg, outs, N)
    W = torch.cat(ws).double()
    ref = x.double() @ W.t() + torch.cat(bs).double()
    close(y, ref, 2e-5, 2e-5, "linear fwd")
    # dgrad
    dy = rnd(M, N, seed=3)
    dyg = dy.to(DEV)
    dys = [dyg[:, s * nper:] for s in range(nseg)]
    dx = torch.empty(M, K, device=DEV)
    ops.linear_dgrad(dys, N, M, wg, dx)
    close(dx, dy.double() @ W, 5e-5, 5e-5, "linear dgrad")
    base = rnd(M, K, seed=4)
    dx2 = base.to(DEV).clone()
    ops.linear_dgrad(dys, N, M, wg, dx2, depi=ops.DEPI_ACCUM)
    close(dx2, dy.double() @ W + base.double(), 5e-5, 5e-5, "linear dgrad accum")
    # wgrad
    dws = [torch.empty(nper, K, device=DEV) for _ in range(nseg)]
    dbs = [torch.empty(nper, device=DEV) for _ in range(nseg)]
    ops.linear_wgrad(dys, N, xg, dws, dbs)
    refw = dy.double().t() @ x.double()
    close(torch.cat(dws), refw, 1e-4 * math.sqrt(M / 100 + 1), 1e-4, "linear wgrad")
    close(torch.cat(dbs), dy.double().sum(0), 1e-4 * math.sqrt(M / 100 + 1), 1e-4, "linear bias grad")


def test_linear_epilogues_no_dropout(ops):
    M, K, N = 200, 64, 256
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.2), rnd(N, seed=3), rnd(M, N, seed=4)
    xg, wg, bg, rg = x.to(DEV), w.to(DEV), b.to(DEV), r.to(DEV)
    y = torch.empty(M, N, device=DEV)
    pre = torch.empty(M, N, device=DEV)
    ops.linear_fwd(xg, [wg], [bg], [y], N, epi=ops.EPI_GELU_DROP, pre=pre)
    u = x.double() @ w.double().t() + b.double()
    close(pre, u, 2e-5, 2e-5, "pre")
    close(y, F.gelu(u), 2e-5, 2e-5, "gelu")
    ops.linear_fwd(xg, [wg], [bg], [y], N, epi=ops.EPI_DROP_RESID, resid=rg)
    close(y, u + r.double(), 2e-5, 2e-5, "resid")
    # GELU backward epilogue: dx = (dy @ W2) * gelu'(pre)
    W2 = rnd(K, N, seed=5, scale=0.1)
    dy = rnd(M, K, seed=6)
    dpre = torch.empty(M, N, device=DEV)
    ops.linear_dgrad([dy.to(DEV)], K, M, [W2.to(DEV)], dpre, depi=ops.DEPI_GELU_BWD, pre=pre)
    ud = u.clone().requires_grad_()
    (F.gelu(ud) * (dy.double() @ W2.double())).sum().backward()
    close(dpre, ud.grad, 2e-5, 1e-4, "gelu bwd")


def test_dropout_sites_consistent(ops):
    """fwd epilogue masks == masks regenerated by the backward kernels; keep rate ~ 1-p."""
    M, K, N, p, seed = 512, 64, 256, 0.1, 1234
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.2), rnd(N, seed=3)
    xg, wg, bg = x.to(DEV), w.to(DEV), b.to(DEV)
    zeros = torch.zeros(M, N, device=DEV)
    y = torch.empty(M, N, device=DEV)
    ops.linear_fwd(xg, [wg], [bg], [y], N, epi=ops.EPI_DROP_RESID, resid=zeros, p=p, seed=seed, site=7)
    u = (x.double() @ w.double().t() + b.double())
    keep = (y != 0).cpu()
    rate = keep.float().mean().item()
    assert abs(rate - (1 - p)) < 0.01, rate
    close(y, torch.where(keep, u / (1 - p), torch.zeros_like(u)), 3e-5, 3e-5, "dropout fwd values")
    ones = torch.ones(M, N, device=DEV)
    dmask = ops.dropout_bwd(ones, p, seed, 7)
    assert torch.equal((dmask != 0).cpu(), keep)
    close(dmask, keep.double() / (1 - p), 1e-6, 1e-6)
    other = ops.dropout_bwd(ones, p, seed, 8)
    assert not torch.equal((other != 0).cpu(), keep)          # different site => different mask
    # GELU_DROP site vs GELU_BWD site
    pre = torch.empty(M, N, device=DEV)
    ops.linear_fwd(xg, [wg], [bg], [y], N, epi=ops.EPI_GELU_DROP, pre=pre, p=p, seed=seed, site=9)
    keep2 = (y != 0).cpu() | (F.gelu(u) == 0)
    dy = rnd(M, K, seed=6)
    W2 = rnd(K, N, seed=5, scale=0.1)
    dpre = torch.empty(M, N, device=DEV)
    ops.linear_dgrad([dy.to(DEV)], K, M, [W2.to(DEV)], dpre, depi=ops.DEPI_GELU_BWD, pre=pre, p=p, seed=seed, site=9)
    ud = u.clone().requires_grad_()
    (F.gelu(ud) * keep2.double() / (1 - p) * (dy.double() @ W2.double())).sum().backward()
    close(dpre, ud.grad, 3e-5, 1e-4, "gelu+dropout bwd")


def ref_attention(q, k, v, mask, scale, keep=None, pkeep=1.0):
    s = q @ k.transpose(-1, -2) * scale
    if mask is not None:
        s = s.masked_fill(mask == 0, -1e9)
    pr = s.softmax(-1)
    pd = pr if keep is None else pr * keep / pkeep
    return pd @ v, pr


@pytest.mark.parametrize("B,H,Lq,Lk,dk,mode", [(3, 4, 20, 20, 16, "pad"), (2, 8, 81, 81, 64, "causal"),
                                               (2, 8, 81, 86, 64, "pad"), (2, 2, 5, 128, 32, "none"),
                                               (1, 8, 80, 80, 64, "pad"), (3, 2, 97, 120, 16, "pad"),
                                               (2, 8, 171, 171, 64, "causal"), (2, 8, 171, 173, 64, "pad"),
                                               (1, 4, 200, 200, 64, "causal"), (2, 2, 5, 208, 32, "none"),
                                               (2, 4, 203, 198, 16, "pad"), (2, 4, 40, 37, 16, "holes"),
                                               (2, 8, 81, 81, 64, "holes"), (2, 4, 150, 70, 64, "pad"),
                                               (2, 4, 150, 70, 64, "causal"), (3, 2, 33, 96, 32, "holes")])
def test_attention(ops, B, H, Lq, Lk, dk, mode):
    d = H * dk
    qkv = rnd(B * max(Lq, Lk), 3 * d, seed=1)
    q2, k2, v2 = rnd(B * Lq, d, seed=2), rnd(B * Lk, 2 * d, seed=3), None
    # q from its own buffer, k/v interleaved in one [B*Lk, 2d] buffer (cross-attention layout)
    qg, kvg = q2.to(DEV), k2.to(DEV)
    kview, vview = kvg[:, :d], kvg[:, d:]
    lens = torch.tensor([Lk - (i * 3) % max(1, Lk // 2) for i in range(B)])
    if mode == "pad":
        mask = (torch.arange(Lk)[None, :] < lens[:, None]).to(torch.uint8)          # [B,Lk]
        mfull = mask[:, None, None, :]
    elif mode == "causal":
        pad = (torch.arange(Lk)[None, :] < lens[:, None])
        mask = (pad[:, None, :] & torch.tril(torch.ones(Lq, Lk, dtype=torch.bool))[None]).to(torch.uint8)
        mfull = mask[:, None]
    elif mode == "holes":
        # arbitrary mask with query rows that see NO key (masked_fill(-1e9) on the whole row => uniform attention,
        # gradient flows to V only) next to ordinary rows -- the left-padded-batch situation
        mask = (torch.rand(B, Lq, Lk, generator=torch.Generator().manual_seed(8)) < 0.6).to(torch.uint8)
        mask[:, ::3, :] = 0
        mask[:, 1, 5:] = 0
        mfull = mask[:, None]
    else:
        mask, mfull = None, None
    mg = None if mask is None else mask.to(DEV)
    o, lse, probs = ops.attn_fwd(qg, kview, vview, d, 2 * d, 2 * d, mg, B, H, Lq, Lk, dk, 0.0, 0, 0, want_probs=True)
    qd = q2.double().view(B, Lq, H, dk).transpose(1, 2).requires_grad_()
    kd = k2[:, :d].double().reshape(B, Lk, H, dk).transpose(1, 2).requires_grad_()
    vd = k2[:, d:].double().reshape(B, Lk, H, dk).transpose(1, 2).requires_grad_()
    oref, pref = ref_attention(qd, kd, vd, mfull, 1 / math.sqrt(dk))
    close(probs, pref, 1e-6, 1e-5, "probs")
    close(o.view(B, Lq, H, dk).transpose(1, 2), oref, 1e-5, 1e-5, "attn out")
    do = rnd(B * Lq, d, seed=5)
    oref.backward(do.double().view(B, Lq, H, dk).transpose(1, 2))
    dq = torch.empty(B * Lq, d, device=DEV)
    dkv = torch.empty(B * Lk, 2 * d, device=DEV)
    ops.attn_bwd(qg, kview, vview, d, 2 * d, 2 * d, mg, o, do.to(DEV), lse, dq, dkv[:, :d], dkv[:, d:],
                 d, 2 * d, 2 * d, B, H, Lq, Lk, dk, 0.0, 0, 0)
    close(dq.view(B, Lq, H, dk).transpose(1, 2), qd.grad, 2e-5, 1e-4, "dq")
    close(dkv[:, :d].reshape(B, Lk, H, dk).transpose(1, 2), kd.grad, 2e-5, 1e-4, "dk")
    close(dkv[:, d:].reshape(B, Lk, H, dk).transpose(1, 2), vd.grad, 2e-5, 1e-4, "dv")


def test_attention_many_pairs_per_workgroup(ops):
    """More (batch, head) pairs than persistent workgroups (2 per CU): every workgroup walks several pairs through
    the software pipeline (next pair's K / V / Q rows loaded during this pair's MFMAs).  Causal + ragged mask,
    dropout off; forward and backward against fp64."""
    B, H, L, dk = 96, 8, 81, 64                      # 768 pairs > 512 workgroups
    d = H * dk
    qkv = rnd(B * L, 3 * d, seed=11)
    lens = torch.tensor([L - (i * 7) % 60 for i in range(B)])
    pad = torch.arange(L)[None, :] < lens[:, None]
    mask = (pad[:, None, :] & torch.tril(torch.ones(L, L, dtype=torch.bool))[None]).to(torch.uint8)
    g = qkv.to(DEV)
    o, lse, _ = ops.attn_fwd(g, g[:, d:], g[:, 2 * d:], 3 * d, 3 * d, 3 * d, mask.to(DEV), B, H, L, L, dk, 0.0, 0, 0)
    sp = lambda t: t.double().reshape(B, L, H, dk).transpose(1, 2).requires_grad_()               # noqa: E731
    qd, kd, vd = sp(qkv[:, :d]), sp(qkv[:, d:2 * d]), sp(qkv[:, 2 * d:])
    oref, _ = ref_attention(qd, kd, vd, mask[:, None], 1 / math.sqrt(dk))
    close(o.view(B, L, H, dk).transpose(1, 2), oref, 1e-5, 1e-5, "attn out (many pairs)")
    do = rnd(B * L, d, seed=12)
    do[40 * L:] = 0.0                                # whole samples without gradient: zero-dO tiles are skipped
    oref.backward(do.double().view(B, L, H, dk).transpose(1, 2))
    dqkv = torch.empty(B * L, 3 * d, device=DEV)
    ops.attn_bwd(g, g[:, d:], g[:, 2 * d:], 3 * d, 3 * d, 3 * d, mask.to(DEV), o, do.to(DEV), lse, dqkv, dqkv[:, d:],
                 dqkv[:, 2 * d:], 3 * d, 3 * d, 3 * d, B, H, L, L, dk, 0.0, 0, 0)
    un = lambda t: t.reshape(B, L, H, dk).transpose(1, 2)                                            # noqa: E731
    close(un(dqkv[:, :d]), qd.grad, 2e-5, 1e-4, "dq (many pairs)")
    close(un(dqkv[:, d:2 * d]), kd.grad, 2e-5, 1e-4, "dk (many pairs)")
    close(un(dqkv[:, 2 * d:]), vd.grad, 2e-5, 1e-4, "dv (many pairs)")


@pytest.mark.parametrize("L", [48, 170])
def test_attention_dropout(ops, L):
    """One-hot V blocks recover the dropped probabilities => the keep mask the FORWARD used (64 keys per run, same
    seed); the backward must regenerate exactly that mask (both of its phases)."""
    B, H, dk, p, seed = 2, 2, 64, 0.25, 99
    d = H * dk
    q, k = rnd(B * L, d, seed=1), rnd(B * L, d, seed=2)
    qg, kg = q.to(DEV), k.to(DEV)
    pd = torch.zeros(B, H, L, L)
    for k0 in range(0, L, dk):
        eye = torch.zeros(B, L, H, dk)
        for i in range(k0, min(L, k0 + dk)):
            eye[:, i, :, i - k0] = 1.0
        o, lse, probs = ops.attn_fwd(qg, kg, eye.view(B * L, d).to(DEV), d, d, d, None, B, H, L, L, dk, p, seed, 3,
                                     want_probs=True)
        n = min(L, k0 + dk) - k0
        pd[..., k0:k0 + n] = o.view(B, L, H, dk).transpose(1, 2)[..., :n].cpu()     # [B,H,L,keys k0..] dropped probs
    pr = probs.cpu()
    keep = pd != 0
    assert abs(keep.float().mean().item() - (1 - p)) < 0.02
    close(pd, torch.where(keep, pr / (1 - p), torch.zeros_like(pr)), 1e-6, 1e-5, "dropped probs")
    # backward with a generic V
    v2 = rnd(B * L, d, seed=4)
    v2g = v2.to(DEV)
    o2, lse2, _ = ops.attn_fwd(qg, kg, v2g, d, d, d, None, B, H, L, L, dk, p, seed, 3)
    qd = q.double().view(B, L, H, dk).transpose(1, 2).requires_grad_()
    kd = k.double().view(B, L, H, dk).transpose(1, 2).requires_grad_()
    vd = v2.double().view(B, L, H, dk).transpose(1, 2).requires_grad_()
    oref, _ = ref_attention(qd, kd, vd, None, 1 / math.sqrt(dk), keep.double(), 1 - p)
    close(o2.view(B, L, H, dk).transpose(1, 2), oref, 1e-5, 1e-5, "dropout attn out")
    do = rnd(B * L, d, seed=5)
    oref.backward(do.double().view(B, L, H, dk).transpose(1, 2))
    dq, dk_, dv = (torch.empty(B * L, d, device=DEV) for _ in range(3))
    ops.attn_bwd(qg, kg, v2g, d, d, d, None, o2, do.to(DEV), lse2, dq, dk_, dv, d, d, d, B, H, L, L, dk, p, seed, 3)
    close(dq.view(B, L, H, dk).transpose(1, 2), qd.grad, 3e-5, 1e-4, "dq (dropout)")
    close(dk_.view(B, L, H, dk).transpose(1, 2), kd.grad, 3e-5, 1e-4, "dk (dropout)")
    close(dv.view(B, L, H, dk).transpose(1, 2), vd.grad, 3e-5, 1e-4, "dv (dropout)")


@pytest.mark.parametrize("n_c", [0, 3])
def test_embed_pe(ops, n_c):
    B, S, d, V = 5, 20, 64, 30
    g = torch.Generator().manual_seed(0)
    tok = torch.randint(0, V, (B, S), generator=g)
    table, pe = rnd(V, d, seed=1), rnd(200, d, seed=2)
    cond = rnd(B, n_c, d, seed=3) if n_c else None
    out = ops.embed_pe_fwd(tok.to(DEV), table.to(DEV), None if cond is None else cond.to(DEV), pe.to(DEV),
                           n_c, math.sqrt(d), 0.0, 0, 0)
    td = table.double().requires_grad_()
    x = F.embedding(tok, td)
    cd = None
    if n_c:
        cd = cond.double().requires_grad_()
        x = torch.cat([cd, x], 1)
    ref = x * math.sqrt(d) + pe.double()[: S + n_c]
    close(out.view(B, S + n_c, d), ref, 1e-6, 1e-6, "embed fwd")
    dout = rnd(B * (S + n_c), d, seed=4)
    ref.backward(dout.double().view(B, S + n_c, d))
    dtable = torch.empty(V, d, device=DEV)
    dcond = torch.empty(B, n_c, d, device=DEV) if n_c else None
    ops.embed_pe_bwd(dout.to(DEV), tok.to(DEV), dtable, dcond, n_c, math.sqrt(d), 0.0, 0, 0)
    close(dtable, td.grad, 1e-4, 1e-5, "embed dtable")
    if n_c:
        close(dcond, cd.grad, 1e-5, 1e-5, "embed dcond")
    # dropout: fwd mask == bwd mask
    p = 0.3
    o2 = ops.embed_pe_fwd(tok.to(DEV), table.to(DEV), None if cond is None else cond.to(DEV), pe.to(DEV),
                          n_c, math.sqrt(d), p, 5, 11)
    keep = (o2 != 0).cpu()
    assert abs(keep.float().mean().item() - (1 - p)) < 0.03
    ones = torch.ones(B * (S + n_c), d, device=DEV)
    dt2 = torch.empty(V, d, device=DEV)
    dc2 = torch.empty(B, n_c, d, device=DEV) if n_c else None
    ops.embed_pe_bwd(ones, tok.to(DEV), dt2, dc2, n_c, 1.0, p, 5, 11)
    exp = torch.zeros(V, d, dtype=torch.double)
    kk = keep.view(B, S + n_c, d)[:, n_c:].double() / (1 - p)
    exp.index_add_(0, tok.reshape(-1), kk.reshape(-1, d))
    close(dt2, exp, 1e-4, 1e-5, "embed bwd dropout mask")


def test_reparam_kld_ce(ops):
    n = (7, 23, 16)
    mu, lv, eps, dz = rnd(*n, seed=1), rnd(*n, seed=2, scale=0.5), rnd(*n, seed=3), rnd(*n, seed=4)
    z, eo = ops.reparam_fwd(mu.to(DEV), lv.to(DEV), eps.to(DEV), 0, 0)
    close(z, eps.double() * torch.exp(0.5 * lv.double()) + mu.double(), 1e-6, 1e-6, "z")
    assert torch.equal(eo.cpu(), eps)
    dmu, dlv = torch.empty(*n, device=DEV), torch.empty(*n, device=DEV)
    ext1, ext2 = rnd(*n, seed=5), rnd(*n, seed=6)
    ops.reparam_bwd(dz.to(DEV), lv.to(DEV), eps.to(DEV), ext1.to(DEV), ext2.to(DEV), dmu, dlv)
    close(dmu, dz.double() + ext1.double(), 1e-6, 1e-6)
    close(dlv, 0.5 * dz.double() * eps.double() * torch.exp(0.5 * lv.double()) + ext2.double(), 1e-6, 1e-5)
    # in-kernel N(0,1)
    big = torch.zeros(1 << 20, device=DEV)
    _, e2 = ops.reparam_fwd(big, big, None, 42, 1)
    assert abs(e2.mean().item()) < 5e-3 and abs(e2.std().item() - 1) < 5e-3
    assert abs((e2 ** 4).mean().item() - 3) < 0.05
    # KLD
    k = ops.kld_fwd(mu.to(DEV), lv.to(DEV))
    md, ld = mu.double().requires_grad_(), lv.double().requires_grad_()
    kr = -0.5 * torch.sum(1 + ld - md.pow(2) - ld.exp())
    close(k, kr, 1e-3, 1e-6, "kld")
    (0.04 * kr).backward()
    gm, gl = ops.kld_bwd(mu.to(DEV), lv.to(DEV), torch.tensor(0.04, device=DEV))
    close(gm, md.grad, 1e-7, 1e-5)
    close(gl, ld.grad, 1e-7, 1e-5)
    # CE (sum, ignore pad)
    rows, V = 333, 30
    logits = rnd(rows, V, seed=7, scale=2.0)
    tgt = torch.randint(0, V, (rows,), generator=torch.Generator().manual_seed(8))
    tgt[::5] = 1
    c = ops.ce_fwd(logits.to(DEV), tgt.to(DEV), 1)
    lg = logits.double().requires_grad_()
    cr = F.cross_entropy(lg, tgt, ignore_index=1, reduction="sum")
    close(c, cr, 1e-3, 1e-6, "ce")
    (1.7 * cr).backward()
    dl = ops.ce_bwd(logits.to(DEV), tgt.to(DEV), torch.tensor(1.7, device=DEV), 1)
    close(dl, lg.grad, 1e-6, 1e-5, "ce bwd")


def test_adam_matches_torch(ops):
    n = 10007
    p0, g = rnd(n, seed=1), rnd(n, seed=2)
    ref = p0.clone().requires_grad_()
    opt = torch.optim.Adam([ref], lr=1e-4, betas=(0.9, 0.98), eps=1e-9)
    pg, m, v = p0.to(DEV).clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 6):
        gi = g * (1 + 0.1 * step)
        ref.grad = gi.clone()
        opt.step()
        ops.adam_step(pg, gi.to(DEV), m, v, 1e-4, 0.9, 0.98, 1e-9, step)
    close(pg, ref, 1e-7, 1e-6, "adam params")
    close(m, opt.state[ref]["exp_avg"], 1e-7, 1e-5)
    close(v, opt.state[ref]["exp_avg_sq"], 1e-9, 1e-5)


def test_small_linear_and_copy_rows(ops):
    B, K, N = 37, 3, 192
    x, w, b, dy = rnd(B, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(B, N, seed=4)
    y = ops.small_linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV))
    close(y, x.double() @ w.double().t() + b.double(), 1e-6, 1e-6)
    dw, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
    ops.small_linear_bwd(dy.to(DEV), x.to(DEV), dw, db)
    close(dw, dy.double().t() @ x.double(), 1e-5, 1e-5)
    close(db, dy.double().sum(0), 1e-5, 1e-5)
    # concat [B,3,d] in front of [B,5,d]
    d = 16
    a, c = rnd(B, 3, d, seed=5), rnd(B, 5, d, seed=6)
    dst = torch.empty(B, 8, d, device=DEV)
    ops.copy_rows(a.to(DEV), 3, 0, dst, 8, 0, B * 3, 3, d)
    ops.copy_rows(c.to(DEV), 5, 0, dst, 8, 3, B * 5, 5, d)
    assert torch.equal(dst.cpu(), torch.cat([a, c], 1))
    s = ops.add(a.to(DEV), a.to(DEV))
    assert torch.equal(s.cpu(), a + a)


# ------------------------------------------------------------------ bf16x6 GEMM mode
def _bf16_to_f32(t_i16):
    return (t_i16.to(torch.int32) << 16).view(torch.float32)


def test_split_planes_exact(ops):
    x = torch.cat([rnd(4096, seed=1), rnd(4096, seed=2, scale=1e-6), rnd(4096, seed=3, scale=3e4),
                   torch.tensor([0.0, -0.0, 1.0, -1.0, 3.1415927, 1e-30, 65504.0, 0.1])]).to(DEV)
    pl = ops.split_planes(x)
    h, m, l = (_bf16_to_f32(pl[i]) for i in range(3))
    assert torch.equal((h + m) + l, x), "hi + mid + lo must reproduce the fp32 value exactly"
    assert (m.abs() <= h.abs() * 2.0 ** -8 + 1e-45).all() and (l.abs() <= h.abs() * 2.0 ** -16 + 1e-45).all()


def _planes_for(ops, ws):
    """Put the weights back to back in one buffer (as flat.py does) and register its planes."""
    flat = torch.cat([w.reshape(-1) for w in ws]).to(DEV).contiguous()
    views, o = [], 0
    for w in ws:
        views.append(flat[o:o + w.numel()].view(w.shape))
        o += w.numel()
    ops.register_planes(flat, ops.split_planes(flat))
    return flat, views


@pytest.mark.parametrize("M,K,nper,nseg", [(300, 512, 512, 3), (1000, 2048, 512, 1), (4099, 512, 2048, 1),
                                           (6400, 64, 520, 1), (2048, 512, 256, 2), (96, 128, 1000, 1),
                                           (3104, 512, 512, 3)])
def test_linear_bf16x6_vs_fp64(ops, M, K, nper, nseg):
    """The bf16x6 kernels against fp64 at the tolerances of the fp32-MFMA kernels, and never worse
    than 2x the fp32 kernels' own error; the launch counter proves which kernels ran."""
    x = rnd(M, K, seed=1)
    ws = [rnd(nper, K, seed=10 + s, scale=K ** -0.5) for s in range(nseg)]
    bs = [rnd(nper, seed=20 + s) for s in range(nseg)]
    N = nper * nseg
    W = torch.cat(ws).double()
    dy = rnd(M, N, seed=3)
    xg, dyg, bg = x.to(DEV), dy.to(DEV), [b.to(DEV) for b in bs]
    dys = [dyg[:, s * nper:] for s in range(nseg)]
    flat, wg = _planes_for(ops, ws)
    res = {}
    try:
        for mode in (ops.GEMM_F32, ops.GEMM_BF16X6):
            ops.gemm_set_mode(mode)
            c0 = ops.gemm_launch_counts()
            y = torch.empty(M, N, device=DEV)
            ops.linear_fwd(xg, wg, bg, [y[:, s * nper:] for s in range(nseg)], N)
            dx = torch.empty(M, K, device=DEV)
            ops.linear_dgrad(dys, N, M, wg, dx)
            dws = [torch.empty(nper, K, device=DEV) for _ in range(nseg)]
            dbs = [torch.empty(nper, device=DEV) for _ in range(nseg)]
            ops.linear_wgrad(dys, N, xg, dws, dbs)
            c1 = ops.gemm_launch_counts()
            res[mode] = (y.cpu().double(), dx.cpu().double(), torch.cat(dws).cpu().double(), torch.cat(dbs).cpu().double(),
                         c1[1] - c0[1])
    finally:
        ops.gemm_set_mode(ops.GEMM_BF16X6)
        ops.unregister_planes(flat)
    refs = (x.double() @ W.t() + torch.cat(bs).double(), dy.double() @ W, dy.double().t() @ x.double(), dy.double().sum(0))
    assert res[ops.GEMM_F32][4] == 0
    x6_launches = res[ops.GEMM_BF16X6][4]
    big = -(-M // 128) * -(-N // 128) >= 192           # smaller forward launches take the skinny-M kernel ...
    small_t, nkt = -(-M // 64) * -(-N // 128), K // 32   # ... or the 64 x 128-tile bf16x6 kernel
    x6s = (-(-M // 128) * -(-N // 256) <= 160 and 96 <= small_t <= (512 if nkt <= 16 else 256) and nkt <= 32 and
           (nseg == 1 or nper % 128 == 0))
    expect = (1 if (x6s or (big and (nseg == 1 or nper % 256 == 0))) else 0) + (1 if N % 32 == 0 else 0) + (1 if M % 32 == 0 else 0)
    assert x6_launches == expect, (x6_launches, expect)
    tols = ((2e-5, 2e-5), (5e-5, 5e-5), (1e-4 * math.sqrt(M / 100 + 1), 1e-4), (1e-4 * math.sqrt(M / 100 + 1), 1e-4))
    for i, what in enumerate(("fwd", "dgrad", "wgrad", "bias grad")):
        close(res[ops.GEMM_BF16X6][i], refs[i], tols[i][0], tols[i][1], f"bf16x6 {what}")
        e6 = (res[ops.GEMM_BF16X6][i] - refs[i]).abs().mean().item()
        e32 = (res[ops.GEMM_F32][i] - refs[i]).abs().mean().item()
        assert e6 <= 2.0 * e32 + 1e-9, f"{what}: bf16x6 mean error {e6:.3e} vs fp32-MFMA {e32:.3e}"


def _adversarial(kind, M, K, N):
    """Operand sets that stress the 3-way bf16 split.  Returns x [M,K], w [N,K], dy [M,N]."""
    x, w, dy = rnd(M, K, seed=31), rnd(N, K, seed=32, scale=K ** -0.5), rnd(M, N, seed=33)
    if kind == "cancel":
        # second half of K cancels the first to 12 bits: the result is 2^-12 of the sum of magnitudes
        x = torch.cat([x, x], 1)
        w = torch.cat([w, -w * (1 + 2.0 ** -12)], 1)
    elif kind == "scales":
        # per-k scale 2^e on x, 2^-e on w (|e| <= 60, ~1e+-18): exact in binary, products unchanged
        e = torch.randint(-60, 61, (K,), generator=torch.Generator().manual_seed(5)).float()
        x, w = x * torch.exp2(e), w * torch.exp2(-e)
        dy = dy * torch.exp2(torch.randint(-60, 61, (N,), generator=torch.Generator().manual_seed(6)).float())
    elif kind == "ones":
        # every significand all ones (2 - 2^-23) * 2^e: h rounds UP, m and l are negative, dropped terms are maximal
        full = 2.0 - 2.0 ** -23
        mk = lambda t, sd: torch.sign(t) * full * torch.exp2(                                     # noqa: E731
            torch.randint(-3, 4, t.shape, generator=torch.Generator().manual_seed(sd)).float())
        x, w, dy = mk(x, 7), mk(w, 8) * K ** -0.5, mk(dy, 9)
    elif kind == "tiny":
        # gradients at the bottom of the fp32 normal range (x, w stay O(1)): the low bf16 pieces of dy leave bf16's
        # normal range and are flushed (documented in gemm_x6.inc: an ABSOLUTE error below 2^-126 per piece)
        dy = dy * 2.0 ** -112
    return x.float(), w.float(), dy.float()


@pytest.mark.parametrize("kind", ["cancel", "scales", "ones", "tiny"])
def test_linear_bf16x6_adversarial_vs_fp64(ops, kind):
    """Worst-case bound stated in gemm_x6.inc: with u = 2^-8 (bf16 unit round-off) the pieces satisfy |m| <= u|x|,
    |l| <= u^2|x|, so the three dropped partial products are bounded by (2u^3 + u^4)|a b| < 2^-23 |a b| per product
    (typical 2^-26); accumulation is fp32 like the fp32-MFMA kernel's, whose own worst relative error on the same
    data (r32 = max |err32| / sum_k |a_k b_k|) stands in for that shared term.  Asserted per output element:
        |err| <= (2^-23 + 1.5 * r32) * sum_k |a_k b_k|  +  flush floor,
    and the mean error never worse than 2x the fp32-MFMA kernel's on the same data (+ the same floor)."""
    M, K0, N = 6144, 512, 512
    x, w, dy = _adversarial(kind, M, K0, N)
    K = x.shape[1]
    b = rnd(N, seed=34) * (0.0 if kind in ("cancel", "tiny") else 1.0)
    xg, dyg, bg = x.to(DEV), dy.to(DEV), b.to(DEV)
    flat, (wg,) = _planes_for(ops, [w])
    res = {}
    try:
        for mode in (ops.GEMM_F32, ops.GEMM_BF16X6):
            ops.gemm_set_mode(mode)
            c0 = ops.gemm_launch_counts()
            y, dx = torch.empty(M, N, device=DEV), torch.empty(M, K, device=DEV)
            dw, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
            ops.linear_fwd(xg, [wg], [bg], [y], N)
            ops.linear_dgrad([dyg], N, M, [wg], dx)
            ops.linear_wgrad([dyg], N, xg, [dw], [db])
            res[mode] = (y.cpu().double(), dx.cpu().double(), dw.cpu().double(), ops.gemm_launch_counts()[1] - c0[1])
    finally:
        ops.gemm_set_mode(ops.GEMM_BF16X6)
        ops.unregister_planes(flat)
    assert res[ops.GEMM_BF16X6][3] == 3 and res[ops.GEMM_F32][3] == 0          # all three ran on the bf16x6 kernels
    X, W, DY = x.double(), w.double(), dy.double()
    refs = (X @ W.t() + b.double(), DY @ W, DY.t() @ X)
    mags = (X.abs() @ W.abs().t() + b.double().abs(), DY.abs() @ W.abs(), DY.abs().t() @ X.abs())
    # pieces below bf16's normal range are flushed: < 2^-126 per piece and product partner
    floors = (3 * 2.0 ** -126 * W.abs().sum(1)[None, :], 3 * 2.0 ** -126 * W.abs().sum(0)[None, :],
              3 * 2.0 ** -126 * (DY.abs().sum(0)[:, None] + X.abs().sum(0)[None, :]))
    for i, what in enumerate(("fwd", "dgrad", "wgrad")):
        e6 = (res[ops.GEMM_BF16X6][i] - refs[i]).abs()
        e32 = (res[ops.GEMM_F32][i] - refs[i]).abs()
        assert torch.isfinite(res[ops.GEMM_BF16X6][i]).all(), what
        r32 = (e32 / mags[i].clamp_min(1e-300)).max().item()
        bound = (2.0 ** -23 + 1.5 * r32) * mags[i] + floors[i]
        worst = (e6 / bound).max().item()
        assert worst <= 1.0, f"{kind} {what}: error {worst:.3f} x the stated bound (r32 = {r32:.3e})"
        assert e6.mean().item() <= 2.0 * e32.mean().item() + floors[i].mean().item() + 1e-300, \
            f"{kind} {what}: bf16x6 mean error {e6.mean().item():.3e} vs fp32-MFMA {e32.mean().item():.3e}"


def test_linear_bf16x6_nonfinite_inputs_documented(ops):
    """gemm_x6.inc range notes, pinned: an infinite operand (or |x| > 3.39e38, which rounds to inf in bf16) gives NaN
    where exact-fp32 arithmetic gives inf (inf - inf in the residual); NaN stays NaN; finite rows are untouched."""
    M, K, N = 6144, 512, 512
    x, w = rnd(M, K, seed=41), rnd(N, K, seed=42, scale=K ** -0.5)
    x[5, 7], x[9, 3], x[11, 1] = float("inf"), 3.4e38, float("nan")
    flat, (wg,) = _planes_for(ops, [w])
    try:
        y = torch.empty(M, N, device=DEV)
        ops.linear_fwd(x.to(DEV), [wg], [torch.zeros(N, device=DEV)], [y], N)
    finally:
        ops.unregister_planes(flat)
    y = y.cpu()
    assert not torch.isfinite(y[5]).any() and not torch.isfinite(y[9]).any() and torch.isnan(y[11]).all()
    keep = torch.ones(M, dtype=torch.bool)
    keep[[5, 9, 11]] = False
    assert torch.isfinite(y[keep]).all()
    close(y[keep], x[keep].double() @ w.double().t(), 2e-5, 2e-5, "finite rows next to non-finite ones")


@pytest.mark.parametrize("M,N", [(640, 2048), (128 * 66 + 40, 1024)])
def test_linear_bf16x6_epilogues_and_dropout_masks(ops, M, N):
    """Fused epilogues in bf16x6 mode: same dropout masks as the fp32 kernels (the mask depends on
    (seed, site, row, col) only), values at the fp32 kernels' tolerance.  The second shape has 268 tiles:
    its last 12 run as a K-split tail launch + fix-up kernel (row_base keeps the dropout coordinates)."""
    K, p, seed = 512, 0.1, 99
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4)
    W2 = rnd(K, N, seed=5, scale=0.05)
    dy = rnd(M, K, seed=6)
    xg, bg, rg, dyg = x.to(DEV), b.to(DEV), r.to(DEV), dy.to(DEV)
    flat, (wg, w2g) = _planes_for(ops, [w, W2])
    out = {}
    try:
        for mode in (ops.GEMM_F32, ops.GEMM_BF16X6):
            ops.gemm_set_mode(mode)
            y1, pre, y2, dpre = (torch.empty(M, N, device=DEV) for _ in range(4))
            ops.linear_fwd(xg, [wg], [bg], [y1], N, epi=ops.EPI_GELU_DROP, pre=pre, p=p, seed=seed, site=3)
            ops.linear_fwd(xg, [wg], [bg], [y2], N, epi=ops.EPI_DROP_RESID, resid=rg, p=p, seed=seed, site=4)
            ops.linear_dgrad([dyg], K, M, [w2g], dpre, depi=ops.DEPI_GELU_BWD, pre=pre, p=p, seed=seed, site=3)
            out[mode] = [t.cpu() for t in (y1, pre, y2, dpre)]
    finally:
        ops.gemm_set_mode(ops.GEMM_BF16X6)
        ops.unregister_planes(flat)
    a, c = out[ops.GEMM_F32], out[ops.GEMM_BF16X6]
    assert torch.equal(a[0] == 0, c[0] == 0) and torch.equal(a[3] == 0, c[3] == 0)
    u = x.double() @ w.double().t() + b.double()
    close(c[1], u, 2e-5, 2e-5, "pre")
    for i, what in enumerate(("gelu+drop", "pre", "drop+resid", "gelu bwd")):
        close(c[i], a[i].double(), 5e-5, 1e-4, what)


@pytest.mark.parametrize("M,K,nper,nseg", [(512, 512, 512, 1), (512, 512, 512, 3), (512, 512, 2048, 1),
                                           (512, 2048, 512, 1), (77, 512, 64, 1), (300, 768, 128, 2)])
def test_linear_panel_kernel_small_problems(ops, M, K, nper, nseg):
    """The decode-step shapes (few 32 x TN tiles, K a multiple of 256) take gemm_f32_panel_kernel: whole reduction in
    one workgroup, fused epilogue, no split-K slabs.  Values vs fp64 for the plain, GELU(+pre) and residual
    epilogues; the same launches with GCT's other skinny route (split-K + fix-up, forced through splitk_ws) agree to
    fp32 rounding."""
    N = nper * nseg
    x = rnd(M, K, seed=1)
    ws = [rnd(nper, K, seed=10 + s, scale=K ** -0.5) for s in range(nseg)]
    bs = [rnd(nper, seed=20 + s) for s in range(nseg)]
    r = rnd(M, N, seed=4)
    xg, rg = x.to(DEV), r.to(DEV)
    wg, bg = [w.to(DEV) for w in ws], [b.to(DEV) for b in bs]
    u = x.double() @ torch.cat(ws).double().t() + torch.cat(bs).double()
    y, pre, y2 = (torch.empty(M, N, device=DEV) for _ in range(3))
    outs = lambda t: [t[:, s * nper:] for s in range(nseg)]                     # noqa: E731
    ops.linear_fwd(xg, wg, bg, outs(y), N)
    close(y, u, 2e-5, 2e-5, "panel: bias")
    if nseg == 1:
        ops.linear_fwd(xg, wg, bg, [y], N, epi=ops.EPI_GELU_DROP, pre=pre, p=0.0, seed=1, site=1)
        close(pre, u, 2e-5, 2e-5, "panel: pre")
        close(y, torch.nn.functional.gelu(u), 2e-5, 2e-5, "panel: gelu")
        ops.linear_fwd(xg, wg, bg, [y2], N, epi=ops.EPI_DROP_RESID, resid=rg, p=0.0, seed=1, site=2)
        close(y2, u + r.double(), 2e-5, 2e-5, "panel: residual")
        wsb = torch.empty(int(ops._L().gct_linear_fwd_ws_bytes(M, K, N)) // 4 + 64, device=DEV)
        y3 = torch.empty(M, N, device=DEV)
        ops.linear_fwd(xg, wg, bg, [y3], N, epi=ops.EPI_DROP_RESID, resid=rg, p=0.0, seed=1, site=2, splitk_ws=wsb)
        close(y3, y2.double(), 2e-5, 2e-5, "panel vs split-K route")


@pytest.mark.parametrize("M,K,nper,nseg", [(4096, 512, 512, 1), (2050, 1024, 520, 1), (3000, 512, 512, 2), (5184, 512, 512, 1),
                                           (2048, 512, 1536, 1), (1500, 64, 1000, 1)])
def test_linear_bf16x6_small_tile_kernel(ops, M, K, nper, nseg):
    """Forward GEMMs with few 128 x 256 tiles but 96-512 tiles of 64 x 128 (decode steps of 1 500-5 000 rows, small
    training batches) take gemm_x6s_kernel: one launch, values vs fp64 at the fp32 kernels' tolerance for the plain,
    GELU(+pre) and dropout+residual epilogues, the dropout mask of the fp32-mode launch, ragged M / N edges and
    two weight segments."""
    N, p, seed = nper * nseg, 0.1, 21
    x = rnd(M, K, seed=1)
    ws = [rnd(nper, K, seed=10 + s, scale=max(K, 64) ** -0.5) for s in range(nseg)]
    bs = [rnd(nper, seed=20 + s) for s in range(nseg)]
    r = rnd(M, N, seed=4)
    xg, rg, bg = x.to(DEV), r.to(DEV), [b.to(DEV) for b in bs]
    flat, wg = _planes_for(ops, ws)
    u = x.double() @ torch.cat(ws).double().t() + torch.cat(bs).double()
    outs = lambda t: [t[:, s * nper:] for s in range(nseg)]                     # noqa: E731
    res = {}
    try:
        for mode in (ops.GEMM_F32, ops.GEMM_BF16X6):
            ops.gemm_set_mode(mode)
            k0, c0 = ops._L().gct_gemm_x6_kernel_launches(), ops.gemm_launch_counts()
            y, pre, y1, y2 = (torch.empty(M, N, device=DEV) for _ in range(4))
            ops.linear_fwd(xg, wg, bg, outs(y), N)
            n_calls = 1
            if nseg == 1:
                ops.linear_fwd(xg, wg, bg, [y1], N, epi=ops.EPI_GELU_DROP, pre=pre, p=p, seed=seed, site=2)
                ops.linear_fwd(xg, wg, bg, [y2], N, epi=ops.EPI_DROP_RESID, resid=rg, p=p, seed=seed, site=3)
                n_calls = 3
            if mode == ops.GEMM_BF16X6:        # one kernel launch per call: no split-K pair, no tail launch
                assert ops._L().gct_gemm_x6_kernel_launches() == k0 + n_calls
                assert ops.gemm_launch_counts()[1] == c0[1] + n_calls
            res[mode] = [t.cpu() for t in (y, pre, y1, y2)]
    finally:
        ops.gemm_set_mode(ops.GEMM_BF16X6)
        ops.unregister_planes(flat)
    a, c = res[ops.GEMM_F32], res[ops.GEMM_BF16X6]
    close(c[0], u, 2e-5, 2e-5, "small-tile bf16x6 vs fp64")
    if nseg == 1:
        close(c[1], u, 2e-5, 2e-5, "pre")
        g_ = torch.nn.functional.gelu(u)
        assert not (((a[2] == 0) != (c[2] == 0)) & (g_.abs() > 1e-4)).any()
        assert not (((a[3] == r) != (c[3] == r)) & (u.abs() > 1e-4)).any()
        close(c[2], a[2].double(), 5e-5, 1e-4, "gelu + dropout")
        close(c[3], a[3].double(), 5e-5, 1e-4, "dropout + residual")


@pytest.mark.parametrize("M,K,N,ldy,bias", [(512, 512, 31, 31, True), (4099, 1024, 28, 40, True), (7, 256, 1, 1, False),
                                            (41000, 512, 30, 30, True), (33, 512, 32, 36, True)])
def test_linear_narrow_output_panel_kernel(ops, M, K, N, ldy, bias):
    """N <= 32 output columns (the vocabulary head, 28-31 columns; the property head, 1 column): the ragged panel
    kernel -- any N, any leading dimension of the output, weight rows beyond N never stored.  Values vs fp64; columns
    of the output buffer beyond N stay untouched."""
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    b = rnd(N, seed=3) if bias else None
    y = torch.full((M, ldy), 7.0, device=DEV)
    ops.linear_fwd(x.to(DEV), [w.to(DEV)], [b.to(DEV) if bias else None], [y], ldy)
    ref = x.double() @ w.double().t() + (b.double() if bias else 0.0)
    close(y[:, :N], ref, 2e-5, 2e-5, "narrow-output panel kernel")
    assert (y[:, N:] == 7.0).all()


def test_linear_bf16x6_split_k_over_the_whole_problem(ops):
    """Few 128x256 tiles and a long reduction (FFN-2 of a decode step with >= 1 024 rows): with a workspace the bf16x6
    kernel runs with K split into slabs + the fix-up kernel (which applies the fused epilogue).  Same dropout mask and
    values as the fp32-mode launch, values vs fp64."""
    M, K, N, p, seed = 1024, 2048, 512, 0.1, 5
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4)
    xg, bg, rg = x.to(DEV), b.to(DEV), r.to(DEV)
    flat, (wg,) = _planes_for(ops, [w])
    ws = torch.empty(int(ops._L().gct_linear_fwd_ws_bytes(M, K, N)) // 4 + 64, device=DEV)
    out = {}
    try:
        for mode in (ops.GEMM_F32, ops.GEMM_BF16X6):
            ops.gemm_set_mode(mode)
            y1, y2 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
            c0, k0 = ops.gemm_launch_counts(), ops._L().gct_gemm_x6_kernel_launches()
            ops.linear_fwd(xg, [wg], [bg], [y1], N, ws=ws)
            ops.linear_fwd(xg, [wg], [bg], [y2], N, epi=ops.EPI_DROP_RESID, resid=rg, p=p, seed=seed, site=4, ws=ws)
            if mode == ops.GEMM_BF16X6:            # both calls took the bf16x6 kernel (one split launch each)
                assert ops.gemm_launch_counts()[1] == c0[1] + 2 and ops._L().gct_gemm_x6_kernel_launches() == k0 + 2
            out[mode] = (y1.cpu(), y2.cpu())
    finally:
        ops.gemm_set_mode(ops.GEMM_BF16X6)
        ops.unregister_planes(flat)
    u = x.double() @ w.double().t() + b.double()
    close(out[ops.GEMM_BF16X6][0], u, 2e-5, 2e-5, "split-K bf16x6 vs fp64")
    a, c = out[ops.GEMM_F32][1], out[ops.GEMM_BF16X6][1]
    assert torch.equal(a == r, c == r)             # same dropped positions (a dropped element leaves the residual)
    close(c, a.double(), 5e-5, 1e-4, "dropout + residual through the fix-up kernel")


def test_dgrad_gelu_bwd_through_quad_map_with_tail_split(ops):
    """GELU-backward dgrad on quad-compacted rows whose pre-activation stays in the forward's row space (read through the
    quad map): at 33 row tiles x 8 column tiles the bf16x6 launch is tail-balanced (head + K-split tail + fix-up), and
    the tail must find its pre-activation rows through row_base, not through a shifted pointer (a batch-128 training
    step hit exactly this: out-of-bounds reads).  Reference: the same call on a gathered copy of the pre-activation."""
    import types
    Mfull, Mc, dff, d, p, seed = 8448, 4224, 2048, 512, 0.1, 11
    g = torch.Generator().manual_seed(5)
    quads = torch.randperm(Mfull // 4, generator=g)[:Mc // 4].sort().values.to(torch.int32)
    live = types.SimpleNamespace(quad_list=quads.to(DEV), Mc=Mc)
    pre = rnd(Mfull, dff, seed=1).to(DEV)
    rows = (quads.long()[:, None] * 4 + torch.arange(4)[None, :]).reshape(-1).to(DEV)
    pre_c = pre[rows].contiguous()
    dy = rnd(Mc, d, seed=2).to(DEV)
    w = rnd(d, dff, seed=3, scale=0.05)
    flat, (wg,) = _planes_for(ops, [w])
    try:
        k0 = ops._L().gct_gemm_x6_kernel_launches()
        a = torch.empty(Mc, dff, device=DEV)
        ops.linear_dgrad([dy], d, Mc, [wg], a, depi=ops.DEPI_GELU_BWD, pre=pre, p=p, seed=seed, site=3, live=live,
                         pre_full=True)
        assert ops._L().gct_gemm_x6_kernel_launches() == k0 + 2          # head + K-split tail
        b = torch.empty(Mc, dff, device=DEV)
        ops.linear_dgrad([dy], d, Mc, [wg], b, depi=ops.DEPI_GELU_BWD, pre=pre_c, p=p, seed=seed, site=3, live=live,
                         pre_full=False)
    finally:
        ops.unregister_planes(flat)
    assert torch.equal(a, b)


def test_wgrad_over_nonzero_row_tiles(ops):
    """gct_nonzero_row_tiles + gct_linear_wgrad_kt: reducing only over the 32-row token tiles whose gradient rows
    are not all zero gives the dense result (the skipped terms are exact zeros)."""
    M, K, N = 32 * 260, 512, 1024
    x = rnd(M, K, seed=1).to(DEV)
    dy = rnd(M, N, seed=2).to(DEV)
    g = torch.Generator().manual_seed(3)
    dead = torch.zeros(M, dtype=torch.bool)
    for s in range(0, M, 81):                       # runs of dead rows like padded target positions
        ln = int(torch.randint(20, 60, (1,), generator=g))
        dead[s + ln:s + 81] = True
    dy[dead.to(DEV)] = 0
    lst, cnt = ops.nonzero_row_tiles(dy)
    live_tiles = (~dead.view(-1, 32).all(1)).nonzero().flatten().to(torch.int32)
    assert int(cnt.item()) == live_tiles.numel() and torch.equal(lst[:live_tiles.numel()].cpu(), live_tiles)
    assert live_tiles.numel() < M // 32
    out = {}
    for name, kt in (("dense", None), ("tiles", (lst, cnt))):
        dw = torch.empty(N, K, device=DEV)
        db = torch.empty(N, device=DEV)
        c0 = ops.gemm_launch_counts()
        ops.linear_wgrad([dy], N, x, [dw], [db], kt=kt)
        assert ops.gemm_launch_counts()[1] == c0[1] + 1          # bf16x6 kernel
        out[name] = (dw.cpu().double(), db.cpu().double())
    ref_w = dy.double().t().cpu() @ x.double().cpu()
    ref_b = dy.double().sum(0).cpu()
    tol = 1e-4 * math.sqrt(M / 100 + 1)
    close(out["tiles"][0], ref_w, tol, 1e-4, "wgrad over listed tiles")
    close(out["tiles"][1], ref_b, tol, 1e-4, "bias grad over listed tiles")
    close(out["tiles"][0], out["dense"][0], 2e-6 * float(ref_w.abs().max()), 1e-5, "listed vs dense (summation split differs)")


@pytest.mark.parametrize("pad", [0, 1, 2, 3])
def test_trg_mask_from_tokens_equals_the_reference_mask(ops, pad):
    """gct_trg_mask_tokens against the ORACLE's get_trg_mask (reference Model/modules.py:17-30, 47-58), nonzero pattern
    bit for bit -- including the reference's `no-peek * pad_idx` quirk (an even pad index blanks the whole mask) -- on a
    contiguous id matrix, on the trainer's trg[:, :-1] view and on sizes that are not multiples of four."""
    from oracle import gct_oracle as O
    g = torch.Generator().manual_seed(pad)
    for B, T in ((5, 21), (3, 81), (7, 13), (1, 1)):
        full = torch.randint(0, 6, (B, T + 1), generator=g)
        for tok in (full[:, :-1], full[:, :-1].contiguous()):
            want = O.get_trg_mask(tok, pad, False, None) != 0
            got = ops.trg_mask_u8(tok.to(DEV), pad)
            assert got.dtype == torch.uint8 and tuple(got.shape) == (B, T, T)
            assert torch.equal(got.cpu() != 0, want), (pad, B, T)


def test_tail_rows_on_the_small_tile_kernel_vs_fp64(ops):
    """Tail-balanced bf16x6 launches whose tail rows run on gemm_x6s_kernel (K <= 1024, <= 512 tiles of 64 x 128): the
    forward with two weight / output segments, the forward with the dropout + residual epilogue (p = 0: comparable with
    fp64), and the
    dgrad mode of the small kernel with a two-segment dY (K' = 1024) accumulating into an existing buffer -- 155 row
    tiles like the decoder's compact rows at batch 512.  Head and tail rows against fp64; exactly two bf16x6 kernel
    launches per call (no K-split, no fix-up)."""
    M, K, nper, nseg = 155 * 128, 512, 512, 2
    N = nper * nseg
    x = rnd(M, K, seed=1)
    ws = [rnd(nper, K, seed=10 + s, scale=K ** -0.5) for s in range(nseg)]
    bs = [rnd(nper, seed=20 + s) for s in range(nseg)]
    resid = rnd(M, N, seed=4)
    flat, wv = _planes_for(ops, ws)
    try:
        xd = x.to(DEV)
        y = torch.empty(M, N, device=DEV)
        outs = [y[:, s * nper:] for s in range(nseg)]
        k0 = ops._L().gct_gemm_x6_kernel_launches()
        ops.linear_fwd(xd, wv, [b.to(DEV) for b in bs], outs, N)
        assert ops._L().gct_gemm_x6_kernel_launches() == k0 + 2
        ref = x.double() @ torch.cat(ws).double().t() + torch.cat(bs).double()
        close(y, ref, 2e-5, 2e-5, "forward (two segments), tail on small tiles")
        y1 = torch.empty(M, nper, device=DEV)                 # fused epilogues take one segment: 155 x 2 tiles
        k0 = ops._L().gct_gemm_x6_kernel_launches()
        ops.linear_fwd(xd, wv[:1], [bs[0].to(DEV)], [y1], nper, epi=ops.EPI_DROP_RESID, resid=resid[:, :nper].contiguous().to(DEV), p=0.0)
        assert ops._L().gct_gemm_x6_kernel_launches() == k0 + 2
        close(y1, ref[:, :nper] + resid[:, :nper].double(), 2e-5, 2e-5, "forward (dropout + residual epilogue), tail on small tiles")
        # dgrad: dX[M, K] (+)= dY[M, 2 x 512] @ [W0; W1]  -- N(out) = 512 -> 155 x 2 tiles = 256 + 54
        dy = rnd(M, N, seed=5).to(DEV)
        dys = [dy[:, s * nper:] for s in range(nseg)]
        base = rnd(M, K, seed=6)
        dx = base.to(DEV).clone()
        k0 = ops._L().gct_gemm_x6_kernel_launches()
        ops.linear_dgrad(dys, N, M, wv, dx, depi=ops.DEPI_ACCUM)
        assert ops._L().gct_gemm_x6_kernel_launches() == k0 + 2
        refd = base.double() + dy.cpu().double() @ torch.cat(ws).double()
        close(dx, refd, 4e-5, 2e-5, "dgrad, tail on small tiles")
    finally:
        ops.unregister_planes(flat)


@pytest.mark.parametrize("M,Kp,N,depi,launches", [(5120, 512, 512, "store", 1),      # 80 tiles of 128 x 256 -> 320 small tiles
                                                  (5120, 2048, 512, "gelu", 1),       # long reduction over 80 tiles: K split 3 ways + fix-up
                                                  (2560, 1536, 512, "accum", 1)])     # three dY segments, 40 tiles, K split 6 ways
def test_dgrad_routes_for_few_tiles_vs_fp64(ops, M, Kp, N, depi, launches):
    """dgrad launches of a training step at batch 32-64: too few 128 x 256 tiles to fill the chip.  K' <= 1024 takes the
    small-tile kernel's dgrad mode, a long reduction takes the K-split-over-the-whole-problem route (both bf16x6: one
    kernel launch counted).  Against fp64, with the GELU-backward and accumulate epilogues."""
    nseg = 3 if Kp == 1536 else 1
    nper = Kp // nseg
    dy = rnd(M, Kp, seed=1)
    ws = [rnd(nper, N, seed=10 + s, scale=Kp ** -0.5) for s in range(nseg)]
    pre = rnd(M, N, seed=3)
    base = rnd(M, N, seed=4)
    flat, wv = _planes_for(ops, ws)
    try:
        dyd = dy.to(DEV)
        dys = [dyd[:, s * nper:] for s in range(nseg)]
        dx = base.to(DEV).clone()
        k0 = ops._L().gct_gemm_x6_kernel_launches()
        kw = {"store": {}, "gelu": dict(depi=ops.DEPI_GELU_BWD, pre=pre.to(DEV), p=0.0),
              "accum": dict(depi=ops.DEPI_ACCUM)}[depi]
        ops.linear_dgrad(dys, Kp, M, wv, dx, **kw)
        assert ops._L().gct_gemm_x6_kernel_launches() == k0 + launches
    finally:
        ops.unregister_planes(flat)
    ref = dy.double() @ torch.cat(ws).double()
    if depi == "gelu":
        u = pre.double()
        ref = ref * (0.5 * (1 + torch.erf(u / math.sqrt(2))) + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi))
    elif depi == "accum":
        ref = ref + base.double()
    close(dx, ref, 4e-5, 2e-5, f"dgrad {depi} M={M} K'={Kp}")
