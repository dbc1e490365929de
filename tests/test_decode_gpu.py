"""KV-cached decode parity: token ids must equal (a) the reference's greedy ids frozen in
tests/golden/g5_decode.pt and (b) the reference-style loop over the un-cached model.decode."""
import os

import pytest
import torch

from gct_plus_amd import synthetic

pytestmark = pytest.mark.gpu
TINY = dict(N=2, d_model=64, dff=128, h=4, latent_dim=16)


def build(mtype, full=False, seed=1):
    from gct_plus_amd.Model import model_dict
    vs, vt = synthetic.vocab_sizes(mtype)
    kw = dict(N=6, d_model=512, dff=2048, h=8, latent_dim=128) if full else TINY
    torch.manual_seed(seed)
    return model_dict[mtype](vs, vt, dropout=0.1, nconds=synthetic.n_conds(mtype), use_cond2lat=True,
                             **kw).cuda().eval()


def test_kv_decode_matches_reference_golden_ids(golden_dir):
    from gct_plus_amd.decode import KVDecoder
    g5 = torch.load(os.path.join(golden_dir, "g5_decode.pt"), weights_only=True)
    for mtype, fx in g5.items():
        model = build(mtype)
        z = fx["z"].cuda()
        dconds = fx["dconds"].cuda() if fx["dconds"] is not None else None
        n, L = z.shape[0], z.shape[1]
        src_mask = torch.ones(n, 1, L, dtype=torch.bool, device="cuda")
        kd = KVDecoder(model, synthetic.PAD_ID, synthetic.SOS_ID, eos_id=-1)   # fixture ran all 15 steps
        kd.start(z, src_mask, dconds, max_total_len=32)
        ys0 = torch.full((n, 1), synthetic.SOS_ID, dtype=torch.long, device="cuda")
        ys = kd.generate(ys0, max_strlen=16)
        assert torch.equal(ys.cpu(), fx["ys"]), mtype
        assert torch.allclose(kd.buf["logits"].cpu(), fx["last_logits"], atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("mtype,graphs", [("vaetf", False), ("pscavaetf", False), ("vaetf", True), ("pscavaetf", True)])
def test_kv_decode_matches_uncached_full_size(mtype, graphs):
    """Full-size model, ragged source masks, a scaffold-style prefix for pscavaetf, early <eos> stop."""
    from gct_plus_amd.decode import KVDecoder, reference_style_decode
    model = build(mtype, full=True, seed=3)
    nc = synthetic.n_conds(mtype)
    n, Le = 24, 40 + nc
    g = torch.Generator().manual_seed(11)
    z = torch.randn(n, Le, 128, generator=g).cuda()
    dconds = torch.randn(n, nc, generator=g).cuda() if nc else None
    lens = torch.randint(10, Le + 1, (n,), generator=g)
    src_mask = (torch.arange(Le)[None, :] < lens[:, None]).unsqueeze(1).cuda()
    if nc:
        pre = torch.randint(5, 30, (n, 6), generator=g)
        ys0 = torch.cat([torch.full((n, 1), synthetic.SOS_ID), pre, torch.full((n, 1), 4)], 1).cuda()
    else:
        ys0 = torch.full((n, 1), synthetic.SOS_ID, dtype=torch.long, device="cuda")
    ref = reference_style_decode(model, z, src_mask, dconds, ys0, synthetic.PAD_ID, synthetic.EOS_ID, 40)
    kd = KVDecoder(model, synthetic.PAD_ID, synthetic.SOS_ID, synthetic.EOS_ID)
    kd.start(z, src_mask, dconds, max_total_len=64)
    ys = kd.generate(ys0, max_strlen=40, use_graphs=graphs)
    assert ys.shape == ref.shape, (ys.shape, ref.shape)
    assert torch.equal(ys, ref)


@pytest.mark.parametrize("graphs", [False, True])
def test_kv_decode_full_size_pscavaetf_vs_oracle_loop(graphs):
    """BASELINE configs[4]'s model type at full size against the ORACLE's restated loop (the un-cached CPU decoder of
    Inference/sampling_tool.py:140-184 with the scaffold prefix of :452-498), not only against the un-cached HIP loop:
    n = 4, 79 generated tokens, no early stop.  Ids must be equal; where they are not, the oracle's two best logits at
    the first differing step must be an fp32 tie (gap < 1e-4 at a logit scale of ~5) -- both sides take the argmax of
    fp32 logits accumulated in different orders."""
    from oracle import gct_oracle as O
    from gct_plus_amd.decode import KVDecoder
    mtype = "pscavaetf"
    model = build(mtype, full=True, seed=3)
    vs, vt = synthetic.vocab_sizes(mtype)
    nc = synthetic.n_conds(mtype)
    n, Le = 4, 40 + nc
    g = torch.Generator().manual_seed(31)
    z = torch.randn(n, Le, 128, generator=g)
    dconds = torch.randn(n, nc, generator=g)
    src_mask = torch.ones(n, 1, Le, dtype=torch.bool)
    pre = torch.randint(5, 30, (n, 6), generator=g)
    ys0 = torch.cat([torch.full((n, 1), synthetic.SOS_ID), pre, torch.full((n, 1), 4)], 1)
    cfg = O.make_cfg(mtype, vs, vt, dropout=0.0, nconds=nc, use_cond2lat=True)
    P = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    trace = []
    ref = O.greedy_decode(P, cfg, z, src_mask, dconds, synthetic.SOS_ID, -1, synthetic.PAD_ID, max_strlen=80,
                          ys0=ys0, trace=trace)
    kd = KVDecoder(model, synthetic.PAD_ID, synthetic.SOS_ID, eos_id=-1)
    kd.start(z.cuda(), src_mask.cuda(), dconds.cuda(), max_total_len=96)
    ys = kd.generate(ys0.cuda(), max_strlen=80, use_graphs=graphs, check_every=0).cpu()
    assert ys.shape == ref.shape == (n, ys0.shape[1] + 79)
    for b in range(n):
        diff = (ys[b] != ref[b]).nonzero()
        if diff.numel() == 0:
            continue
        t = int(diff[0]) - ys0.shape[1]                 # first differing generated token of this sample
        top2 = trace[t][b].topk(2).values
        assert float(top2[0] - top2[1]) < 1e-4, (b, t, top2.tolist(), ys[b].tolist(), ref[b].tolist())


def test_folded_cross_attention_is_cached_until_the_weights_change():
    """KVDecoder._fold_weights (the weights-only part of the latent cross-attention fold) runs once per set of weights:
    a second start() reuses it; an in-place torch update of a parameter (version counter) or a FusedAdam step (raw
    kernel write, reported through invalidate_weight_planes) triggers a re-fold, and the ids follow the new weights."""
    from gct_plus_amd.decode import KVDecoder, reference_style_decode
    from gct_plus_amd.optim import FusedAdam
    mtype = "pvaetf"
    model = build(mtype)
    nc = synthetic.n_conds(mtype)
    n, Le = 6, 9 + nc
    g = torch.Generator().manual_seed(3)
    z = torch.randn(n, Le, 16, generator=g).cuda()
    dconds = torch.randn(n, nc, generator=g).cuda()
    src_mask = torch.ones(n, 1, Le, dtype=torch.bool, device="cuda")
    ys0 = torch.full((n, 1), synthetic.SOS_ID, dtype=torch.long, device="cuda")
    kd = KVDecoder(model, synthetic.PAD_ID, synthetic.SOS_ID, eos_id=-1)
    calls = []
    real = kd._fold_weights
    kd._fold_weights = lambda: (calls.append(1), real())[1]

    def run():
        kd.start(z, src_mask, dconds, max_total_len=32)
        assert kd.zattn
        ys = kd.generate(ys0, max_strlen=14)
        ref = reference_style_decode(model, z, src_mask, dconds, ys0, synthetic.PAD_ID, -1, 14)
        assert torch.equal(ys, ref)

    run()
    run()
    assert len(calls) == 1                                          # same weights: folded once
    with torch.no_grad():
        model.decoder.layers[0].attn_2.k_linear.weight.mul_(1.5)    # torch in-place update: version counter
    run()
    assert len(calls) == 2
    model.train()
    opt = FusedAdam(model.parameters(), lr=1e-2, betas=(0.9, 0.98), eps=1e-9, model=model)
    for p in model.parameters():
        p.grad = torch.randn_like(p) * 0.1
    opt.step()                                                      # raw kernel write behind torch's back
    model.eval()
    run()
    assert len(calls) == 3


def test_multinomial_matches_probabilities():
    """Sampling cannot share an RNG stream with torch.multinomial; compare at the probability
    level: empirical frequencies of the first sampled token vs softmax(logits)."""
    from gct_plus_amd import ops
    n, V = 4096, 30
    logits = torch.randn(1, V, generator=torch.Generator().manual_seed(0)).repeat(n, 1).cuda().contiguous()
    ys = torch.zeros(n, 2, dtype=torch.int64, device="cuda")
    valid = torch.zeros(n, 2, dtype=torch.uint8, device="cuda")
    done = torch.zeros(n, dtype=torch.uint8, device="cuda")
    probs = torch.empty(n, V, device="cuda")
    ops.select_token(logits, ys, 1, valid, done, 1, synthetic.PAD_ID, synthetic.EOS_ID, seed=123, probs_out=probs)
    p = torch.softmax(logits[0].double().cpu(), -1)
    assert torch.allclose(probs[0].double().cpu(), p, atol=1e-6)
    freq = torch.bincount(ys[:, 1].cpu(), minlength=V).double() / n
    assert float((freq - p).abs().max()) < 0.03
    assert torch.equal(valid[:, 1].cpu().bool(), ys[:, 1].cpu() != synthetic.PAD_ID)
    assert torch.equal(done.cpu().bool(), ys[:, 1].cpu() == synthetic.EOS_ID)


def test_sampling_front_end_all_model_types():
    """sample_smiles of every model type runs on the KV-cached decoder and returns the strings of
    the token ids the un-cached reference-style loop produces for the same z / prefix."""
    from gct_plus_amd import data
    from gct_plus_amd.Inference.sampling_tool import get_sampler, sample_token_lengths
    from gct_plus_amd.decode import reference_style_decode
    from tests.test_data_pipeline import SMILES
    import numpy as np
    lens = sample_token_lengths([20, 25, 25, 30, 31, 40, 22, 28], 500, np.random.default_rng(0))
    assert 15 <= lens.min() and lens.max() <= 45 and abs(lens.mean() - 27.6) < 2.5
    for mtype in ("vaetf", "pvaetf", "scavaetf", "pscavaetf"):
        sep = mtype in ("scavaetf", "pscavaetf")
        strs = [("c1ccccc1<sep>" + s) if sep else s for s in SMILES]
        SRC, TRG = data.Vocab.build(strs, False, sep), data.Vocab.build(strs, True, sep)
        from gct_plus_amd.Model import model_dict
        nc = synthetic.n_conds(mtype)
        torch.manual_seed(4)
        model = model_dict[mtype](len(SRC), len(TRG), dropout=0.1, nconds=nc, use_cond2lat=True, **TINY).cuda().eval()
        sp = get_sampler(mtype, model, SRC, TRG, latent_dim=16, max_strlen=24, cond_dim=nc,
                         toklen_data=[12, 14, 15, 18, 20, 16])
        n = 6
        args = {"vaetf": (n,), "pvaetf": (np.zeros((n, 3)),), "scavaetf": (n, "c1ccccc1"),
                "pscavaetf": (np.ones((n, 3)) * 0.3, "c1ccccc1")}[mtype]
        kw = {} if nc == 0 else {"transform": False}
        smiles, toklen, toklen_gen = sp.sample_smiles(*args, **kw)
        assert len(smiles) == n and all(isinstance(s, str) for s in smiles) and len(toklen) == n
        ys = sp.kv.ys[:, :sp.kv.ys.shape[1]]
        # same z again through the un-cached loop
        z = None
        if mtype in ("vaetf", "pvaetf"):
            z = torch.randn(n, 20 + nc, 16)
            a2 = (n,) if mtype == "vaetf" else args
            s1, _, _ = sp.sample_smiles(*a2, zs=z, **kw)
            src_mask = torch.ones(n, 1, z.size(1), dtype=torch.bool, device="cuda")
            dc = None if nc == 0 else torch.zeros(n, 3, device="cuda")
            ref = reference_style_decode(model, z.cuda(), src_mask, dc, sp.init_y(n).cuda(), sp.pad_id, sp.eos_id, 24)
            assert s1 == [sp.id_to_smi(r) for r in ref.cpu().numpy()]
        # encode path (Sampling.encode_smiles -> model.encode)
        if mtype == "vaetf":
            zz, mu, lv = sp.encode_smiles(SMILES[:3])
            assert mu.shape[0] == 3 and mu.shape[2] == 16


def test_kv_decode_long_scaffold_prefix():
    """Caches and cross-attention keys beyond 128 positions (a long scaffold prefix + a 150-row latent): the cached
    path equals the un-cached reference-style loop token for token (attention kernels and gct_attn_decode cover the
    reference's whole positional table, Model/modules.py:117)."""
    from gct_plus_amd.decode import KVDecoder, reference_style_decode
    mtype = "pscavaetf"
    model = build(mtype)
    n, Le = 5, 150 + 3
    g = torch.Generator().manual_seed(23)
    z = torch.randn(n, Le, TINY["latent_dim"], generator=g).cuda()
    dconds = torch.randn(n, 3, generator=g).cuda()
    lens = torch.randint(100, Le + 1, (n,), generator=g)
    src_mask = (torch.arange(Le)[None, :] < lens[:, None]).unsqueeze(1).cuda()
    pre = torch.randint(5, 30, (n, 118), generator=g)
    ys0 = torch.cat([torch.full((n, 1), synthetic.SOS_ID), pre, torch.full((n, 1), 4)], 1).cuda()      # 120 tokens
    ref = reference_style_decode(model, z, src_mask, dconds, ys0, synthetic.PAD_ID, -1, 50)
    kd = KVDecoder(model, synthetic.PAD_ID, synthetic.SOS_ID, -1)
    kd.start(z, src_mask, dconds, max_total_len=176)
    ys = kd.generate(ys0, max_strlen=50)
    assert ys.shape[1] == 169 and torch.equal(ys, ref)


def build_c2d(mtype="pvaetf", seed=2):
    from gct_plus_amd.Model import model_dict
    vs, vt = synthetic.vocab_sizes(mtype)
    torch.manual_seed(seed)
    return model_dict[mtype](vs, vt, dropout=0.1, nconds=3, use_cond2dec=True, use_cond2lat=False, **TINY).cuda().eval()


@pytest.mark.parametrize("graphs", [False, True])
def test_kv_decode_use_cond2dec(graphs):
    """-use_cond2dec (reference Model/vaetf.py:83-86, Model/modules.py:19-26, Inference/sampling_tool.py:151-160): the
    n_c condition tokens sit in front of the decoder stream, see each other and the first token; the cached decoder
    prefills them with the prefix and must reproduce the un-cached reference-style loop token for token."""
    from gct_plus_amd.decode import KVDecoder, reference_style_decode
    from gct_plus_amd.Model.modules import get_trg_mask
    model = build_c2d()
    n, Le = 7, 19
    g = torch.Generator().manual_seed(3)
    z = torch.randn(n, Le, TINY["latent_dim"], generator=g).cuda()
    dconds = torch.randn(n, 3, generator=g).cuda()
    lens = torch.randint(8, Le + 1, (n,), generator=g)
    src_mask = (torch.arange(Le)[None, :] < lens[:, None]).unsqueeze(1).cuda()
    ys0 = torch.full((n, 1), synthetic.SOS_ID, dtype=torch.long, device="cuda")
    # the reference-style loop for cond2dec: block mask, logits of the token rows only
    ys = ys0.clone()
    for _ in range(29):
        tm = get_trg_mask(ys, synthetic.PAD_ID, True, dconds)
        logits = model.decode(ys, z, src_mask, tm, dconds)[:, 3:]
        ys = torch.cat([ys, logits[:, -1].argmax(-1)[:, None]], dim=1)
    kd = KVDecoder(model, synthetic.PAD_ID, synthetic.SOS_ID, -1)
    kd.start(z, src_mask, dconds, max_total_len=40)
    out = kd.generate(ys0, max_strlen=30, use_graphs=graphs)
    assert torch.equal(out, ys)


def test_graph_replay_survives_restarts_with_other_shapes():
    """ADVICE r1 (decode.py): graphs are captured against the decoder's buffers.  A second sample call with another
    batch size reallocates them -- the stale graphs must go (they hold raw pointers); with the same geometry they are
    reused; the multinomial seed is read from device memory, so a new seed needs no new graph."""
    from gct_plus_amd.decode import KVDecoder
    model = build("scavaetf")
    kd_g = KVDecoder(model, synthetic.PAD_ID, synthetic.SOS_ID, synthetic.EOS_ID)
    kd_e = KVDecoder(model, synthetic.PAD_ID, synthetic.SOS_ID, synthetic.EOS_ID)
    g = torch.Generator().manual_seed(9)
    ngraphs = []
    for call, (n, Le, t0) in enumerate([(6, 14, 5), (6, 14, 5), (11, 20, 3), (6, 14, 5)]):
        z = torch.randn(n, Le, TINY["latent_dim"], generator=g).cuda()
        src_mask = torch.ones(n, 1, Le, dtype=torch.bool, device="cuda")
        pre = torch.randint(5, 30, (n, t0 - 1), generator=g)
        ys0 = torch.cat([torch.full((n, 1), synthetic.SOS_ID), pre], 1).cuda()
        for algo in ("greedy", "multinomial"):
            outs = []
            for kd, graphs in ((kd_g, True), (kd_e, False)):
                kd.start(z, src_mask, None, max_total_len=40)
                outs.append(kd.generate(ys0, max_strlen=20, algo=algo, seed=100 + call, use_graphs=graphs))
            assert outs[0].shape == outs[1].shape and torch.equal(outs[0], outs[1]), (call, algo)
        ngraphs.append(len(kd_g.graphs))
    assert ngraphs == [2, 2, 2, 2]          # one graph per selection mode, re-captured only when the geometry changed


@pytest.mark.parametrize("mtype", ["vaetf", "pvaetf"])
def test_kv_decode_source_masks_prefix_holes_and_empty(mtype):
    """The cross-attention of a decode step reads only the leading visible memory rows when a sample's source mask is
    a non-empty prefix (KVDecoder.src_klen); masks with holes, left-padded masks and a sample that sees NO memory row
    (uniform attention over all of them) keep every row.  Token ids equal the un-cached reference-style loop."""
    from gct_plus_amd.decode import KVDecoder, reference_style_decode
    model = build(mtype, seed=5)
    nc = synthetic.n_conds(mtype)
    n, Le = 12, 24
    g = torch.Generator().manual_seed(3)
    z = torch.randn(n, Le, 16, generator=g).cuda()
    dconds = torch.randn(n, nc, generator=g).cuda() if nc else None
    lens = torch.randint(3, Le + 1, (n,), generator=g)
    m = torch.arange(Le)[None, :] < lens[:, None]                       # prefixes
    m[1] = torch.rand(Le, generator=g) < 0.5                             # holes
    m[2] = torch.arange(Le) >= Le - 5                                    # left-padded
    m[3] = False                                                         # sees nothing
    m[4] = True                                                          # sees everything
    src_mask = m.unsqueeze(1).cuda()
    ys0 = torch.full((n, 1), synthetic.SOS_ID, dtype=torch.long, device="cuda")
    ref = reference_style_decode(model, z, src_mask, dconds, ys0, synthetic.PAD_ID, -1, 20)
    kd = KVDecoder(model, synthetic.PAD_ID, synthetic.SOS_ID, eos_id=-1)
    kd.start(z, src_mask, dconds, max_total_len=32)
    ys = kd.generate(ys0, max_strlen=20, use_graphs=True)
    assert torch.equal(ys, ref)
    klen = kd.src_klen.cpu()
    Lk = Le + nc
    assert int(klen[0]) == int(lens[0]) + nc and int(klen[1]) == Lk and int(klen[4]) == Lk
    assert int(klen[3]) == (nc if nc else Lk)      # nothing visible: all rows (uniform); only the condition rows: those
    # left-padded: with the condition rows in front (pvaetf) it is not a prefix either
    assert int(klen[2]) == Lk


@pytest.mark.parametrize("nc,lat,H,dk,Le", [(0, 128, 8, 64, 80), (3, 128, 8, 64, 77), (0, 16, 4, 16, 19), (3, 32, 2, 32, 5),
                                            (0, 64, 4, 16, 200), (3, 8, 4, 16, 1)])
def test_attn_decode_z_equals_attention_over_projected_memory(nc, lat, H, dk, Le):
    """gct_attn_decode_z (cross-attention of a decode step over the latent rows, keys / values never projected) against
    the reference formulation in fp64: memory = [cond2lat rows ; fc_z(z)], k = W_k mem + b_k, v = W_v mem + b_v,
    softmax(q k^T / sqrt(dk), masked_fill -1e9) v (Model/sublayers.py:29-41).  Masks: full, prefix (with klen), holes, and a
    sample that sees no key (uniform over all keys, like the reference)."""
    from gct_plus_amd import ops
    d, n = H * dk, 9
    g = torch.Generator().manual_seed(100 + lat + Le)
    rd = lambda *s, sc=1.0: (torch.randn(*s, generator=g, dtype=torch.float64) * sc)       # noqa: E731
    Wz, bz = rd(d, lat, sc=lat ** -0.5), rd(d, sc=0.3)
    Wk, bk, Wv, bv = rd(d, d, sc=d ** -0.5), rd(d, sc=0.3), rd(d, d, sc=d ** -0.5), rd(d, sc=0.3)
    q, z, cond = rd(n, d), rd(n, Le, lat), rd(n, nc, d)
    Lk = nc + Le
    valid = torch.ones(n, Lk, dtype=torch.uint8)
    klen = torch.full((n,), Lk, dtype=torch.int32)
    for b in range(n):
        kind = b % 4
        if kind == 1:                                  # visible prefix: the kernel may stop at klen
            ln = int(torch.randint(nc + (0 if nc else 1), Lk + 1, (1,), generator=g))
            ln = max(ln, 1)
            valid[b, ln:] = 0
            klen[b] = ln
        elif kind == 2:                                # holes: every row is read
            valid[b, nc:] = (torch.rand(Le, generator=g) < 0.6).to(torch.uint8)
        elif kind == 3 and nc == 0:                    # no visible key at all
            valid[b] = 0
    mem = torch.cat([cond, z @ Wz.t() + bz], 1)                                  # [n, Lk, d]
    k = (mem @ Wk.t() + bk).view(n, Lk, H, dk).transpose(1, 2)
    v = (mem @ Wv.t() + bv).view(n, Lk, H, dk).transpose(1, 2)
    s = (q.view(n, H, 1, dk) @ k.transpose(-1, -2)) / dk ** 0.5                   # [n, H, 1, Lk]
    s = s.masked_fill(valid.view(n, 1, 1, Lk) == 0, -1e9)
    ref = (s.softmax(-1) @ v).transpose(1, 2).reshape(n, d)
    # the folding (decode.KVDecoder._fold_cross), here in fp64
    G, Hm = Wk @ Wz, Wv @ Wz                                                      # [d, lat]
    c, dv = Wk @ bz + bk, Wv @ bz + bv
    qf = torch.einsum("hrc,nhr->nhc", G.view(H, dk, lat), q.view(n, H, dk)).reshape(n, H * lat)
    qbuf = torch.cat([q, qf], 1).float().contiguous().cuda()                      # plain q | folded q
    ckv = None
    if nc:
        ck = cond @ Wk.t() + bk - c
        cvv = cond @ Wv.t() + bv - dv
        ckv = torch.cat([ck, cvv], 2).view(n * nc, 2 * d).float().contiguous().cuda()
    out = torch.zeros(n, d + H * lat, device="cuda")
    ops.attn_decode_z(qbuf, d, z.float().contiguous().cuda(), ckv, nc, valid.cuda(), out, d, n, H, dk, klen=klen.cuda())
    ctx = out[:, d:].cpu().double().view(n, H, lat)
    o = torch.einsum("hrc,nhc->nhr", Hm.view(H, dk, lat), ctx).reshape(n, d) + dv
    if nc:
        o = o + out[:, :d].cpu().double()
    err = (o - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err


def test_kv_decode_latent_cross_attention_vs_projected_keys():
    """The two cross-attention forms of the cached decoder (over the latent rows / over projected K, V): same token ids and
    logits within fp32 rounding at full size, with cond2lat condition rows in the memory."""
    from gct_plus_amd import decode
    from gct_plus_amd.decode import KVDecoder
    model = build("pscavaetf", full=True, seed=5)
    n, Le = 48, 33
    g = torch.Generator().manual_seed(2)
    z = torch.randn(n, Le, 128, generator=g).cuda()
    dconds = torch.randn(n, 3, generator=g).cuda()
    lens = torch.randint(5, Le + 1, (n,), generator=g)
    src_mask = (torch.arange(Le)[None, :] < lens[:, None]).unsqueeze(1).cuda()
    ys0 = torch.full((n, 1), synthetic.SOS_ID, dtype=torch.long, device="cuda")
    res = {}
    keep = decode.ZATTN
    try:
        for zz in (False, True):
            decode.ZATTN = zz
            kd = KVDecoder(model, synthetic.PAD_ID, synthetic.SOS_ID, -1)
            kd.start(z, src_mask, dconds, max_total_len=40)
            assert kd.zattn == zz
            res[zz] = (kd.generate(ys0, max_strlen=30).clone(), kd.buf["logits"].clone())
    finally:
        decode.ZATTN = keep
    assert torch.equal(res[False][0], res[True][0])
    assert torch.allclose(res[False][1], res[True][1], atol=1e-4, rtol=1e-4)
