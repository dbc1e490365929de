"""Pins oracle/gct_oracle.py to the reference through the fixtures produced by
tests/golden/make_golden.py (which imports the real reference in the build
container).  CPU only."""
import hashlib
import json
import os

import pytest
import torch

from oracle import gct_oracle as O
from gct_plus_amd import synthetic

TYPES = ["vaetf", "pvaetf", "scavaetf", "pscavaetf"]


def sha(t):
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()


def load_g2(golden_dir, mtype):
    return torch.load(os.path.join(golden_dir, f"g2_{mtype}.pt"), weights_only=True)


def cfg_from(fx):
    vs, vt = synthetic.vocab_sizes(fx["model_type"])
    c = fx["cfg"]
    return O.make_cfg(fx["model_type"], vs, vt, N=c["N"], d_model=c["d_model"], dff=c["dff"],
                      h=c["h"], latent_dim=c["latent_dim"], dropout=c["dropout"],
                      nconds=c["nconds"], use_cond2dec=c["use_cond2dec"],
                      use_cond2lat=c["use_cond2lat"])


def test_known_answers(golden_dir):
    g1 = json.load(open(os.path.join(golden_dir, "g1_known_answers.json")))
    y = O.norm(torch.tensor([[1.0, 2.0, 3.0, 4.0]]), torch.ones(4), torch.zeros(4))
    assert torch.allclose(y, torch.tensor(g1["norm4_1234"]), atol=1e-7)
    # SURVEY 8(a) a10 known answer
    assert torch.allclose(y[0], torch.tensor([-1.161894, -0.387298, 0.387298, 1.161894]), atol=1e-6)
    assert O.nopeak_mask(3, False, 1, 0).tolist() == g1["nopeak_3_F_1_0"]
    assert O.nopeak_mask(2, True, 1, 2).tolist() == g1["nopeak_2_T_1_2"]
    assert O.nopeak_mask(3, False, 1, 0).dtype == torch.int64
    pe = O.positional_table(512)
    assert sha(pe) == g1["pe512_sha256"]
    assert sha(O.positional_table(64)) == g1["pe64_sha256"]
    assert pe[0, 1, :4].tolist() == g1["pe512_pos1_first4"]
    assert torch.allclose(pe[0, 79, :4], torch.tensor([-0.44411266, 0.68946636, -0.95164901, -0.22938789]), atol=1e-7)
    x = torch.tensor(g1["norm12_in"])
    y = O.norm(x, torch.linspace(0.5, 1.5, 12), torch.linspace(-1, 1, 12))
    assert torch.allclose(y, torch.tensor(g1["norm12_out"]), atol=1e-6)


@pytest.mark.parametrize("mtype", TYPES)
def test_init_order_parity_and_keys(golden_dir, mtype):
    fx = load_g2(golden_dir, mtype)
    cfg = cfg_from(fx)
    st = O.init_state(cfg, seed=1)
    assert list(st.keys()) == list(fx["init_sha256"].keys())          # state_dict layout
    for k, v in st.items():
        assert sha(v) == fx["init_sha256"][k], k                      # bit-identical init
    assert O.param_names(cfg) == fx["param_order"]                   # Adam state index order
    assert len(st) == {"vaetf": 104, "pvaetf": 104, "scavaetf": 100, "pscavaetf": 104}[mtype]


@pytest.mark.parametrize("mtype", TYPES)
def test_forward_loss_grads(golden_dir, mtype):
    fx = load_g2(golden_dir, mtype)
    cfg = cfg_from(fx)
    P = O.make_leaves(O.init_state(cfg, seed=1))
    b = fx["batch"]
    src_mask, trg_mask, trg_in = O.batch_masks(cfg, b, synthetic.PAD_ID)
    prop, mol, mu, lv, z = O.forward(P, cfg, b["src"], trg_in, src_mask, trg_mask,
                                     b.get("econds"), b.get("dconds"), eps=fx["eps"], train=True)
    assert (prop is None) == fx["prop_is_none"]
    for got, key in ((mol, "logits"), (mu, "mu"), (lv, "log_var"), (z, "z")):
        assert torch.allclose(got, fx[key], atol=2e-6, rtol=1e-6), key
    assert torch.equal(mol.argmax(-1), fx["logits"].argmax(-1))
    ys = b["trg"][:, 1:].contiguous().view(-1)
    nc = cfg["nconds"]
    ys_cond = b["dconds"].unsqueeze(2).contiguous().view(-1, nc, 1) if nc else None
    loss, rce, _, kld = O.loss_function(fx["beta"], prop, mol, ys_cond, ys, mu, lv, False,
                                        synthetic.PAD_ID)
    assert abs(loss.item() - fx["loss"]) <= 1e-5 * abs(fx["loss"])
    assert abs(rce.item() - fx["rce"]) <= 1e-5 * abs(fx["rce"])
    assert abs(kld.item() - fx["kld"]) <= 1e-5 * abs(fx["kld"])
    loss.backward()
    for name in fx["param_order"]:
        if name in fx["no_grad_params"]:
            assert P[name].grad is None, name
            continue
        g, e = P[name].grad, fx["grads"][name]
        assert torch.allclose(g, e, atol=1e-5 * float(e.abs().max()) + 1e-7, rtol=1e-4), name
    if mtype == "vaetf":   # dead encoder.fc_* (SURVEY 2.3 caveat)
        assert set(fx["no_grad_params"]) == {"encoder.fc_mu.weight", "encoder.fc_mu.bias",
                                             "encoder.fc_log_var.weight", "encoder.fc_log_var.bias"}


@pytest.mark.parametrize("mtype", TYPES)
def test_five_step_history(golden_dir, mtype):
    """G3: oracle train_step reproduces the reference's run_epoch history."""
    g3 = json.load(open(os.path.join(golden_dir, "g3_history.json")))[mtype]
    vs, vt = synthetic.vocab_sizes(mtype)
    cfg = O.make_cfg(mtype, vs, vt, N=2, d_model=64, dff=128, h=4, latent_dim=16, dropout=0.0,
                     nconds=synthetic.n_conds(mtype), use_cond2lat=True)
    P = O.make_leaves(O.init_state(cfg, seed=1))
    opt = O.make_adam(O.trainable(P, cfg))
    ds = synthetic.make_dataset(20, max_len=20, model_type=mtype, seed=11)
    torch.manual_seed(2024)
    for i, batch in enumerate(synthetic.batches(ds, 4)):
        n = batch["src"].size(0)
        loss, rce, kld, lr = O.train_step(P, cfg, opt, batch, 0.04, synthetic.PAD_ID, i + 1)
        assert abs(rce / n - g3["RCE"][i]) <= 2e-5 * abs(g3["RCE"][i]), (i, rce / n, g3["RCE"][i])
        assert abs(kld / n - g3["KLD"][i]) <= 2e-5 * abs(g3["KLD"][i])
        assert abs(loss / n - g3["LOSS"][i]) <= 2e-5 * abs(g3["LOSS"][i])
        assert abs(lr - g3["LR"][i]) <= 1e-12
    assert g3["final_step"] == 5


def test_greedy_decode(golden_dir):
    g5 = torch.load(os.path.join(golden_dir, "g5_decode.pt"), weights_only=True)
    for mtype, fx in g5.items():
        vs, vt = synthetic.vocab_sizes(mtype)
        nc = synthetic.n_conds(mtype)
        cfg = O.make_cfg(mtype, vs, vt, N=2, d_model=64, dff=128, h=4, latent_dim=16,
                         dropout=0.0, nconds=nc, use_cond2lat=True)
        P = O.init_state(cfg, seed=1)
        n, L = fx["z"].shape[0], fx["z"].shape[1]
        src_mask = torch.ones(n, 1, L, dtype=torch.bool)
        ys = O.greedy_decode(P, cfg, fx["z"], src_mask, fx["dconds"], synthetic.SOS_ID,
                             -1, synthetic.PAD_ID, max_strlen=16)
        assert torch.equal(ys, fx["ys"]), mtype                     # bit-exact token ids


def test_schedules():
    assert abs(O.kl_beta(1) - 0.04) < 1e-12                        # 0.04 at epoch 1
    assert abs(O.warmup_lr(1, 512, 8000) - 512 ** -0.5 * 8000 ** -1.5) < 1e-18


def test_cond2dec_path(golden_dir):
    """-use_cond2dec: cond tokens in the decoder stream, block mask, prop_fc head + MSE term."""
    fx = torch.load(os.path.join(golden_dir, "g2_pvaetf_cond2dec.pt"), weights_only=True)
    cfg = O.make_cfg("pvaetf", 28, 30, N=2, d_model=64, dff=128, h=4, latent_dim=16, dropout=0.0,
                     nconds=3, use_cond2dec=True, use_cond2lat=False)
    st = O.init_state(cfg, seed=1)
    assert list(st.keys()) == list(fx["init_sha256"].keys())
    for k, v in st.items():
        assert sha(v) == fx["init_sha256"][k], k
    P = O.make_leaves(st)
    b = fx["batch"]
    src_mask, trg_mask, trg_in = O.batch_masks(cfg, b, synthetic.PAD_ID)
    prop, mol, mu, lv, _ = O.forward(P, cfg, b["src"], trg_in, src_mask, trg_mask, b["econds"], b["dconds"],
                                     eps=fx["eps"], train=True)
    assert torch.allclose(prop, fx["prop"], atol=2e-6) and torch.allclose(mol, fx["logits"], atol=2e-6)
    ys = b["trg"][:, 1:].contiguous().view(-1)
    ys_cond = b["dconds"].unsqueeze(2).contiguous().view(-1, 3, 1)
    loss, rce, rce_prop, kld = O.loss_function(fx["beta"], prop, mol, ys_cond, ys, mu, lv, True, synthetic.PAD_ID)
    for got, key in ((loss, "loss"), (rce, "rce"), (rce_prop, "rce_prop"), (kld, "kld")):
        assert abs(got.item() - fx[key]) <= 1e-5 * abs(fx[key]), key
    loss.backward()
    for name, e in fx["grads"].items():
        assert torch.allclose(P[name].grad, e, atol=1e-5 * float(e.abs().max()) + 1e-7, rtol=1e-4), name


# ------------------------------------------------------------------ G6: reference-written checkpoints
def _load_ref_ckpt(golden_dir, name):
    from gct_plus_amd.Model import load_checkpoint          # weights-only loader (+ numpy-scalar allow-list)
    return load_checkpoint(os.path.join(golden_dir, name))


@pytest.mark.parametrize("mtype", ["pvaetf", "vaetf"])
def test_reference_checkpoint_continues_in_oracle(golden_dir, mtype):
    """G6: a file written by the REFERENCE's save_checkpoint (3 steps of its run_epoch + stock Adam) is loaded into
    the oracle (weights + Adam state); the next two steps and model.encode reproduce the reference's."""
    ck = _load_ref_ckpt(golden_dir, f"g6_ckpt_{mtype}.pt")
    exp = json.load(open(os.path.join(golden_dir, "g6_expect.json")))[mtype]
    assert set(ck) == {"model_state_dict", "opt_state_dict", "model_params"}
    vs, vt = synthetic.vocab_sizes(mtype)
    nc = synthetic.n_conds(mtype)
    assert ck["model_params"]["nconds"] == nc and ck["model_params"]["d_ff"] == 128
    cfg = O.make_cfg(mtype, vs, vt, N=2, d_model=64, dff=128, h=4, latent_dim=16, dropout=0.0, nconds=nc,
                     use_cond2lat=True)
    st = O.init_state(cfg, seed=5)
    assert list(st.keys()) == list(ck["model_state_dict"].keys())      # checkpoint layout == reference's
    P = O.make_leaves(ck["model_state_dict"])
    opt = O.make_adam(O.trainable(P, cfg))
    opt.load_state_dict(ck["opt_state_dict"])
    ds = synthetic.make_dataset(20, max_len=20, model_type=mtype, seed=11)
    loader = list(synthetic.batches(ds, 4))
    # encode path (eval): mu, log_var, z under the pinned eps
    b = loader[0]
    eps = torch.randn(4, 20 + nc, 16, generator=torch.Generator().manual_seed(exp["encode_eps_seed"]))
    with torch.no_grad():
        sm = O.get_src_mask(b["src"], synthetic.PAD_ID, b.get("econds"))
        z, mu, lv = O.encode(P, cfg, b["src"], sm, b.get("econds"), eps=eps, train=False)
    for got, key in ((z, "z"), (mu, "mu"), (lv, "log_var")):
        assert torch.allclose(got, torch.tensor(exp["encode"][key]), atol=2e-6), key
    torch.manual_seed(77)
    for j, batch in enumerate(loader[3:5]):
        n = batch["src"].size(0)
        loss, rce, kld, lr = O.train_step(P, cfg, opt, batch, 0.04, synthetic.PAD_ID, 3 + j + 1)
        for got, key in ((loss, "LOSS"), (rce, "RCE"), (kld, "KLD")):
            assert abs(got / n - exp["continued"][key][j]) <= 2e-5 * abs(exp["continued"][key][j]), (key, j)
        assert abs(lr - exp["continued"]["LR"][j]) <= 1e-12


def test_reference_checkpoint_module_prefix_and_refusal(golden_dir, tmp_path):
    a = _load_ref_ckpt(golden_dir, "g6_ckpt_pvaetf.pt")
    b = _load_ref_ckpt(golden_dir, "g6_ckpt_pvaetf_module.pt")
    assert ["module." + k for k in a["model_state_dict"]] == list(b["model_state_dict"])
    assert all(torch.equal(a["model_state_dict"][k], b["model_state_dict"]["module." + k]) for k in a["model_state_dict"])
    # the reference writes its lr as a numpy scalar: that (and nothing else) is allow-listed
    assert type(a["opt_state_dict"]["param_groups"][0]["lr"]).__module__ == "numpy"
    import pickle
    import subprocess  # noqa: F401  (the refused global below)
    bad = str(tmp_path / "bad.pt")
    torch.save({"model_state_dict": {}, "x": subprocess.Popen.__init__}, bad)
    with pytest.raises(pickle.UnpicklingError):
        _load_ref_ckpt(str(tmp_path), "bad.pt")


def test_full_size_init_hashes(golden_dir):
    """G4 init: every tensor of the full-size vaetf (config 1, seed 1) built by the oracle AND by the product's
    module classes hashes to the reference's sha256 (tests/golden/g4_init_sha.json)."""
    from gct_plus_amd.Model import model_dict
    ref = json.load(open(os.path.join(golden_dir, "g4_init_sha.json")))
    cfg = O.make_cfg("vaetf", 28, 30, dropout=0.0, nconds=0, use_cond2lat=True)
    st = O.init_state(cfg, seed=1)
    assert list(st.keys()) == list(ref.keys())
    for k, v in st.items():
        assert sha(v) == ref[k], f"oracle {k}"
    torch.manual_seed(1)
    m = model_dict["vaetf"](28, 30, N=6, d_model=512, dff=2048, h=8, latent_dim=128, dropout=0.0, nconds=0,
                            use_cond2dec=False, use_cond2lat=True)
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    for k, v in sd.items():
        assert sha(v) == ref[k], f"product {k}"
