"""CPU oracle for the GCT-Plus Transformer-VAE training step.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The shipped path (``gct_plus_amd``) never routes through this file and fails loudly
when the HIP library is missing.

It is a *functional* restatement (plain functions over a ``{name: tensor}`` state
dict, PyTorch fp32 on CPU) of the arithmetic of the reference hot path.  Every
function cites the reference file:line it follows (paths relative to
/root/reference).  The reference is floating point PyTorch, so the restatement is
PyTorch fp32 as well (same ATen CPU kernels => identical rounding).

Parity pinning: ``tests/golden/make_golden.py`` imports the real reference in the
build container and freezes inputs/outputs/gradients/loss curves as fixtures under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this file against them
(bit-exact for init and masks, <=1e-6 abs for forward values).
"""
from __future__ import annotations

import copy
import math
from collections import OrderedDict
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

MODEL_CLASS = {  # Model/build_model.py:8-14
    "vaetf": "vaetf",
    "pvaetf": "cvaetf",
    "scavaetf": "cvaetf",
    "pscavaetf": "cvaetf",
}


def make_cfg(model_type, src_vocab, trg_vocab, N=6, d_model=512, dff=2048, h=8,
             latent_dim=128, dropout=0.1, nconds=0, use_cond2dec=False,
             use_cond2lat=False, variational=True):
    """Hyper-parameter bundle; mirrors Model/build_model.py:42-56 (extract_params)."""
    return dict(model_type=model_type, src_vocab=src_vocab, trg_vocab=trg_vocab, N=N,
                d_model=d_model, dff=dff, h=h, latent_dim=latent_dim, dropout=dropout,
                nconds=nconds, use_cond2dec=use_cond2dec, use_cond2lat=use_cond2lat,
                variational=variational)


# --------------------------------------------------------------------------------------
# constants / masks
# --------------------------------------------------------------------------------------
def positional_table(d_model: int, max_seq_len: int = 200) -> torch.Tensor:
    """Model/modules.py:123-131.  Note the non-Vaswani exponents: sin uses 2*i/d,
    cos uses 2*(i+1)/d with i already the even column index."""
    pe = torch.zeros(max_seq_len, d_model)
    for pos in range(max_seq_len):
        for i in range(0, d_model, 2):
            pe[pos, i] = math.sin(pos / (10000 ** ((2 * i) / d_model)))
            pe[pos, i + 1] = math.cos(pos / (10000 ** ((2 * (i + 1)) / d_model)))
    return pe.unsqueeze(0)


def nopeak_mask(trg_size: int, use_cond2dec: bool, pad_idx: int, cond_dim: int = 0):
    """Model/modules.py:17-30.  Lower-triangular 'may attend' pattern multiplied by
    pad_idx (an int64 0/1 tensor only because pad_idx == 1)."""
    allow = torch.tril(torch.ones(trg_size, trg_size, dtype=torch.bool))
    if use_cond2dec:
        n = cond_dim + trg_size
        full = torch.zeros(n, n, dtype=torch.bool)
        full[:cond_dim, :cond_dim] = True            # cond rows see all conds
        full[:cond_dim, cond_dim] = True             # ... and the first target token
        full[cond_dim:, :cond_dim] = True            # target rows see all conds
        full[cond_dim:, cond_dim:] = allow
        allow = full
    return allow.unsqueeze(0) * pad_idx


def get_src_mask(src, pad_idx, conditions=None):
    """Model/modules.py:38-44."""
    m = (src != pad_idx).unsqueeze(-2)
    if conditions is not None:
        ones = torch.ones(conditions.size(0), 1, conditions.size(1), dtype=torch.bool)
        m = torch.cat([ones, m], dim=2)
    return m


def get_trg_mask(target, pad_id, use_cond2dec, conditions=None):
    """Model/modules.py:47-58 (CPU-safe: the original's .to(target.get_device())
    raises on CPU tensors)."""
    m = (target != pad_id).unsqueeze(-2)
    if use_cond2dec:
        ones = torch.ones(conditions.size(0), 1, conditions.size(1), dtype=torch.bool)
        m = torch.cat([ones, m], dim=2)
    cond_dim = 0 if conditions is None else conditions.size(-1)
    return m & nopeak_mask(target.size(1), use_cond2dec, pad_id, cond_dim)


# --------------------------------------------------------------------------------------
# initial state with init-order parity
# --------------------------------------------------------------------------------------
class _Box(nn.Module):
    """Anonymous container: children registered in call order."""


class _NormP(nn.Module):  # parameter holder for Model/modules.py:80-90
    def __init__(self, d):
        super().__init__()
        self.alpha = nn.Parameter(torch.ones(d))
        self.bias = nn.Parameter(torch.zeros(d))


class _PEBuf(nn.Module):  # buffer holder for Model/modules.py:116-132
    def __init__(self, d):
        super().__init__()
        self.register_buffer("pe", positional_table(d))


def _attn_box(d):  # Model/sublayers.py:54-59: registration order q, v, k, out
    b = _Box()
    b.q_linear = nn.Linear(d, d)
    b.v_linear = nn.Linear(d, d)
    b.k_linear = nn.Linear(d, d)
    b.out = nn.Linear(d, d)
    return b


def _ff_box(d, dff):  # Model/sublayers.py:81-83
    b = _Box()
    b.linear_1 = nn.Linear(d, dff)
    b.linear_2 = nn.Linear(dff, d)
    return b


def _enc_layer_box(d, dff):  # Model/layers.py:9-18
    b = _Box()
    b.norm_1 = _NormP(d)
    b.attn = _attn_box(d)
    b.norm_2 = _NormP(d)
    b.ff = _ff_box(d, dff)
    return b


def _dec_layer_box(d, dff):  # Model/layers.py:42-54
    b = _Box()
    b.norm_1 = _NormP(d)
    b.attn_1 = _attn_box(d)
    b.norm_2 = _NormP(d)
    b.attn_2 = _attn_box(d)
    b.norm_3 = _NormP(d)
    b.ff = _ff_box(d, dff)
    return b


def _clones(layer, n):  # Model/modules.py:73-74
    return nn.ModuleList([copy.deepcopy(layer) for _ in range(n)])


def _embed_box(vocab, d):  # Model/modules.py:101-106
    b = _Box()
    b.embed = nn.Embedding(vocab, d)
    return b


def build_skeleton(cfg) -> nn.Module:
    """Parameter-holder tree whose construction consumes the RNG exactly like the
    reference constructors (Model/vaetf.py:14-31,57-77,117-138; Model/cvaetf.py:14-33,
    72-91,136-160) followed by reset_parameters (vaetf.py:140-143)."""
    d, dff, N, lat, nc = cfg["d_model"], cfg["dff"], cfg["N"], cfg["latent_dim"], cfg["nconds"]
    c2d, c2l = cfg["use_cond2dec"], cfg["use_cond2lat"]
    kind = MODEL_CLASS[cfg["model_type"]]
    root = _Box()
    enc = _Box()
    enc.embed_sentence = _embed_box(cfg["src_vocab"], d)
    if kind == "cvaetf" and nc > 0:
        enc.embed_cond2enc = nn.Linear(nc, d * nc)
    enc.norm = _NormP(d)
    enc.pe = _PEBuf(d)
    enc.layers = _clones(_enc_layer_box(d, dff), N)
    enc.fc_mu = nn.Linear(d, lat)
    enc.fc_log_var = nn.Linear(d, lat)
    if kind == "vaetf" and nc > 0:
        enc.embed_cond2enc = nn.Linear(nc, d * nc)
    root.encoder = enc

    dec = _Box()
    dec.embed = _embed_box(cfg["trg_vocab"], d)
    if kind == "cvaetf":
        if c2d and nc > 0:
            dec.embed_cond2dec = nn.Linear(nc, d * nc)
        if c2l and nc > 0:
            dec.embed_cond2lat = nn.Linear(nc, d * nc)
    dec.pe = _PEBuf(d)
    dec.fc_z = nn.Linear(lat, d)
    dec.layers = _clones(_dec_layer_box(d, dff), N)
    dec.norm = _NormP(d)
    if kind == "vaetf":
        if c2d and nc > 0:
            dec.embed_cond2dec = nn.Linear(nc, d * nc)
        if c2l and nc > 0:
            dec.embed_cond2lat = nn.Linear(nc, d * nc)
    root.decoder = dec

    if kind == "vaetf":
        smp = _Box()
        smp.fc_mu = nn.Linear(d, lat)
        smp.fc_log_var = nn.Linear(d, lat)
        root.sampler = smp
        root.out = nn.Linear(d, cfg["trg_vocab"])
        if c2d and nc > 0:
            root.prop_fc = nn.Linear(cfg["trg_vocab"], 1)
    else:
        if c2d and nc > 0:
            root.prop_fc = nn.Linear(cfg["trg_vocab"], 1)
        root.out = nn.Linear(d, cfg["trg_vocab"])

    for _, p in root.named_parameters():  # vaetf.py:140-143 / cvaetf.py:162-165
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)
    return root


def init_state(cfg, seed: Optional[int] = None) -> "OrderedDict[str, torch.Tensor]":
    """state_dict (parameters + pe buffers, reference key order) of a freshly built
    model; with ``seed`` it is bit-identical to the reference built under
    torch.manual_seed(seed)."""
    if seed is not None:
        torch.manual_seed(seed)
    skel = build_skeleton(cfg)
    return OrderedDict((k, v.detach().clone()) for k, v in skel.state_dict().items())


def param_names(cfg):
    """named_parameters() order (== torch.optim.Adam state index order)."""
    with torch.random.fork_rng():
        skel = build_skeleton(cfg)
    return [n for n, _ in skel.named_parameters()]


# --------------------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------------------
def _lin(P, pre, x):
    return F.linear(x, P[pre + ".weight"], P[pre + ".bias"])


def norm(x, alpha, bias, eps=1e-6):
    """Model/modules.py:92-95: unbiased std, eps added to std (not to variance)."""
    return alpha * (x - x.mean(dim=-1, keepdim=True)) / (x.std(dim=-1, keepdim=True) + eps) + bias


def _norm(P, pre, x):
    return norm(x, P[pre + ".alpha"], P[pre + ".bias"])


def _drop(x, p, train):
    return F.dropout(x, p, train) if (train and p > 0) else x


def pos_encode(P, pre, x, d_model, p, train):
    """Model/modules.py:134-144."""
    x = x * math.sqrt(d_model)
    x = x + P[pre + ".pe"][:, : x.size(1)]
    return _drop(x, p, train)


def attention(q, k, v, d_k, mask, p, train):
    """Model/sublayers.py:29-41 (dropout is applied to the probabilities)."""
    scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(d_k)
    if mask is not None:
        scores = scores.masked_fill(mask.unsqueeze(1) == 0, -1e9)
    probs = F.softmax(scores, dim=-1)
    out = torch.matmul(_drop(probs, p, train), v)
    return out, probs


def mha(P, pre, q, k, v, mask, h, p, train, want_probs=False):
    """Model/sublayers.py:61-74 (evaluation order k, q, v)."""
    bs, d = q.size(0), q.size(-1)
    dk = d // h
    kk = _lin(P, pre + ".k_linear", k).view(bs, -1, h, dk).transpose(1, 2)
    qq = _lin(P, pre + ".q_linear", q).view(bs, -1, h, dk).transpose(1, 2)
    vv = _lin(P, pre + ".v_linear", v).view(bs, -1, h, dk).transpose(1, 2)
    o, probs = attention(qq, kk, vv, dk, mask, p, train)
    o = o.transpose(1, 2).contiguous().view(bs, -1, d)
    o = _lin(P, pre + ".out", o)
    return (o, probs) if want_probs else o


def feed_forward(P, pre, x, p, train):
    """Model/sublayers.py:85-89 (exact erf GELU)."""
    x = F.gelu(_lin(P, pre + ".linear_1", x))
    x = _drop(x, p, train)
    return _lin(P, pre + ".linear_2", x)


def encoder_layer(P, pre, x, mask, h, p, train):
    """Model/layers.py:20-38: the residual branches start from the NORMALISED x."""
    x = _norm(P, pre + ".norm_1", x)
    x = x + _drop(mha(P, pre + ".attn", x, x, x, mask, h, p, train), p, train)
    x = _norm(P, pre + ".norm_2", x)
    x = x + _drop(feed_forward(P, pre + ".ff", x, p, train), p, train)
    return x


def decoder_layer(P, pre, x, e, src_mask, trg_mask, h, p, train):
    """Model/layers.py:56-82 (standard pre-norm)."""
    x2 = _norm(P, pre + ".norm_1", x)
    x = x + _drop(mha(P, pre + ".attn_1", x2, x2, x2, trg_mask, h, p, train), p, train)
    x2 = _norm(P, pre + ".norm_2", x)
    x = x + _drop(mha(P, pre + ".attn_2", x2, e, e, src_mask, h, p, train), p, train)
    x2 = _norm(P, pre + ".norm_3", x)
    x = x + _drop(feed_forward(P, pre + ".ff", x2, p, train), p, train)
    return x


def encoder_trunk(P, cfg, src, src_mask, econds, train):
    """Model/vaetf.py:32-54 / Model/cvaetf.py:35-52 up to and including the final Norm."""
    p = cfg["dropout"]
    x = F.embedding(src, P["encoder.embed_sentence.embed.weight"])
    if cfg["nconds"] > 0:
        c = _lin(P, "encoder.embed_cond2enc", econds).view(econds.size(0), econds.size(1), -1)
        x = torch.cat([c, x], dim=1)
    x = pos_encode(P, "encoder.pe", x, cfg["d_model"], p, train)
    for i in range(cfg["N"]):
        x = encoder_layer(P, f"encoder.layers.{i}", x, src_mask, cfg["h"], p, train)
    return _norm(P, "encoder.norm", x)


def sample(mu, log_var, variational, eps):
    """Model/sublayers.py:14-20 / Model/cvaetf.py:63-69."""
    if not variational:
        return mu
    std = torch.exp(0.5 * log_var)
    if eps is None:
        eps = torch.randn_like(std)
    return eps.mul(std).add(mu)


def encode(P, cfg, src, src_mask, econds=None, eps=None, train=False):
    """Vaetf.encode (vaetf.py:145-148) / Cvaetf.encode (cvaetf.py:171-173)."""
    x = encoder_trunk(P, cfg, src, src_mask, econds, train)
    head = "sampler" if MODEL_CLASS[cfg["model_type"]] == "vaetf" else "encoder"
    mu = _lin(P, head + ".fc_mu", x)
    log_var = _lin(P, head + ".fc_log_var", x)
    return sample(mu, log_var, cfg["variational"], eps), mu, log_var


def decoder_trunk(P, cfg, trg, z, src_mask, trg_mask, dconds, train):
    """Model/vaetf.py:79-114 / Model/cvaetf.py:93-133."""
    p, nc = cfg["dropout"], cfg["nconds"]
    x = F.embedding(trg, P["decoder.embed.embed.weight"])
    e = _lin(P, "decoder.fc_z", z)
    if cfg["use_cond2dec"] and nc > 0:
        c = _lin(P, "decoder.embed_cond2dec", dconds).view(dconds.size(0), dconds.size(1), -1)
        x = torch.cat([c, x], dim=1)
    elif cfg["use_cond2lat"] and nc > 0:
        c = _lin(P, "decoder.embed_cond2lat", dconds).view(dconds.size(0), dconds.size(1), -1)
        e = torch.cat([c, e], dim=1)
    x = pos_encode(P, "decoder.pe", x, cfg["d_model"], p, train)
    if cfg["use_cond2lat"] and nc > 0:
        ones = torch.ones(dconds.size(0), 1, dconds.size(1), dtype=torch.bool)
        src_mask = torch.cat([ones, src_mask], dim=2)
    for i in range(cfg["N"]):
        x = decoder_layer(P, f"decoder.layers.{i}", x, e, src_mask, trg_mask, cfg["h"], p, train)
    return _norm(P, "decoder.norm", x)


def decode(P, cfg, trg, z, src_mask, trg_mask, dconds=None, train=False):
    """Vaetf.decode (vaetf.py:150-152) / Cvaetf.decode (cvaetf.py:175-177)."""
    return _lin(P, "out", decoder_trunk(P, cfg, trg, z, src_mask, trg_mask, dconds, train))


def forward(P, cfg, src, trg, src_mask, trg_mask, econds=None, dconds=None, eps=None,
            train=False):
    """Vaetf.forward (vaetf.py:154-182) / Cvaetf.forward (cvaetf.py:179-193).
    Returns (output_prop, output_mol, mu, log_var, z)."""
    z, mu, log_var = encode(P, cfg, src, src_mask, econds, eps, train)
    output = decode(P, cfg, trg, z, src_mask, trg_mask, dconds, train)
    nc = cfg["nconds"]
    if cfg["use_cond2dec"] and (nc > 0 or MODEL_CLASS[cfg["model_type"]] == "vaetf"):
        prop = _lin(P, "prop_fc", output[:, :nc, :])
        mol = output[:, nc:, :]
    elif MODEL_CLASS[cfg["model_type"]] == "vaetf" or nc > 0:
        prop = torch.zeros(output.size(0), nc, 1)
        mol = output
    else:
        prop, mol = None, output
    return prop, mol, mu, log_var, z


# --------------------------------------------------------------------------------------
# loss / schedules / step  (Train/trainer1.py)
# --------------------------------------------------------------------------------------
def loss_function(beta, preds_prop, preds_mol, ys_cond, ys_mol, mu, log_var,
                  use_cond2dec, pad_id):
    """Train/trainer1.py:19-30: CE(sum, ignore pad) + beta*KLD over ALL latent
    elements (padded source positions included)."""
    rce = F.cross_entropy(preds_mol.contiguous().view(-1, preds_mol.size(-1)), ys_mol,
                          ignore_index=pad_id, reduction="sum")
    kld = -0.5 * torch.sum(1 + log_var - mu.pow(2) - log_var.exp())
    if use_cond2dec:
        rce_prop = F.mse_loss(preds_prop, ys_cond, reduction="sum")
        loss = rce + rce_prop + beta * kld
    else:
        rce_prop = torch.zeros(1)
        loss = rce + beta * kld
    return loss, rce, rce_prop, kld


def kl_beta(epoch, ini=0.02, inc=0.02, beg=1):
    """Train/trainer1.py:14-16."""
    return ini + inc * ((epoch + 1) - beg)


def warmup_lr(step, d_model, warm):
    """Train/trainer1.py:117-123 (value written AFTER step `step`, used by step+1)."""
    return float(d_model) ** -0.5 * min(float(step) ** -0.5, float(step) * float(warm) ** -1.5)


def batch_masks(cfg, batch, pad_id):
    """Model/forward_propagation1.py:4-40."""
    conds_e = batch.get("econds") if cfg["nconds"] > 0 else None
    conds_d = batch.get("dconds") if cfg["nconds"] > 0 else None
    trg_in = batch["trg"][:, :-1]
    return (get_src_mask(batch["src"], pad_id, conds_e),
            get_trg_mask(trg_in, pad_id, cfg["use_cond2dec"], conds_d), trg_in)


def make_leaves(state):
    """Split a state dict into trainable leaf tensors (requires_grad) + buffers."""
    P = OrderedDict()
    for k, v in state.items():
        t = v.detach().clone()
        if not k.endswith(".pe.pe"):
            t.requires_grad_(True)
        P[k] = t
    return P


def trainable(P, cfg):
    names = param_names(cfg)
    return [P[n] for n in names]


def train_step(P, cfg, opt, batch, beta, pad_id, step, warm=8000, eps=None, train=True):
    """One iteration of Train/trainer1.py:80-127 on the functional state.
    Returns (loss, rce, kld, lr_logged)."""
    src_mask, trg_mask, trg_in = batch_masks(cfg, batch, pad_id)
    prop, mol, mu, lv, _ = forward(P, cfg, batch["src"], trg_in, src_mask, trg_mask,
                                   batch.get("econds") if cfg["nconds"] > 0 else None,
                                   batch.get("dconds") if cfg["nconds"] > 0 else None,
                                   eps=eps, train=train)
    ys = batch["trg"][:, 1:].contiguous().view(-1)
    ys_cond = None
    if cfg["nconds"] > 0:
        ys_cond = batch["dconds"].unsqueeze(2).contiguous().view(-1, cfg["nconds"], 1)
    opt.zero_grad(set_to_none=True)
    loss, rce, _, kld = loss_function(beta, prop, mol, ys_cond, ys, mu, lv,
                                      cfg["use_cond2dec"], pad_id)
    loss.backward()
    opt.step()
    lr = warmup_lr(step, cfg["d_model"], warm)
    for g in opt.param_groups:
        g["lr"] = lr
    return loss.item(), rce.item(), kld.item(), lr


def make_adam(params, lr=1e-4, b1=0.9, b2=0.98, eps=1e-9):
    """train1.py:116-119."""
    return torch.optim.Adam(params, lr=lr, betas=(b1, b2), eps=eps)


# --------------------------------------------------------------------------------------
# greedy decode (Inference/sampling_tool.py:140-184), restated loop around decode()
# --------------------------------------------------------------------------------------
@torch.no_grad()
def greedy_decode(P, cfg, z, src_mask, dconds, sos_id, eos_id, pad_id, max_strlen=80, ys0=None, trace=None):
    """ys0: the prefix the scaffold samplers start from (<sos> scaffold <sep>, sampling_tool.py:452-498) instead of the
    single <sos> column; trace (a list): receives the logits of the last position of every step (tie diagnostics)."""
    n = z.size(0)
    ys = torch.full((n, 1), sos_id, dtype=torch.long) if ys0 is None else ys0.clone()
    done = torch.zeros(n, dtype=torch.bool)
    for i in range(max_strlen - 1):
        trg_mask = get_trg_mask(ys, pad_id, cfg["use_cond2dec"], dconds if cfg["nconds"] > 0 else None)
        logits = decode(P, cfg, ys, z, src_mask, trg_mask, dconds)
        if trace is not None:
            trace.append(logits[:, -1].clone())
        nxt = F.softmax(logits, dim=-1)[:, -1].argmax(dim=-1)
        ys = torch.cat([ys, nxt.unsqueeze(1)], dim=1)
        done |= nxt == eos_id
        if bool(done.all()):
            break
    return ys
