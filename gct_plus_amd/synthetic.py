"""Synthetic MOSES-shaped SMILES token batches.

The real pipeline (Utils/dataset.py:251-329, Model/collate_fn.py:5-137 in the
reference) needs torchtext/rdkit and the MOSES download; none are available, so the
benchmark and the parity tests feed token batches with the same *layout*:

  src  [B, S]    int64, <pad>=1 right-padded          (collate_fn.py:8)
  trg  [B, S+2]  int64, <sos>=2 ... <eos>=3, <pad>=1  (collate_fn.py:12)
  econds/dconds [B, n_c] float32                      (collate_fn.py:31-34)

Vocabulary ids follow torchtext-0.6 Field.build_vocab specials order
(<unk>,<pad>,<sos>,<eos>[,<sep>], then symbols): V_src/V_trg = 28/30, or 29/31 for
the scaffold ("_sep") fields (field.py:102-114, preprocess.py:125,128).
"""
from __future__ import annotations

import numpy as np
import torch

PAD_ID, SOS_ID, EOS_ID = 1, 2, 3
SEP_ID = 4                # <sep> of the TRG_sep field (specials order unk, pad, sos, eos, sep); 2 in SRC_sep
N_SYMBOLS = 26


def vocab_sizes(model_type: str):
    sep = model_type in ("scavaetf", "pscavaetf")
    return (28 + int(sep), 30 + int(sep))


def n_conds(model_type: str) -> int:
    return 3 if model_type in ("pvaetf", "pscavaetf") else 0


def make_dataset(n_samples=1000, max_len=80, model_type="vaetf", seed=0, fixed_len=False):
    """Returns dict(src [n,S], trg [n,S+2], econds, dconds).  Sample 0 always has
    the full length so every batch that contains it pads to S = max_len; with
    fixed_len every sample is max_len long (worst case, pure-throughput runs)."""
    rng = np.random.default_rng(seed)
    sep = model_type in ("scavaetf", "pscavaetf")
    first_sym = 3 if sep else 2                      # SRC ids of the 26 symbols
    w = 1.0 / np.arange(1, N_SYMBOLS + 1) ** 1.1     # Zipf-like symbol frequencies
    w /= w.sum()
    src = np.full((n_samples, max_len), PAD_ID, dtype=np.int64)
    trg = np.full((n_samples, max_len + 2), PAD_ID, dtype=np.int64)
    for i in range(n_samples):
        if fixed_len or i == 0:
            ln = max_len
        else:
            ln = int(np.clip(np.rint(rng.normal(35, 8)), 15, max_len))
        toks = rng.choice(N_SYMBOLS, size=ln, p=w) + first_sym
        if sep and ln >= 5:                          # scaffold <sep> smiles
            cut = int(rng.integers(1, max(2, ln // 3)))
            toks[cut] = 2                            # <sep> id in SRC_sep
        src[i, :ln] = toks
        trg[i, 0] = SOS_ID
        trg[i, 1:ln + 1] = toks + 2                  # TRG ids are SRC ids shifted by 2
        trg[i, ln + 1] = EOS_ID
    out = {"src": torch.from_numpy(src), "trg": torch.from_numpy(trg)}
    nc = n_conds(model_type)
    if nc:
        c = torch.from_numpy(rng.normal(0, 1, size=(n_samples, nc)).astype(np.float32))
        out["econds"] = c
        out["dconds"] = c.clone()
    return out


def batches(ds, batch_size, drop_last=False):
    """Sequential mini-batches of a dataset dict, each padded only to the longest
    sample it contains is NOT done: like Field.process with fix_length unset the
    reference pads per batch, but SURVEY 8(d) forces S = max_len via sample 0 /
    fixed_len, so batches keep the dataset's width."""
    n = ds["src"].size(0)
    for s in range(0, n, batch_size):
        e = min(s + batch_size, n)
        if drop_last and e - s < batch_size:
            break
        yield {k: v[s:e] for k, v in ds.items()}


def shard_indices(n, world_size, rank, epoch=0, seed=0, shuffle=True):
    """DistributedSampler(dataset, W, rank, shuffle, drop_last=False) index list
    (Utils/dataset.py:304-307, Train/trainer1.py:165-166): seed+epoch permutation,
    wrap-around padding to a multiple of W, strided split."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    total = -(-n // world_size) * world_size
    pad = total - len(idx)
    if pad > 0:
        idx += (idx * (-(-pad // len(idx))))[:pad]
    return idx[rank:total:world_size]
