"""Tokeniser, vocabulary, dataset and collate for real SMILES data.

Replaces the reference's torchtext-0.6 / dill-pickled `Field` pipeline (Utils/field.py:8-125,
Utils/dataset.py:251-329, Model/collate_fn.py:5-137) -- which cannot even be imported on a
current stack -- with the native scanner in libgctplus_hip.so and plain tensors, producing
exactly the batch layout the model consumes: src [B,S], trg [B,S+2], econds/dconds [B,n_c].

Vocabulary order follows torchtext `Field.build_vocab`: specials first (<unk>, <pad>[, <sos>,
<eos>][, <sep>]), then tokens by descending frequency, ties alphabetical.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from collections import Counter
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .synthetic import shard_indices


def tokenize(smiles: str, add_sep: bool = False) -> List[str]:
    """Utils/field.py moltokenize.tokenizer via the native scanner."""
    lib = _lib.load()
    b = smiles.encode("utf-8")
    cap = len(b) + 1
    st = (C.c_int32 * cap)()
    ln = (C.c_int32 * cap)()
    n = lib.gct_smiles_tokenize(b, int(add_sep), st, ln, cap)
    _lib.check(0 if n >= 0 else n, "gct_smiles_tokenize")
    return [b[st[i]:st[i] + ln[i]].decode("utf-8") for i in range(n)]


class Vocab:
    def __init__(self, itos: Sequence[str]):
        self.itos = list(itos)
        self.stoi = {t: i for i, t in enumerate(self.itos)}
        self._c = (C.c_char_p * len(self.itos))(*[t.encode("utf-8") for t in self.itos])

    def __len__(self):
        return len(self.itos)

    @classmethod
    def build(cls, smiles: Sequence[str], target: bool, add_sep: bool) -> "Vocab":
        cnt = Counter()
        for s in smiles:
            cnt.update(tokenize(s, add_sep))
        specials = ["<unk>", "<pad>"] + (["<sos>", "<eos>"] if target else []) + (["<sep>"] if add_sep else [])
        for sp in specials:
            cnt.pop(sp, None)
        words = sorted(cnt.items(), key=lambda kv: kv[0])
        words.sort(key=lambda kv: kv[1], reverse=True)          # torchtext: freq desc, ties alphabetical
        return cls(specials + [w for w, _ in words])

    def save(self, path):
        json.dump(self.itos, open(path, "w"))

    @classmethod
    def load(cls, path):
        return cls(json.load(open(path)))

    def encode_batch(self, smiles: Sequence[str], add_sep: bool, sos_eos: bool, width: int = 0):
        """-> (int64 tensor [n, longest], lengths) ; rows = [<sos>] ids [<eos>] <pad>..."""
        lib = _lib.load()
        n = len(smiles)
        if width <= 0:
            width = max((len(s) for s in smiles), default=0) + 2
        arr = (C.c_char_p * n)(*[s.encode("utf-8") for s in smiles])
        out = np.empty((n, width), dtype=np.int64)
        lens = np.empty(n, dtype=np.int32)
        sos = self.stoi["<sos>"] if sos_eos else -1
        eos = self.stoi["<eos>"] if sos_eos else -1
        mx = lib.gct_smiles_encode_batch(arr, n, int(add_sep), self._c, len(self.itos), self.stoi["<unk>"],
                                         self.stoi["<pad>"], sos, eos, out.ctypes.data_as(C.c_void_p), width,
                                         lens.ctypes.data_as(C.c_void_p))
        _lib.check(0 if mx >= 0 else mx, "gct_smiles_encode_batch")
        return torch.from_numpy(out[:, :max(mx, 1)].copy()), torch.from_numpy(lens)


def get_fields(model_type: str, util_folder: str, train_smiles: Optional[Sequence[str]] = None):
    """(SRC, TRG) vocabularies: loaded from {util_folder}/SRC[_sep].json / TRG[_sep].json, or built
    from the training SMILES and saved there (the reference's preprocess.py:106-131 does the same
    with torchtext pickles)."""
    add_sep = model_type in ("scavaetf", "pscavaetf")
    sfx = "_sep" if add_sep else ""
    ps, pt = (os.path.join(util_folder, f"{n}{sfx}.json") for n in ("SRC", "TRG"))
    if os.path.exists(ps) and os.path.exists(pt):
        return Vocab.load(ps), Vocab.load(pt), add_sep
    if train_smiles is None:
        raise FileNotFoundError(f"{ps} / {pt} not found and no training data to build them from")
    SRC = Vocab.build(train_smiles, target=False, add_sep=add_sep)
    TRG = Vocab.build(train_smiles, target=True, add_sep=add_sep)
    os.makedirs(util_folder, exist_ok=True)
    SRC.save(ps)
    TRG.save(pt)
    return SRC, TRG, add_sep


class SmilesLoader:
    """DataLoader + DistributedSampler + collate_fn of the reference in one object: per-rank index
    shard (DistributedSampler semantics), per-batch tokenisation (like SmilesDataset.__getitem__,
    Utils/dataset.py:269-286) and padding to the batch's longest row (Field.process)."""

    def __init__(self, frame, SRC: Vocab, TRG: Vocab, model_type, property_list, batch_size, rank, world,
                 shuffle, seed, device, use_scaffold=False):
        self.f, self.SRC, self.TRG = frame, SRC, TRG
        self.props = list(property_list)
        self.add_sep = model_type in ("scavaetf", "pscavaetf")
        self.use_scaffold = use_scaffold or self.add_sep
        self.bs, self.rank, self.world, self.shuffle, self.seed = batch_size, rank, world, shuffle, seed or 0
        self.device, self.epoch = device, 0
        self.src_col = frame["src"].astype(str).tolist()
        self.sca_col = frame["src_scaffold"].astype(str).tolist() if self.use_scaffold else None
        if self.props:
            self.econds = torch.tensor(frame[[f"src_{p}" for p in self.props]].values, dtype=torch.float32)
            self.dconds = torch.tensor(frame[[f"trg_{p}" for p in self.props]].values, dtype=torch.float32)

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return -(-(-(-len(self.src_col) // self.world)) // self.bs)

    def collate(self, idx: Sequence[int]) -> Dict[str, torch.Tensor]:
        if self.use_scaffold:                      # collate_fn.py:104-124: scaffold + <sep> + smiles
            strs = [self.sca_col[i] + "<sep>" + self.src_col[i] for i in idx]
        else:
            strs = [self.src_col[i] for i in idx]
        src, _ = self.SRC.encode_batch(strs, self.add_sep, sos_eos=False)
        trg, _ = self.TRG.encode_batch(strs, self.add_sep, sos_eos=True)
        out = {"src": src.to(self.device), "trg": trg.to(self.device)}
        if self.props:
            ii = torch.as_tensor(idx)
            out["econds"] = self.econds[ii].to(self.device)
            out["dconds"] = self.dconds[ii].to(self.device)
        return out

    def __iter__(self):
        idx = shard_indices(len(self.src_col), self.world, self.rank, self.epoch, self.seed, self.shuffle)
        for s in range(0, len(idx), self.bs):
            yield self.collate(idx[s:s + self.bs])
