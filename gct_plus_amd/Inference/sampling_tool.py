"""Sampling front end with the reference's call surface (Inference/sampling_tool.py:19-647,
Model/build_model.py:90-116 `get_sampler`): per-model-type classes exposing
`sample_smiles(...) -> (smiles, toklen, toklen_gen)`, `encode_smiles(...)`, `id_to_smi`,
`sample_toklen`, `sample_z` -- running on the KV-cached decoder (gct_plus_amd.decode) instead
of re-running the decoder per generated token.

Deliberate deviations, none in the arithmetic:
  * vocabularies are gct_plus_amd.data.Vocab objects (no torchtext Field);
  * `id_to_smi` appends each token once (the reference appends it twice, sampling_tool.py:55-63);
  * token lengths are drawn from the same histogram-with-jitter distribution as
    Inference/toklen_sampling.py:19-36, with numpy's Generator instead of the global RNG.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch

from ..Model.modules import get_src_mask
from ..data import Vocab, tokenize
from ..decode import KVDecoder


def sample_token_lengths(data: Sequence[int], size: int, rng: np.random.Generator) -> np.ndarray:
    """Histogram of the training token lengths (one bin per integer) sampled by inverse CDF,
    plus half-bin Gaussian jitter, rounded (toklen_sampling.py:9-36, sampling_tool.py:75-81)."""
    data = np.asarray(data, dtype=float)
    nbins = max(1, int(data.max() - data.min()))
    count, edges = np.histogram(data, bins=nbins)
    pdf = count / count.sum()
    dx = edges[1] - edges[0]
    centres = edges[:-1] + 0.5 * dx
    cdf = np.concatenate([[0.0], np.cumsum(pdf)])
    u = rng.uniform(0, 1, size)
    idx = np.clip(np.argmax(cdf[None, :] >= u[:, None], axis=1) - 1, 0, nbins - 1)
    return np.rint(centres[idx] + dx * rng.normal(size=size) / 2).astype(int)


class Sampling:
    def __init__(self, model, SRC: Vocab, TRG: Vocab, latent_dim: int, max_strlen: int = 80,
                 cond_dim: int = 0, decode_algo: str = "greedy", toklen_data: Optional[Sequence[int]] = None,
                 scaler=None, device="cuda", seed: int = 0, use_graphs: bool = False):
        self.model, self.SRC, self.TRG = model.eval(), SRC, TRG
        self.pad_id, self.sos_id, self.eos_id = SRC.stoi["<pad>"], TRG.stoi["<sos>"], TRG.stoi["<eos>"]
        self.sep_id = TRG.stoi.get("<sep>")
        self.add_sep = self.sep_id is not None
        self.latent_dim, self.max_strlen, self.cond_dim = latent_dim, max_strlen, cond_dim
        self.decode_algo, self.toklen_data, self.scaler = decode_algo, toklen_data, scaler
        self.device, self.use_graphs = device, use_graphs
        self.rng = np.random.default_rng(seed)
        self.gen = torch.Generator().manual_seed(seed)
        self.seed = seed
        self.kv = KVDecoder(model, self.pad_id, self.sos_id, self.eos_id)

    # ---- helpers with the reference's names ------------------------------------------------
    def init_y(self, n, add_sos=True, sca_ids=None, add_sep=False):
        ids = ([self.sos_id] if add_sos else []) + list(sca_ids or []) + ([self.sep_id] if add_sep else [])
        return torch.tensor([ids] * n, dtype=torch.long)

    def id_to_smi(self, ids) -> str:
        out = []
        for i in ids:
            i = int(i)
            if i == self.eos_id:
                break
            if i != self.sos_id:
                out.append(self.TRG.itos[i])
        return "".join(out)

    def smi_to_id(self, smi, add_sos=False, add_sep=False, add_eos=False) -> List[int]:
        ids = ([self.sos_id] if add_sos else []) + ([self.sep_id] if add_sep else [])
        ids += [self.TRG.stoi.get(t, self.TRG.stoi["<unk>"]) for t in tokenize(smi, self.add_sep)]
        return ids + ([self.eos_id] if add_eos else [])

    def sample_toklen(self, n):
        if self.toklen_data is None:
            raise ValueError("toklen_data (training-set token lengths) is required to sample lengths")
        return sample_token_lengths(self.toklen_data, n, self.rng) + self.cond_dim

    def sample_z(self, toklen, n):
        return torch.randn(n, toklen, self.latent_dim, generator=self.gen)

    def transform(self, prop):
        if self.scaler is not None:
            prop = self.scaler.transform(np.asarray(prop))
        return torch.as_tensor(np.asarray(prop), dtype=torch.float32)

    def tokenize_smiles(self, smiles_list):
        return self.SRC.encode_batch(list(smiles_list), self.add_sep, sos_eos=False)[0]

    # ---- decode: KV-cached equivalent of Sampling.decode (sampling_tool.py:140-184) ----------
    @torch.no_grad()
    def decode(self, zs, ys, src_mask, dconds=None):
        self.seed += 1
        zs, ys, src_mask = zs.to(self.device), ys.to(self.device), src_mask.to(self.device)
        dconds = None if dconds is None else dconds.to(self.device)
        total = ys.size(1) + self.max_strlen
        # the positional table has 200 rows, of which use_cond2dec spends n_c on the condition tokens
        self.kv.start(zs, src_mask, dconds, max_total_len=min(200 - self.kv.off, total))
        return self.kv.generate(ys, self.max_strlen, algo=self.decode_algo, seed=self.seed,
                                use_graphs=self.use_graphs)

    def _latent_setup(self, n, zs, toklen, extra=0):
        if zs is not None:
            assert n == zs.size(0)
            if toklen is None:
                toklen = [zs.size(1) - extra] * n
        elif toklen is None:
            toklen = list(self.sample_toklen(n))
        lat = extra + max(toklen)
        if zs is None:
            zs = self.sample_z(lat, n)
        stop = torch.as_tensor(toklen, dtype=torch.long).view(n, 1, 1) + extra
        src_mask = torch.arange(lat).expand(n, 1, lat) < stop
        return zs, toklen, src_mask

    def _finish(self, outs, toklen, skip=0):
        outs = outs.cpu().numpy()
        smiles = [self.id_to_smi(ids[skip:]) for ids in outs]
        return smiles, toklen, [len(tokenize(s, self.add_sep)) for s in smiles]


class VaetfSampling(Sampling):
    def encode_smiles(self, smiles_list):
        src = self.tokenize_smiles(smiles_list).to(self.device)
        return self.model.encode(src=src, src_mask=get_src_mask(src, self.pad_id))

    def sample_smiles(self, n, zs=None, toklen=None):
        zs, toklen, src_mask = self._latent_setup(n, zs, toklen)
        outs = self.decode(zs, self.init_y(n), src_mask)
        return self._finish(outs, toklen)


class CvaetfSampling(Sampling):
    def encode_smiles(self, smiles_list, econds, transform=True):
        src = self.tokenize_smiles(smiles_list).to(self.device)
        econds = (self.transform(econds) if transform else torch.as_tensor(econds, dtype=torch.float32)).to(self.device)
        return self.model.encode(src=src, src_mask=get_src_mask(src, self.pad_id, econds), econds=econds)

    def sample_smiles(self, dconds, zs=None, toklen=None, transform=True):
        n = len(dconds)
        dconds = self.transform(dconds) if transform else torch.as_tensor(dconds, dtype=torch.float32)
        if zs is None and toklen is not None:
            toklen = [t + self.cond_dim for t in toklen]
        zs, toklen, src_mask = self._latent_setup(n, zs, toklen)
        outs = self.decode(zs, self.init_y(n), src_mask, dconds)
        return self._finish(outs, toklen)


class ScaVaeSampling(Sampling):
    def encode_smiles(self, smiles_list, scaffold_list):
        src = self.tokenize_smiles([b + "<sep>" + a for a, b in zip(smiles_list, scaffold_list)]).to(self.device)
        return self.model.encode(src=src, src_mask=get_src_mask(src, self.pad_id))

    def sample_smiles(self, n, scaffold, zs=None, toklen=None):
        sca_ids = self.smi_to_id(scaffold)
        zs, toklen, src_mask = self._latent_setup(n, zs, toklen, extra=len(sca_ids) + 1)
        outs = self.decode(zs, self.init_y(n, True, sca_ids, True), src_mask)
        return self._finish(outs, toklen, skip=1 + len(sca_ids) + 1)


class PscavaetfSampling(Sampling):
    def encode_smiles(self, smiles_list, scaffold_list, econds, transform=True):
        src = self.tokenize_smiles([b + "<sep>" + a for a, b in zip(smiles_list, scaffold_list)]).to(self.device)
        econds = (self.transform(econds) if transform else torch.as_tensor(econds, dtype=torch.float32)).to(self.device)
        return self.model.encode(src=src, src_mask=get_src_mask(src, self.pad_id, econds), econds=econds)

    def sample_smiles(self, dconds, scaffold, zs=None, toklen=None, transform=True):
        """prefix = <sos> scaffold <sep> (sampling_tool.py:452-498)."""
        n = len(dconds)
        dconds = self.transform(dconds) if transform else torch.as_tensor(dconds, dtype=torch.float32)
        sca_ids = self.smi_to_id(scaffold)
        zs, toklen, src_mask = self._latent_setup(n, zs, toklen, extra=len(sca_ids) + 1)
        outs = self.decode(zs, self.init_y(n, True, sca_ids, True), src_mask, dconds)
        return self._finish(outs, toklen, skip=1 + len(sca_ids) + 1)


sampling_tool_dict = {
    "vaetf": VaetfSampling,
    "pvaetf": CvaetfSampling,
    "scavaetf": ScaVaeSampling,
    "pscavaetf": PscavaetfSampling,
}


def get_sampler(model_type, model, SRC, TRG, **kwargs):
    """Model/build_model.py:90-116 counterpart."""
    return sampling_tool_dict[model_type](model, SRC, TRG, **kwargs)
