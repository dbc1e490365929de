"""Tokeniser / vocabulary / collate (CPU): the native scanner against Python's `re` with the
reference's pattern (Utils/field.py:16 -- written there as a NON-raw string, so its `\\\\\\\\` is one
escaped backslash), on the SMILES the reference keeps in Inference/test_encoder.py:388-445 and
on random strings; batch layout against Model/collate_fn.py semantics."""
import random
import re

import pandas as pd
import torch

from gct_plus_amd import data

PATTERN = re.compile(r"(\[[^\]]+]|Br?|Cl?|N|O|S|P|F|I|b|c|n|o|s|p|\(|\)|\.|=|#|-|\+|\\|\/|:|~|@|\?|>|\*|\$|\%[0-9]{2}|[0-9])")
# molecules listed in the reference's Inference/test_encoder.py:388-445 (data, not code)
SMILES = [
    "CCc1cccc(OCC(=O)Nc2ccc(C(=O)N(C)C)cc2)c1", "CCc1cccc(OCC(=O)Nc2ccc(C(=O)N(C)C)c(F)c2)c1",
    "Cc1cccc(OCC(=O)NC(C)(C)Cc2ccc3c(c2)OCCO3)c1", "CCN1CCCC2(CCN(C(=O)c3cc(C)nc4ccccc34)C2)C1=O",
    "CC1CCCN(C(=O)CN(C)C(=O)NC(C)(C)c2ccccc2F)C1", "CC1CCN(C(=O)CN2CCN(C(=O)Nc3cccc(Cl)c3)CC2)CC1",
    "CC1CCN(C(=O)CNC(c2ccc(F)cc2)c2cnn(C)c2)CC1", "CC1CN(C(=O)CNC(c2cccc(F)c2)C(C)(C)C)CC(C)O1",
    "CC1CCN(C(=O)CN(C)C(=O)c2ccc3c(c2)CCC3)C(C)C1", "CC1CCN(C(=O)CN(C)c2nc3ccccc3s2)CC1",
    "C1(C(=O)N2CC3c4ccccc4CCCN3C(=O)C2)CCCCC1", "C(Oc1c(OC)cccc1)(c1ccccc1OC(C)=O)=O",
    "c12nc(N)nc(N)c1c(C)c(Cc1cc(OC)ccc1OC)cn2", "n1c2c(c(N)nc1N)c(C)c(Cc1cc(OC)ccc1OC)cn2",
]


def test_tokenizer_matches_reference_regex():
    for s in SMILES + ["C[C@@H](Br)Cl.[Na+]%12xyz[", "B%1r[]Cl\\/", "", "[", "[]", "%9", "%123", "Brr", "Cll", "<sep>"]:
        assert data.tokenize(s) == PATTERN.findall(s), s
    rnd = random.Random(0)
    alphabet = "BrClNOSPFIbcnosp()[].=#-+\\/:~@?>*$%0123456789 xyzH<>e"
    for _ in range(3000):
        s = "".join(rnd.choice(alphabet) for _ in range(rnd.randint(0, 40)))
        assert data.tokenize(s) == PATTERN.findall(s), repr(s)


def test_tokenizer_with_sep():
    assert data.tokenize("c1ccccc1<sep>CCO", True) == ["c", "1", "c", "c", "c", "c", "c", "1", "<sep>", "C", "C", "O"]
    assert data.tokenize("CC<sep>C<sep>O", True) == []          # field.py:25-33: more than one <sep>
    assert data.tokenize("CCO", True) == ["C", "C", "O"]


def test_vocab_order_and_collate_layout():
    SRC = data.Vocab.build(SMILES, target=False, add_sep=False)
    TRG = data.Vocab.build(SMILES, target=True, add_sep=False)
    assert SRC.itos[:2] == ["<unk>", "<pad>"] and TRG.itos[:4] == ["<unk>", "<pad>", "<sos>", "<eos>"]
    assert SRC.itos[2:] == TRG.itos[4:]                             # same token order, ids shifted by 2
    assert SRC.itos[2] == "C" or SRC.itos[2] == "c"                 # most frequent symbol first
    src, ls = SRC.encode_batch(SMILES[:4] + ["CZ"], False, sos_eos=False)
    trg, lt = TRG.encode_batch(SMILES[:4] + ["CZ"], False, sos_eos=True)
    assert trg.shape[1] == src.shape[1] + 2                        # SURVEY: trg [B,S+2]
    for i, s in enumerate(SMILES[:4]):
        toks = PATTERN.findall(s)
        assert src[i, :len(toks)].tolist() == [SRC.stoi[t] for t in toks]
        assert (src[i, len(toks):] == SRC.stoi["<pad>"]).all()
        assert trg[i, 0] == 2 and trg[i, len(toks) + 1] == 3
        assert trg[i, 1:len(toks) + 1].tolist() == [TRG.stoi[t] for t in toks]
        assert (trg[i, 1:len(toks) + 1] == src[i, :len(toks)] + 2).all()
    assert src[4, 1] == SRC.stoi["<unk>"] or src[4, :2].tolist() == [SRC.stoi["C"], SRC.stoi["<pad>"]]


def test_loader_scaffold_batches_and_sharding(tmp_path):
    rows = []
    for i, s in enumerate(SMILES):
        rows.append({"src": s, "src_scaffold": "c1ccccc1", "src_logP": 0.1 * i, "trg_logP": 0.1 * i,
                     "src_tPSA": 1.0, "trg_tPSA": 1.0, "src_QED": 0.5, "trg_QED": 0.5})
    f = pd.DataFrame(rows)
    strs = [r["src_scaffold"] + "<sep>" + r["src"] for r in rows]
    SRC = data.Vocab.build(strs, False, True)
    TRG = data.Vocab.build(strs, True, True)
    assert SRC.itos[2] == "<sep>" and TRG.itos[4] == "<sep>"
    seen = []
    for rank in range(2):
        ld = data.SmilesLoader(f, SRC, TRG, "pscavaetf", ["logP", "tPSA", "QED"], 4, rank, 2, False, 0, "cpu")
        assert len(ld) == 2
        for b in ld:
            assert b["src"].shape[0] <= 4 and b["trg"].shape[1] == b["src"].shape[1] + 2
            assert b["econds"].shape == (b["src"].shape[0], 3)
            assert (b["src"] == SRC.stoi["<sep>"]).sum(1).eq(1).all()
            seen.append(b["econds"][:, 0])
    got = sorted(round(float(x) * 10) for x in torch.cat(seen))
    assert got == list(range(len(SMILES)))                         # the two shards cover the data once


def test_bench_synthetic_smiles_fill_the_benchmarked_vocabularies(tmp_path):
    """bench.py's trainer_loop leg runs run_epoch behind the tokenizer + collate loader on synthetic SMILES strings: the
    vocabularies built from them must have exactly the 28 / 30 entries of the synthetic token batches (so that the
    benchmarked model can consume them and the leg is not skipped), and a collated batch must have the trainer's layout."""
    import importlib.util
    import os
    import torch
    from gct_plus_amd import data, synthetic
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    frame = mod.synthetic_smiles_frame(600, seed=3)
    SRC, TRG, add_sep = data.get_fields("vaetf", str(tmp_path / "utils"), frame["src"].tolist())
    assert (len(SRC), len(TRG)) == synthetic.vocab_sizes("vaetf") and not add_sep
    ld = data.SmilesLoader(frame, SRC, TRG, "vaetf", [], 64, 0, 1, shuffle=False, seed=0, device=torch.device("cpu"))
    b = next(iter(ld))
    assert b["src"].shape[0] == 64 and b["trg"].shape == (64, b["src"].shape[1] + 2)
    assert int(b["src"].max()) < len(SRC) and int(b["trg"].max()) < len(TRG)
    assert (b["trg"][:, 0] == TRG.stoi["<sos>"]).all()
    lens = (b["src"] != SRC.stoi["<pad>"]).sum(1)
    assert 15 <= int(lens.min()) and int(lens.max()) <= 78
