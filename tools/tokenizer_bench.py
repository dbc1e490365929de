"""Host-side throughput of the tokenizer + batch encoder: the native scanner behind
gct_plus_amd.data.Vocab.encode_batch against a Python `re` tokenizer + list padding (what the reference's
Field.process does per batch).  CPU only, single thread.  python tools/tokenizer_bench.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from gct_plus_amd import data

sys.path.insert(0, "tests")
from test_data_pipeline import PATTERN as rx          # the tokenizer pattern the parity test checks against
rng = np.random.default_rng(0)
frag = ["C", "c1ccccc1", "N", "O", "(", ")", "=", "Cl", "Br", "[nH]", "[C@@H]", "F", "S", "#", "1", "2", "c", "n", "[O-]", "[N+]"]
smiles = ["".join(rng.choice(frag, size=int(rng.integers(10, 30)))) for _ in range(20000)]
SRC = data.Vocab.build(smiles[:2000], target=False, add_sep=False)
TRG = data.Vocab.build(smiles[:2000], target=True, add_sep=False)
B = 512
batches = [smiles[i:i + B] for i in range(0, len(smiles), B)]

t0 = time.perf_counter()
ntok = 0
for b in batches:
    src, ln = SRC.encode_batch(b, False, sos_eos=False)
    trg, _ = TRG.encode_batch(b, False, sos_eos=True)
    ntok += int(ln.sum())
t1 = time.perf_counter()

def py_batch(b, stoi, sos_eos):
    rows = [[stoi.get(t, 0) for t in rx.findall(s)] for s in b]
    if sos_eos:
        rows = [[stoi["<sos>"]] + r + [stoi["<eos>"]] for r in rows]
    w = max(len(r) for r in rows)
    return np.array([r + [stoi["<pad>"]] * (w - len(r)) for r in rows], dtype=np.int64)

t2 = time.perf_counter()
for b in batches:
    a = py_batch(b, SRC.stoi, False)
    c = py_batch(b, TRG.stoi, True)
t3 = time.perf_counter()
assert (a == src.numpy()).all() and (c == trg.numpy()).all()
n = len(smiles)
print(f"native scanner: {n / (t1 - t0):,.0f} SMILES/s ({ntok / (t1 - t0) / 1e6:.1f} M tokens/s, src + trg rows)   "
      f"python re + padding: {n / (t3 - t2):,.0f} SMILES/s   ratio {(t3 - t2) / (t1 - t0):.1f}x   "
      f"(a 512-sample batch: {1e3 * (t1 - t0) / len(batches):.2f} ms against the 56 ms training step)")
