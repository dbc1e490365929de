#!/usr/bin/env python3
"""Decoded SMILES/s (BASELINE config 5 shape: pscavaetf-style decode, batch 512, max_strlen 80):
KV-cached decode (eager and graph replay) vs the reference-style full re-run loop."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gct_plus_amd import synthetic  # noqa: E402
from gct_plus_amd.Model import model_dict  # noqa: E402
from gct_plus_amd.decode import KVDecoder, reference_style_decode  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--model-type", default="vaetf")
ap.add_argument("--ref-n", type=int, default=64, help="batch for the (slow) reference-style loop")
ap.add_argument("--ragged", action="store_true", help="latent length 80 with MOSES-like valid lengths N(35,8) (padded "
                "memory, as Inference/*_sampling.py batches it) instead of 40 fully valid positions")
a = ap.parse_args()
mtype = a.model_type
vs, vt = synthetic.vocab_sizes(mtype)
nc = synthetic.n_conds(mtype)
torch.manual_seed(1)
model = model_dict[mtype](vs, vt, N=6, d_model=512, dff=2048, h=8, latent_dim=128, dropout=0.1, nconds=nc,
                          use_cond2lat=True).cuda().eval()
n, Le = a.n, (80 if a.ragged else 40) + nc
z = torch.randn(n, Le, 128, device="cuda")
dconds = torch.randn(n, nc, device="cuda") if nc else None
src_mask = torch.ones(n, 1, Le, dtype=torch.bool, device="cuda")
if a.ragged:
    lens = (torch.randn(n, device="cuda") * 8 + 35).round().clamp(15, 80).long() + nc
    src_mask = (torch.arange(Le, device="cuda")[None, :] < lens[:, None]).unsqueeze(1)
ys0 = torch.full((n, 1), synthetic.SOS_ID, dtype=torch.long, device="cuda")
kd = KVDecoder(model, synthetic.PAD_ID, synthetic.SOS_ID, eos_id=-1)       # never stop early: worst case
for graphs in (False, True):
    kd.start(z, src_mask, dconds, max_total_len=96)
    kd.generate(ys0, 80, use_graphs=graphs, check_every=0)                 # warm-up / capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kd.start(z, src_mask, dconds, max_total_len=96)                        # prefill of the cross K/V is inside
    ys = kd.generate(ys0, 80, use_graphs=graphs, check_every=0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"kv-cached decode graphs={graphs}: n={n} 79 steps {dt*1e3:.1f} ms -> {n/dt:.0f} SMILES/s "
          f"({dt/79*1e3:.2f} ms/step)", flush=True)
m = a.ref_n
torch.cuda.synchronize()
t0 = time.perf_counter()
ref = reference_style_decode(model, z[:m], src_mask[:m], None if dconds is None else dconds[:m], ys0[:m],
                             synthetic.PAD_ID, -1, 80)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"reference-style full re-run (same kernels, no cache): n={m} {dt*1e3:.1f} ms -> {m/dt:.0f} SMILES/s")
print("token ids equal:", bool(torch.equal(ref, ys[:m])))
