#!/usr/bin/env python3
"""One attention shape, a few launches -- for rocprofv3 --pmc passes (tools/kernel_bench.py does the timing)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gct_plus_amd import ops  # noqa: E402

B, H, L, dk = 512, 8, int(os.environ.get("ATTN_L", "80")), 64
causal = os.environ.get("ATTN_CAUSAL", "0") == "1"
fixed = os.environ.get("ATTN_FIXED", "0") == "1"
d = H * dk
dev = "cuda"
torch.manual_seed(0)
qkv = torch.randn(B * L, 3 * d, device=dev)
lens = torch.clamp(torch.round(torch.randn(B, device=dev) * 8 + 35), 15, L).long()
lens[0] = L
if fixed:
    lens[:] = L
pad = torch.arange(L, device=dev)[None, :] < lens[:, None]
if causal:
    mask = (pad[:, None, :] & torch.ones(L, L, dtype=torch.bool, device=dev).tril_()[None]).to(torch.uint8).contiguous()
else:
    mask = pad.to(torch.uint8).contiguous()
mb = ops.pack_mask(mask, B, L, L)
q, k, v = qkv, qkv[:, d:], qkv[:, 2 * d:]
for _ in range(5):
    o, lse, _ = ops.attn_fwd(q, k, v, 3 * d, 3 * d, 3 * d, mb, B, H, L, L, dk, 0.1, 1, 1)
do = torch.randn_like(o)
dqkv = torch.empty_like(qkv)
for _ in range(5):
    ops.attn_bwd(q, k, v, 3 * d, 3 * d, 3 * d, mb, o, do, lse, dqkv, dqkv[:, d:], dqkv[:, 2 * d:], 3 * d, 3 * d, 3 * d,
                 B, H, L, L, dk, 0.1, 1, 1)
torch.cuda.synchronize()
print("done")
