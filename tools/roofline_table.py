#!/usr/bin/env python3
"""Per-kernel roofline table of one profile round (tools/profile_round.sh <tag>):
    python tools/roofline_table.py <training_only_kernel_stats.csv> <traffic.json> <mfma_busy.json> [steps] > table.md
Every kernel of the training step above 0.1 ms per step against its bound: MFMA-busy fraction (PMC) scaled to the bf16
pipe at 2.4 GHz for the GEMMs, HBM bytes by the counters / duration under the counters against 8 TB/s for the rest."""
import csv
import json
import re
import sys

stats, traffic, busy = sys.argv[1:4]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 16
rows = list(csv.DictReader(open(stats)))
T = json.load(open(traffic))["all"]
B = json.load(open(busy))["kernels"]


def find(d, name):
    for k in d:
        if k.split(" ")[0].split("<")[0] in name and (("<" not in k) or k.split(" ")[0].rstrip(",.>") in name.replace(", ", ",")):
            return d[k]
    return None


print("| kernel | launches / step | ms / step | bound | achieved | fraction of the bound |")
print("|---|---|---|---|---|---|")
tot_ns = tot_calls = 0
for r in rows:
    name, calls, ns = r["Name"], int(r["Calls"]), int(r["TotalDurationNs"])
    if "mfma_probe" in name:
        continue
    tot_ns += ns
    tot_calls += calls
    if ns / steps < 1e5:
        continue
    m = re.search(r"(?:::)?(\w+(?:<[^>]*>)?)\(", name)
    short = m.group(1) if m else name[:40]
    t, b = find(T, name), find(B, name)
    if "gemm_x6" in name and b:
        f = b["mfma_busy_frac"] * b["clock_GHz"] / 2.4
        print(f"| `{short}` | {calls / steps:.0f} | {ns / steps / 1e6:.2f} | mfma | {100 * b['mfma_busy_frac']:.1f} % MFMA-busy at "
              f"{b['clock_GHz']:.2f} GHz | {f:.2f} of the bf16 pipe at 2.4 GHz |")
    elif t:
        print(f"| `{short}` | {calls / steps:.0f} | {ns / steps / 1e6:.2f} | hbm | {t['hbm_GBps_under_pmc'] / 1e3:.2f} TB/s "
              f"({t['hbm_bytes_per_launch'] / 1e6:.0f} MB per launch) | {t['hbm_GBps_under_pmc'] / 8000:.2f} |")
    else:
        print(f"| `{short}` | {calls / steps:.0f} | {ns / steps / 1e6:.2f} | | | |")
print(f"| all kernels | {tot_calls / steps:.0f} | {tot_ns / steps / 1e6:.2f} | | | |")
