#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (counter_collection.csv) per kernel into the JSON files bench.py and DESIGN.md cite.

  python tools/pmc_summary.py --commit <sha> --out-traffic profiles/rNN_gemm_x6_traffic.json \
         --out-busy profiles/rNN_mfma_busy.json  <dir with the passes' CSVs> [...]

Counters are collected in SEPARATE passes (MI355X_MICROARCH.md, rocprofv3 PMC slots: FETCH_SIZE costs 3 TCC slots and
WRITE_SIZE 2 -- they cannot share a pass); every pass also carries the dispatch timestamps, so durations come from
the same rows.  gfx950 correction (same guide, section HBM): FETCH_SIZE counts 128-B requests at 64 B, so
  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024        [both counters are in KiB].
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); clock = GRBM_GUI_ACTIVE / 8 / duration.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re

SHORT = [(r"gemm_x6_kernel<0>", "gemm_x6_kernel<0> (forward)"), (r"gemm_x6_kernel<1>", "gemm_x6_kernel<1> (dgrad)"),
         (r"gemm_x6_kernel<2>", "gemm_x6_kernel<2> (wgrad)"),
         (r"gemm_x6s_kernel<0>", "gemm_x6s_kernel<0> (forward, 64x128 tiles)"),
         (r"gemm_x6s_kernel<1>", "gemm_x6s_kernel<1> (dgrad, 64x128 tiles)"),
         (r"splitk_epilogue_kernel", "splitk_epilogue_kernel"), (r"reduce_slabs2_kernel", "reduce_slabs2_kernel"),
         (r"reduce_multi_kernel", "reduce_multi_kernel (a layer's slab reductions)"),
         (r"attn_fwd_kernel<4", "attn_fwd_kernel<4,..>"),
         (r"attn_bwd_kernel<4", "attn_bwd_kernel<4,..>"), (r"attn_fwd_direct_kernel<4", "attn_fwd_direct_kernel<4,6>"),
         (r"attn_bwd_dq_kernel<4", "attn_bwd_dq_kernel<4,6>"), (r"attn_bwd_dkv_kernel<4", "attn_bwd_dkv_kernel<4,6>"),
         (r"norm_fwd_kernel", "norm_fwd_kernel"),
         (r"norm_bwd_kernel", "norm_bwd_kernel"), (r"adam_kernel", "adam_kernel"),
         (r"gather_quads_kernel", "gather_quads_kernel"), (r"dropout_bwd_kernel", "dropout_bwd_kernel"),
         (r"gemm_f32_fast_kernel", "gemm_f32_fast_kernel"), (r"embed_pe_bwd_kernel", "embed_pe_bwd_kernel")]


def short(name):
    for pat, s in SHORT:
        if pat in name:
            return s
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--commit", default="unknown")
    ap.add_argument("--command", default="")
    ap.add_argument("--out-traffic")
    ap.add_argument("--out-busy")
    a = ap.parse_args()
    # kernel -> counter -> [values]; kernel -> [durations]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    durs = collections.defaultdict(list)
    for d in a.dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = set()
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k is None:
                    continue
                vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                key = (f, r["Dispatch_Id"])
                if key not in seen:
                    seen.add(key)
                    durs[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    avg = lambda x: sum(x) / len(x) if x else None                                                # noqa: E731
    traffic, busy = {}, {}
    for k in sorted(vals):
        c = vals[k]
        dur_ns = avg(durs[k])
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            f, w = avg(c["FETCH_SIZE"]), avg(c["WRITE_SIZE"])
            hb = (2 * f + w) * 1024
            traffic[k] = {"launches": len(c["FETCH_SIZE"]), "FETCH_SIZE_KB_avg": round(f, 1),
                          "WRITE_SIZE_KB_avg": round(w, 1), "hbm_bytes_per_launch": int(hb),
                          "avg_duration_us_under_pmc": round(dur_ns / 1e3, 1),
                          "hbm_GBps_under_pmc": round(hb / dur_ns, 1)}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            m, g = avg(c["SQ_VALU_MFMA_BUSY_CYCLES"]), avg(c["GRBM_GUI_ACTIVE"])
            busy[k] = {"launches": len(c["GRBM_GUI_ACTIVE"]), "avg_duration_us": round(dur_ns / 1e3, 1),
                       "SQ_VALU_MFMA_BUSY_CYCLES_avg": int(m), "GRBM_GUI_ACTIVE_avg": int(g),
                       "clock_GHz": round(g / 8 / dur_ns, 3), "mfma_busy_frac": round(m / (1024 * g / 8), 4)}
            for extra in ("SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU",
                          "SQ_INSTS_MFMA", "SQ_LDS_BANK_CONFLICT"):
                if extra in c:
                    busy[k][extra + "_avg"] = int(avg(c[extra]))
    head = {"commit": a.commit, "command": a.command,
            "source": "rocprofv3 --pmc <one counter group per pass> --kernel-trace, program directly after `--`",
            "correction": "hbm = (2*FETCH_SIZE + WRITE_SIZE)*1024 B (gfx950: FETCH_SIZE counts 128-B requests at 64 B)"}
    if a.out_traffic and traffic:
        out = dict(head, kernel="gemm_x6_kernel<0> (nn.Linear forward, bf16x6)", all=traffic)
        if "adam_kernel" in traffic:
            out["steps_profiled"] = traffic["adam_kernel"]["launches"]          # one Adam launch per training step
        if "gemm_x6_kernel<0> (forward)" in traffic:
            out["hbm_bytes_per_launch"] = traffic["gemm_x6_kernel<0> (forward)"]["hbm_bytes_per_launch"]
        json.dump(out, open(a.out_traffic, "w"), indent=1)
        print("wrote", a.out_traffic)
    if a.out_busy and busy:
        json.dump(dict(head, formula="mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8); "
                                     "clock = GRBM_GUI_ACTIVE / 8 / duration", kernels=busy),
                  open(a.out_busy, "w"), indent=1)
        print("wrote", a.out_busy)
    print(json.dumps({"traffic": traffic, "busy": busy}, indent=1)[:3000])


if __name__ == "__main__":
    main()
