#!/usr/bin/env python3
"""Headline benchmark: SMILES/s of the vaetf training step on MI355X (BASELINE.json).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" = forward + CE/KL loss + backward + (RCCL gradient all-reduce) + fused Adam + LR write
of the configuration BASELINE.json's metric is quoted on (configs[1]): vaetf 6+6 layers,
d_model 512, 8 heads, d_ff 2048, latent 128, batch 512 per GPU, seq_len 80, dropout 0.1,
fp32, synthetic MOSES-shaped token batches resident in HBM before the timed region.
Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the forward GEMM: the bf16x6
kernel by default, the fp32-MFMA kernel under GCT_GEMM_MODE=f32) timed live with HIP events on its
launch stream inside the timed region;
`cpu_baseline` times the CPU oracle (port of the reference step) on the host cores at N=1.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: bf16 MFMA, dense (no sparsity)
FLOP_PER_SMILES_STEP = 22.09e9    # 3 x 7.363 GFLOP forward (vaetf, S=80, T=81)


def cpu_baseline(batch_size=64, steps=3, warm=1, dropout=0.1):
    """The reference's training step restated on CPU (oracle/gct_oracle.py, pinned to the
    reference by tests/golden) on config 1: vaetf, B=64, S=80, Adam.  Bounded sample."""
    from oracle import gct_oracle as O
    from gct_plus_amd import synthetic
    # the GPU box gives one GPU's share of the host: 16 cores (not the 128 torch reports)
    torch.set_num_threads(min(16, os.cpu_count() or 16))
    cfg = O.make_cfg("vaetf", 28, 30, dropout=dropout, nconds=0, use_cond2lat=True)
    P = O.make_leaves(O.init_state(cfg, seed=1))
    opt = O.make_adam(O.trainable(P, cfg))
    ds = synthetic.make_dataset(batch_size * (steps + warm), 80, "vaetf", seed=0)
    times = []
    for i, b in enumerate(synthetic.batches(ds, batch_size)):
        t0 = time.perf_counter()
        O.train_step(P, cfg, opt, b, 0.04, synthetic.PAD_ID, i + 1)
        times.append(time.perf_counter() - t0)
    t = sum(times[warm:]) / max(1, len(times) - warm)
    return {"value": round(batch_size / t, 2), "unit": "SMILES/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{steps} steps after {warm} warm-up, vaetf B={batch_size} S=80 dropout {dropout} "
                      f"Adam, torch {torch.__version__} CPU, {t*1e3:.0f} ms/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="per-GPU batch (BASELINE configs[1])")
    ap.add_argument("--model-type", default="vaetf")
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--all-kernel-timing", action="store_true")
    ap.add_argument("--no-alt-mode", action="store_true", help="skip the short fp32-MFMA-mode comparison run")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world == 1:
        sys.exit("bench.py --gpus N>1 must be launched through torch.distributed.run (see docstring)")
    if world != a.gpus:
        sys.exit(f"WORLD_SIZE={world} does not match --gpus {a.gpus}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from gct_plus_amd import ops, synthetic
    from gct_plus_amd.Model import forward_propagation, model_dict
    from gct_plus_amd.Train.trainer1 import loss_function, warmup_lr
    from gct_plus_amd.dp import FlatDataParallel
    from gct_plus_amd.optim import FusedAdam

    main_mode = ops.gemm_get_mode()
    mtype = a.model_type
    vs, vt = synthetic.vocab_sizes(mtype)
    nc = synthetic.n_conds(mtype)
    torch.manual_seed(1)
    model = model_dict[mtype](vs, vt, N=6, d_model=512, dff=2048, h=8, latent_dim=128,
                              dropout=a.dropout, nconds=nc, use_cond2dec=False, use_cond2lat=True)
    model = model.cuda().train()
    inner = model
    if world > 1:
        model = FlatDataParallel(model)
    opt = FusedAdam(inner.parameters(), lr=1e-4, betas=(0.9, 0.98), eps=1e-9, model=inner)

    # synthetic MOSES-shaped pool, sharded like DistributedSampler, staged in HBM up front
    n_pool = 4
    ds = synthetic.make_dataset(a.batch * n_pool * world, 80, mtype, seed=0)
    idx = synthetic.shard_indices(ds["src"].size(0), world, rank, epoch=0, seed=0, shuffle=False)
    shard = {k: v[idx] for k, v in ds.items()}
    pool = [{k: v.to(dev) for k, v in b.items()} for b in synthetic.batches(shard, a.batch)]
    # every rank's first pool batch must pad to S=80 (sample 0 only lives on rank 0)
    pad_id, beta = synthetic.PAD_ID, 0.04
    losses = torch.zeros(3, device=dev)

    def step(i):
        batch = pool[i % len(pool)]
        prop, mol, mu, lv, _ = forward_propagation[mtype](model, batch, pad_id, False)
        ys = batch["trg"][:, 1:].contiguous().view(-1)
        ys_cond = batch["dconds"].unsqueeze(2).contiguous().view(-1, nc, 1) if nc else None
        opt.zero_grad(set_to_none=True)
        loss, rce, _, kld = loss_function(beta, prop, mol, ys_cond, ys, mu, lv, False, pad_id)
        loss.backward()
        opt.step()
        lr = warmup_lr(i + 1, 512, 8000)
        for g in opt.param_groups:
            g["lr"] = lr
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    fence()
    if not a.no_kernel_timing:
        ops.PROFILE = {}
        # only the roofline kernel is bracketed by events inside the timed region (all three GEMM
        # kinds cost ~2 % of the step in event overhead); --all-kernel-timing restores the rest
        ops.PROFILE_KINDS = None if a.all_kernel_timing else {"gemm_fwd"}
    t0 = time.perf_counter()
    for i in range(a.warmup, a.warmup + a.steps):
        last = step(i)
    fence()
    dt = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    final_loss = float(last.item()) / a.batch

    # the same step with the GEMMs on the fp32 MFMA pipe (v_mfma_f32_32x32x2_f32), for reference: a short
    # second timed region after the headline one (every rank runs it, so the collectives stay matched)
    alt = None
    if main_mode == ops.GEMM_BF16X6 and not a.no_alt_mode:
        ops.gemm_set_mode(ops.GEMM_F32)
        k2 = max(3, a.steps // 3)
        for i in range(2):
            step(a.warmup + a.steps + i)
        fence()
        t1 = time.perf_counter()
        for i in range(k2):
            step(a.warmup + a.steps + 2 + i)
        fence()
        dt2 = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([dt2], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt2 = float(tt.item())
        ops.gemm_set_mode(main_mode)
        alt = {"gemm_arithmetic": "fp32 MFMA (v_mfma_f32_32x32x2_f32)", "steps": k2,
               "ms_per_step": round(dt2 / k2 * 1e3, 3), "value": round(a.batch * world * k2 / dt2, 1)}

    if rank == 0:
        ms = dt / a.steps * 1e3
        value = a.batch * world * a.steps / dt
        roof = None
        kern = {}
        if prof:
            klaunch = {k.split(":", 1)[1]: v for k, v in prof.items() if k.startswith("_x6_kernel_launches:")}
            for kind, recs in prof.items():
                if kind.startswith("_"):
                    continue
                tsum = sum(e0.elapsed_time(e1) for _, e0, e1 in recs) * 1e-3
                fsum = sum(f for f, _, _ in recs)
                kern[kind] = {"launches": len(recs), "avg_us": round(tsum / len(recs) * 1e6, 1),
                              "tflops": round(fsum / tsum / 1e12, 2),
                              "share_of_step": round(tsum / dt, 3)}
                if klaunch.get(kind):        # GEMM calls vs gemm_x6_kernel launches (tail-balanced calls make two)
                    kern[kind]["kernel_launches"] = klaunch[kind]
                    kern[kind]["avg_kernel_us"] = round(tsum / klaunch[kind] * 1e6, 1)
            x6 = "gemm_fwd[x6]" in kern
            dom = "gemm_fwd[x6]" if x6 else "gemm_fwd"
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "r01_gemm_x6_traffic.json" if x6 else "r01_gemm_fwd_traffic.json")
            if os.path.exists(pmc):
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            if x6:
                # bf16 MFMA pipe, six bf16 partial products per fp32 product: the fp32-equivalent
                # ceiling of the kernel is the dense bf16 peak / 6
                peak = PEAK_BF16_MFMA_TFLOPS / 6.0
                roof = {"kernel": "gemm_x6_kernel<FWD> (nn.Linear forward, exact 3-way bf16 split of both fp32 "
                                  "operands, 6 partial products on v_mfma_f32_16x16x32_bf16, fp32 accumulate)",
                        "bound": "mfma", "achieved": kern[dom]["tflops"], "peak": round(peak, 1),
                        "unit": "TFLOP/s", "frac": round(kern[dom]["tflops"] / peak, 4),
                        "traffic": traffic, "avg_launch_us": kern[dom].get("avg_kernel_us", kern[dom]["avg_us"]),
                        "launches": kern[dom].get("kernel_launches", kern[dom]["launches"]),
                        "gemm_calls": kern[dom]["launches"], "avg_call_us": kern[dom]["avg_us"],
                        "note": "avg_launch_us = time of all forward GEMM calls / gemm_x6_kernel<FWD> launches (a "
                                "tail-balanced call launches the kernel twice plus a small fix-up kernel, which is "
                                "inside the bracket); achieved/peak in fp32-equivalent (algorithmic) FLOP/s; on the pipe itself: "
                                f"{round(6 * kern[dom]['tflops'], 1)} of {PEAK_BF16_MFMA_TFLOPS} bf16 TFLOP/s; "
                                f"the fp32 MFMA pipe these GEMMs ran on before peaks at {PEAK_F32_MFMA_TFLOPS}",
                        "kernels": kern}
            else:
                roof = {"kernel": "gemm_f32_fast_kernel<true,true> (nn.Linear forward, every shape of the step)",
                        "bound": "mfma", "achieved": kern[dom]["tflops"], "peak": PEAK_F32_MFMA_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(kern[dom]["tflops"] / PEAK_F32_MFMA_TFLOPS, 4),
                        "traffic": traffic, "avg_launch_us": kern[dom]["avg_us"],
                        "launches": kern[dom]["launches"], "kernels": kern}
        out = {
            "metric": "SMILES/sec training step (vaetf, seq_len=80, d_model=512)",
            "value": round(value, 1), "unit": "SMILES/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "gemm_arithmetic": ("bf16x6: fp32 operands split exactly into 3 bf16 pieces, 6 partial products, fp32 "
                                "accumulate (error vs fp64 <= the fp32 fma chain's; GCT_GEMM_MODE=f32 selects "
                                "v_mfma_f32_32x32x2_f32)") if ops.gemm_get_mode() == ops.GEMM_BF16X6
                               else "fp32 MFMA (v_mfma_f32_32x32x2_f32)",
            "config": {"workload": f"{mtype} training step: 6+6 layers d_model=512 h=8 d_ff=2048 "
                                   f"latent=128, batch {a.batch}/GPU, seq_len=80 (T=81), dropout "
                                   f"{a.dropout}, CE+KL loss, fused Adam, fp32 (BASELINE configs[1])",
                       "global_batch": a.batch * world, "seq_len": 80,
                       "parallelism": f"dp{world}"},
            "step_tflops_algorithmic": round(FLOP_PER_SMILES_STEP * value / 1e12, 2),
            "final_loss_per_sample": round(final_loss, 4),
            "roofline": roof,
        }
        if alt is not None:
            out["same_step_fp32_mfma_gemms"] = alt
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
