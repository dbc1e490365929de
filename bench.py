#!/usr/bin/env python3
"""Headline benchmark: SMILES/s of the vaetf training step on MI355X (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W            # N = 1, 2, 4, 8 on ONE node
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Started plainly with --gpus N > 1 (no RANK/WORLD_SIZE in the environment) this file is its own
launcher: the parent process touches no GPU, picks a free rendezvous port on 127.0.0.1 and starts
N fresh `python bench.py` worker processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set
(one process per GPU, the layout of the reference's train1.py:152-171); rank 0's stdout is the
parent's stdout, so exactly ONE JSON line comes out.  Under torch.distributed.run the same worker
code runs directly.

A "step" = forward + CE/KL loss + backward + (RCCL gradient all-reduce) + fused Adam + LR write
of the configuration BASELINE.json's metric is quoted on (configs[1]): vaetf 6+6 layers,
d_model 512, 8 heads, d_ff 2048, latent 128, batch 512 per GPU, seq_len 80, dropout 0.1,
fp32, synthetic MOSES-shaped token batches resident in HBM before the timed region
(--model-type scavaetf --gpus 8 is configs[3]: global batch 4096).
Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the forward GEMM: the bf16x6
kernel by default, the fp32-MFMA kernel under GCT_GEMM_MODE=f32) timed live with HIP events on its
launch stream inside the timed region; `fixed_len_80` is the same step on a batch without padding
(every sample 80 tokens: the worst case, none of the zero-row shortcuts apply); `decode` is
BASELINE configs[4]'s metric (KV-cached greedy decode of pscavaetf, decoded SMILES/s, at n = 4096 and n = 512); `cpu_baseline` times the
CPU oracle (port of the reference step) on the host cores at N=1.

--selftest-cpu swaps the HIP model for a few-kB plain-torch stand-in on the CPU so that THIS file's
launcher, rendezvous, sharding, step loop, fences, max-over-ranks timing and JSON line can be
driven end to end by `tests/test_bench_launcher.py` with two gloo ranks and no GPU; its line is
labelled as a self-test and carries no performance claim.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: bf16 MFMA, dense (no sparsity)
FLOP_PER_SMILES_STEP = {"vaetf": 22.09e9, "scavaetf": 22.09e9,     # 3 x 7.363 GFLOP forward
                        "pvaetf": 22.58e9, "pscavaetf": 22.58e9}   # 3 x 7.528 (n_c = 3)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # >= 50 timed
    ap.add_argument("--warmup", type=int, default=10)     # ... after >= 10 warm-up steps
    ap.add_argument("--batch", type=int, default=512, help="per-GPU batch (BASELINE configs[1])")
    ap.add_argument("--model-type", default="vaetf", choices=["vaetf", "pvaetf", "scavaetf", "pscavaetf"])
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--fixed-len", action="store_true", help="headline region on unpadded (all 80-token) batches")
    ap.add_argument("--dense-decoder", action="store_true",
                    help="compute every decoder row in the forward pass (the trainer skips the rows of padded targets)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--all-kernel-timing", action="store_true")
    ap.add_argument("--no-alt-mode", action="store_true", help="skip the short fp32-MFMA-mode comparison run")
    ap.add_argument("--no-fixed-len-leg", action="store_true", help="skip the short fixed_len_80 region")
    ap.add_argument("--no-decode", action="store_true", help="skip the decode block (configs[4] metric)")
    ap.add_argument("--no-model-types", action="store_true", help="skip the short legs of the other three model types")
    ap.add_argument("--no-trainer-loop", action="store_true", help="skip the Train.trainer1.run_epoch legs")
    ap.add_argument("--all-legs", action="store_true",
                    help="N > 1 runs skip the model_types and trainer_loop legs (the scaling runs measure the headline step and "
                         "report the exchange; the 1-GPU line carries those legs) unless this is given")
    ap.add_argument("--decode-n", type=int, default=4096, help="sequences decoded per GPU (the sampler's n is a free parameter)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the product path); gloo only for the launcher tests")
    ap.add_argument("--share-gpu", action="store_true",
                    help="test rigs with ONE GPU: every rank uses cuda:0 and gradients are exchanged through "
                         "host memory over gloo (RCCL refuses two ranks on one device)")
    ap.add_argument("--tiny", action="store_true", help="N=2 d_model=64 d_ff=128 h=4 latent=16 (launcher tests)")
    ap.add_argument("--selftest-cpu", action="store_true", help="launcher/step-loop self-test on CPU (see docstring)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------- launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_workers(a, argv):
    """Parent of a plain `bench.py --gpus N` call.  Touches no GPU (no torch import here): N fresh
    interpreters, each of which initialises its own device after it has its rank."""
    port = _free_port()
    procs = []
    ncpu = os.cpu_count() or 8
    for r in range(a.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GCT_BENCH_LAUNCHER="self-spawn")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, ncpu // a.gpus)))
        # rank 0 owns the parent's stdout (the ONE JSON line); other ranks' stdout goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    alive = list(procs)
    while alive:                                   # a dead rank must not leave the others in a collective
        for p in list(alive):
            c = p.poll()
            if c is None:
                continue
            alive.remove(p)
            if c != 0 and rc == 0:
                rc = c
                for q in alive:
                    q.terminate()
        time.sleep(0.05)
    return rc


# ------------------------------------------------------------------------------ cpu baseline
def cpu_baseline(batch_size=64, steps=10, warm=3):
    """The reference's training step restated on CPU (oracle/gct_oracle.py, pinned to the
    reference by tests/golden) on config 1: vaetf, B=64, S=80, Adam.  SURVEY.md 8(d): median of
    >= 10 steps after 3 warm-up, with dropout 0.1 (the headline's setting) and with dropout 0
    (CPU bernoulli_ is a large share of the CPU step).  Bounded sample: ~26 steps of ~2-3 s."""
    import statistics
    import torch
    from oracle import gct_oracle as O
    from gct_plus_amd import synthetic
    # the GPU box gives one GPU's share of the host: 16 cores (not the 128 torch reports)
    torch.set_num_threads(min(16, os.cpu_count() or 16))
    res = {}
    for dropout in (0.1, 0.0):
        cfg = O.make_cfg("vaetf", 28, 30, dropout=dropout, nconds=0, use_cond2lat=True)
        P = O.make_leaves(O.init_state(cfg, seed=1))
        opt = O.make_adam(O.trainable(P, cfg))
        ds = synthetic.make_dataset(batch_size * (steps + warm), 80, "vaetf", seed=0)
        times = []
        for i, b in enumerate(synthetic.batches(ds, batch_size)):
            t0 = time.perf_counter()
            O.train_step(P, cfg, opt, b, 0.04, synthetic.PAD_ID, i + 1)
            times.append(time.perf_counter() - t0)
        res[dropout] = statistics.median(times[warm:])
    t = res[0.1]
    return {"value": round(batch_size / t, 2), "unit": "SMILES/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"median of {steps} steps after {warm} warm-up, vaetf B={batch_size} S=80 dropout 0.1 "
                      f"Adam, torch {torch.__version__} CPU, {t*1e3:.0f} ms/step",
            "dropout0": {"value": round(batch_size / res[0.0], 2), "ms_per_step": round(res[0.0] * 1e3)}}


# ---------------------------------------------------------------------------------- workloads
def selftest_workload(a, dev, world, rank):
    """Plain-torch stand-in (CPU): same batch dict, same FlatDataParallel wrapper, same loop."""
    import torch
    import torch.nn as nn
    from gct_plus_amd import synthetic
    from gct_plus_amd.dp import FlatDataParallel
    from gct_plus_amd.flat import FlatModelMixin

    class Stub(FlatModelMixin, nn.Module):
        def __init__(self, vs, vt):
            super().__init__()
            self.emb = nn.Embedding(vs, 16)
            self.out = nn.Linear(16, vt)

        def forward(self, src, trg):
            h = self.emb(src).mean(1, keepdim=True).expand(-1, trg.size(1), -1)
            return self.out(torch.tanh(h))

    if os.environ.get("GCT_BENCH_SELFTEST_DIE") == str(rank):    # launcher test: a rank that dies early
        sys.exit(3)
    mtype = a.model_type
    vs, vt = synthetic.vocab_sizes(mtype)
    torch.manual_seed(1 + rank)                   # ranks differ until the wrapper's broadcast
    inner = Stub(vs, vt)
    inner.flatten_parameters()
    model = FlatDataParallel(inner) if world > 1 else inner
    opt = torch.optim.Adam(inner.parameters(), lr=1e-3)

    def fwd_loss(batch):
        logits = model(batch["src"], batch["trg"][:, :-1])
        ys = batch["trg"][:, 1:].reshape(-1)
        return nn.functional.cross_entropy(logits.reshape(-1, vt), ys, ignore_index=synthetic.PAD_ID,
                                           reduction="sum")
    fwd_loss.model = model
    return inner, opt, fwd_loss


def hip_workload(a, dev, world, rank, mtype=None):
    import torch
    from gct_plus_amd import synthetic
    from gct_plus_amd.Model import forward_propagation, model_dict
    from gct_plus_amd.Train.trainer1 import loss_function
    from gct_plus_amd.dp import FlatDataParallel
    from gct_plus_amd.optim import FusedAdam

    mtype = mtype or a.model_type
    vs, vt = synthetic.vocab_sizes(mtype)
    nc = synthetic.n_conds(mtype)
    dims = dict(N=2, d_model=64, dff=128, h=4, latent_dim=16) if a.tiny else \
        dict(N=6, d_model=512, dff=2048, h=8, latent_dim=128)
    torch.manual_seed(1)
    inner = model_dict[mtype](vs, vt, dropout=a.dropout, nconds=nc, use_cond2dec=False, use_cond2lat=True, **dims)
    inner = inner.to(dev).train()
    model = inner
    if world > 1:
        kw = {}
        if a.share_gpu:
            from gct_plus_amd.testing import host_staged_allreduce, host_staged_broadcast
            kw = dict(allreduce=host_staged_allreduce, broadcast=host_staged_broadcast)
        model = FlatDataParallel(inner, **kw)
    opt = FusedAdam(inner.parameters(), lr=1e-4, betas=(0.9, 0.98), eps=1e-9, model=inner)
    pad_id, beta = synthetic.PAD_ID, 0.04

    def fwd_loss(batch):
        # exactly what Train/trainer1.run_epoch does per step (skip_ignored: decoder rows of padded targets are not
        # computed -- they never reach the ignore_index loss; --dense-decoder measures without the shortcut)
        prop, mol, mu, lv, _ = forward_propagation[mtype](model, batch, pad_id, False, skip_ignored=not a.dense_decoder)
        ys = batch["trg"][:, 1:].contiguous().view(-1)
        ys_cond = batch["dconds"].unsqueeze(2).contiguous().view(-1, nc, 1) if nc else None
        loss, _, _, _ = loss_function(beta, prop, mol, ys_cond, ys, mu, lv, False, pad_id)
        return loss
    fwd_loss.model = model            # the data-parallel wrapper at N > 1 (what a trainer calls), else the model itself

    def announce(batch):
        # run_epoch's look-ahead: the next batch's masks and row maps are queued between this step's forward and its
        # backward, so their read-back is on the host before the next forward asks for it
        from gct_plus_amd.Model.forward_propagation1 import prefetch
        prefetch(mtype, model, batch, pad_id, False, skip_ignored=not a.dense_decoder)
    fwd_loss.prefetch = announce
    return inner, opt, fwd_loss


def decode_block(train_model, a, dev, world, fence, reduce_max):
    """BASELINE configs[4] metric: decoded SMILES/s of the KV-cached greedy decode of **pscavaetf** (3 conditions,
    <sos> scaffold <sep> prefix -- the reference's Inference/sampling_tool.py:452-498 front end around the loop of
    :140-184), n sequences per GPU, max_strlen 80 => 79 generated tokens, no early stop (eos never matches: the worst
    case).  Latent length 40 + n_c: the reference draws the latent length from the training-set token-length
    distribution (MOSES mean ~35-40).  Two legs: n = --decode-n (4096: the chip is full) and n = 512 (the reference's
    chunk size, Inference/uc_sampling.py:16-23).  Whatever the training leg's model type, this block decodes pscavaetf.
    The step is captured into one hipGraph and replayed; KVDecoder's replay guard times replay against eager launches of
    the same step on this box and falls back (and says so) where replay is the slower mode -- `launch_mode` names what
    the timed figure used, never an eager figure under the graph's name."""
    import warnings
    import torch
    from gct_plus_amd import graphdiag, synthetic
    from gct_plus_amd.Model import model_dict
    from gct_plus_amd.decode import KVDecoder
    mtype = "pscavaetf"
    nc = synthetic.n_conds(mtype)
    metric = "decoded SMILES/sec (KV-cached greedy decode, pscavaetf, scaffold prefix, max_strlen 80)"
    was_training = train_model.training
    err, model = None, None
    try:
        if a.model_type == mtype:
            model = train_model
        else:
            vs, vt = synthetic.vocab_sizes(mtype)
            torch.manual_seed(1)
            model = model_dict[mtype](vs, vt, dropout=a.dropout, nconds=nc, use_cond2dec=False, use_cond2lat=True,
                                      N=6, d_model=512, dff=2048, h=8, latent_dim=128).to(dev)
        model.eval()
    except Exception as exc:                                      # noqa: BLE001
        err = repr(exc)
    Le = 40 + nc
    lat = 128
    pre_len = 10                                                  # scaffold tokens between <sos> and <sep>

    def leg(n):
        """One decode leg.  No collective sits inside a try: a rank that fails still reaches every fence and
        max-reduction, so a decode problem on one rank can never hang the job or cost it the headline line."""
        g = torch.Generator().manual_seed(5)
        z = torch.randn(n, Le, lat, generator=g).to(dev)
        dconds = torch.randn(n, nc, generator=g).to(dev)
        src_mask = torch.ones(n, 1, Le, dtype=torch.bool, device=dev)
        scaf = torch.randint(5, 30, (n, pre_len), generator=g)
        ys0 = torch.cat([torch.full((n, 1), synthetic.SOS_ID), scaf, torch.full((n, 1), synthetic.SEP_ID)], 1).to(dev)
        e, kd, ys, prefill_ms = err, None, None, None
        if e is None:
            try:
                kd = KVDecoder(model, synthetic.PAD_ID, synthetic.SOS_ID, eos_id=-1)
                kd.start(z, src_mask, dconds, max_total_len=96)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")               # the guard's message is reported in the JSON instead
                    kd.generate(ys0, 80, use_graphs=True, check_every=0)      # warm-up, graph capture, replay guard
                torch.cuda.synchronize()
                t0 = time.perf_counter()                          # the prefill alone (one decoder forward over the prefix)
                kd.prefill(ys0)
                torch.cuda.synchronize()
                prefill_ms = (time.perf_counter() - t0) * 1e3
            except Exception as exc:                              # noqa: BLE001
                e = repr(exc)
        fence()
        reps = []                                                 # three repetitions, the median counts (all are reported)
        for _ in range(3):
            dt_local = 1e30                                       # 1e30: "this rank failed" through the MAX reduction
            if e is None:
                try:
                    t0 = time.perf_counter()
                    kd.start(z, src_mask, dconds, max_total_len=96)   # per-sequence set-up and the prefill are inside
                    ys = kd.generate(ys0, 80, use_graphs=True, check_every=0)
                    torch.cuda.synchronize()
                    dt_local = time.perf_counter() - t0
                except Exception as exc:                          # noqa: BLE001
                    e = repr(exc)
            fence()
            reps.append(reduce_max(dt_local))
        dt = sorted(reps)[1]
        # replay that passed the guard's four-step probe but is slow over the whole sequence (all ranks decide together):
        # time the same call with eager launches and report whichever mode is faster, under its own name
        sus = 0.0
        if e is None and dt < 1e20 and kd.graph_replay and kd.replay_probe and ys is not None:
            per_tok = (dt * 1e3 - prefill_ms) / max(int(ys.shape[1]) - ys0.shape[1] - 1, 1)
            sus = 1.0 if per_tok > 2.0 * kd.replay_probe["ms_per_step_eager"] else 0.0
        if os.environ.get("GCT_BENCH_FORCE_EAGER_CHECK"):         # exercises the branch on a healthy box
            sus = 1.0
        eager_reps = []
        if reduce_max(sus) > 0.5:
            for _ in range(3):
                dt_local = 1e30
                if e is None:
                    try:
                        t0 = time.perf_counter()
                        kd.start(z, src_mask, dconds, max_total_len=96)
                        ys = kd.generate(ys0, 80, use_graphs=False, check_every=0)
                        torch.cuda.synchronize()
                        dt_local = time.perf_counter() - t0
                    except Exception as exc:                      # noqa: BLE001
                        e = repr(exc)
                fence()
                eager_reps.append(reduce_max(dt_local))
        if e is not None or dt > 1e20:
            return {"n_per_gpu": n, "value": None, "error": e or "another rank failed"}
        replay = bool(kd.graph_replay and kd.graphs.get(0) not in (None, False))
        replay_reps = reps
        if eager_reps and sorted(eager_reps)[1] < dt:
            dt, reps, replay = sorted(eager_reps)[1], eager_reps, False
        steps = int(ys.shape[1]) - ys0.shape[1] - 1               # tokens that came out of single-token steps
        out = {"value": round(n * world / dt, 1), "unit": "SMILES/s", "n_per_gpu": n,
               "generated_tokens": int(ys.shape[1]) - ys0.shape[1], "prefix_tokens": int(ys0.shape[1]), "latent_len": Le,
               "launch_mode": "hipGraph replay (one captured graph for every step)" if replay else
                              ("eager launches (graph replay passed the guard's probe but was the slower mode over the "
                               "whole sequence on this box)" if eager_reps and kd.graph_replay else
                               "eager launches (replay guard: graph replay is the slower mode on this box)"),
               "graph_replay": replay,
               "ms_per_token_step": round((dt * 1e3 - prefill_ms) / max(steps, 1), 3),
               "prefill_and_setup_ms": round(prefill_ms, 2), "repetitions_ms": [round(r * 1e3, 1) for r in reps],
               "replay_guard": kd.replay_probe}
        if eager_reps:
            out["whole_sequence_ms"] = {"replay": [round(r * 1e3, 1) for r in replay_reps],
                                        "eager": [round(r * 1e3, 1) for r in eager_reps]}
        return out

    legs = [leg(n) for n in dict.fromkeys((a.decode_n, 512))]
    if model is not None and model is train_model:
        model.train(was_training)
    head = legs[0]
    res = {"metric": metric, "model_type": mtype, **head, "n512": legs[1] if len(legs) > 1 else None}
    slow = [l for l in legs if (l.get("replay_guard") and
                                l["replay_guard"]["ms_per_step_graph"] > 1.3 * l["replay_guard"]["ms_per_step_eager"])
            or l.get("whole_sequence_ms")]
    if slow:
        # replay is slower than eager launches here: say what the box is and where the time goes (do-nothing kernels:
        # per-node cost, by-value kernarg fetch, dynamic-LDS opt-in, arguments behind a device pointer)
        try:
            res["graph_diagnosis"] = {"device_facts": graphdiag.device_facts(),
                                      "micro_probe_70_nodes": graphdiag.micro_probe(70, 10)}
        except Exception as exc:                                  # noqa: BLE001
            res["graph_diagnosis"] = {"error": repr(exc)}
        if int(os.environ.get("RANK", "0")) == 0:
            res["graph_diagnosis"]["runtime_switches"] = replay_switch_children()
    return res


def multi_rank_report(a, dpm, inner, host, ops, dev, world, rank, cpu, step, region, nxt):
    """What a first run on N real GPUs needs to be read without a debugger (every rank takes part; rank 0 reports):
    who is there (rank -> device), how the gradient exchange went (buckets launched from inside the backward pass vs at
    its end, device time the compute stream waited for the exchange), what the host spent per step, whether the ranks
    still hold bit-identical parameters, and the same step with the in-backward launches switched off."""
    import statistics
    import torch
    import torch.distributed as dist
    if cpu:
        ident = {"rank": rank, "device": "cpu", "pid": os.getpid()}
    else:
        pr = torch.cuda.get_device_properties(dev)
        ident = {"rank": rank, "device": f"cuda:{dev.index}", "name": pr.name, "pid": os.getpid(),
                 "uuid": str(getattr(pr, "uuid", "")), "pci_bus_id": getattr(pr, "pci_bus_id", None),
                 "visible": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")}
    mine = {"identity": ident,
            "host_ms_per_step": round(host["s"] / max(host["n"], 1) * 1e3, 3),
            "host_blocked_in_readback_ms_per_step": round(ops.HOST_BLOCKED_S[0] / max(host["n"], 1) * 1e3, 3) if ops else None,
            "exchange": dpm.diag_summary() if dpm is not None else None}
    every = [None] * world
    dist.all_gather_object(every, mine)
    # bit-identical parameters on every rank: MAX - MIN of an integer checksum of the flat parameter buffer
    with torch.no_grad():
        flat = inner.flat_params() if getattr(inner, "_gct_flat", None) is not None else \
            torch.cat([p.detach().reshape(-1) for p in inner.parameters()])
        cs = flat.view(torch.int32).to(torch.int64).sum().reshape(1)
        hi, lo = cs.clone(), cs.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    # A/B: the same step with every bucket launched from the end-of-backward callback (nothing overlapped)
    ab = None
    if dpm is not None:
        dpm.overlap = False
        step(nxt)
        dt2, _ = region(nxt + 1, 5)
        dpm.overlap = True
        ab = {"steps": 5, "ms_per_step": round(dt2 / 5 * 1e3, 3)}
    ex = [e["exchange"] for e in every if e and e.get("exchange")]
    waits = [e["exposed_wait_ms_median"] for e in ex if e.get("exposed_wait_ms_median") is not None]
    return {"ranks_seen": len([e for e in every if e]), "identities": [e["identity"] for e in every if e],
            "distinct_devices": len({(e["identity"].get("uuid") or e["identity"].get("pci_bus_id") or e["identity"]["device"])
                                     for e in every if e}),
            "host_ms_per_step": [e["host_ms_per_step"] for e in every if e],
            "host_blocked_in_readback_ms_per_step": [e["host_blocked_in_readback_ms_per_step"] for e in every if e],
            "exchange_rank0": ex[0] if ex else None,
            "exposed_wait_ms_median_over_ranks": round(statistics.median(waits), 3) if waits else None,
            "exposed_wait_ms_max_over_ranks": round(max(e["exposed_wait_ms_max"] for e in ex if e.get("exposed_wait_ms_max") is not None), 3)
            if waits else None,
            "parameter_checksum_max_minus_min": int((hi - lo).item()),
            "same_step_overlap_off": ab}


def synthetic_smiles_frame(n, seed=0):
    """n SMILES-like strings over 26 regular tokens (so that the vocabularies built from them have the 28 / 30 entries of
    the synthetic token batches and the benchmarked model can consume them), MOSES-like token counts N(35, 8) in [15, 78]."""
    import numpy as np
    import pandas as pd
    toks = ["C", "c", "N", "O", "n", "(", ")", "1", "2", "=", "F", "S", "o", "s", "3", "#", "Cl", "Br", "[nH]", "-",
            "4", "[C@H]", "[C@@H]", "/", "[O-]", "[N+]"]
    rng = np.random.default_rng(seed)
    lens = np.clip(np.round(rng.normal(35, 8, n)), 15, 78).astype(int)
    w = 1.0 / np.arange(1, len(toks) + 1)
    w /= w.sum()
    rows = []
    for i in range(n):
        ids = rng.choice(len(toks), size=lens[i], p=w)
        if i < len(toks):
            ids[0] = i                      # every token occurs
        rows.append("".join(toks[j] for j in ids))
    return pd.DataFrame({"src": rows, "trg": rows})


def trainer_loop_legs(a, inner, opt, state, step, region, fence, reduce_max, make_pool, dev, world, rank, nxt):
    import logging
    import tempfile
    import types
    from gct_plus_amd import data, synthetic
    from gct_plus_amd.Train.trainer1 import run_epoch
    mtype = a.model_type
    nc = synthetic.n_conds(mtype)
    tmp = tempfile.mkdtemp(prefix="gct_bench_")
    LOG = logging.getLogger("gct_bench_trainer")
    LOG.setLevel(logging.INFO)
    LOG.propagate = False
    fh = logging.FileHandler(os.path.join(tmp, "records.log"))     # the reference logs every step to records.log (+ console)
    fh.setFormatter(logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s"))
    LOG.addHandler(fh)
    args = types.SimpleNamespace(model_type=mtype, pad_id=synthetic.PAD_ID, use_cond2dec=False,
                                 property_list=["logP", "tPSA", "QED"][:nc], lr_scheduler="WarmUpDefault",
                                 lr_WarmUpSteps=8000, d_model=512, print_every=1)
    model = state.get("dp_model", inner)
    k = 20
    res = {"steps": k, "note": "Train.trainer1.run_epoch (the three loss scalars of every step read back -- one step late, behind the "
                               "next step's row plan --, records.log line and LR write every step, -print_every 1 as the "
                               "reference's default) over batches staged in HBM vs this file's bare step on the same "
                               "batches; ratio = bare / trainer (1.0: the loop costs nothing)"}

    def timed_epoch(loader, cur):
        fence()
        t0 = time.perf_counter()
        _, cur = run_epoch(args, model, opt, loader, cur, 0.04, LOG, train=True)
        fence()
        return reduce_max(time.perf_counter() - t0), cur

    cur = 100000
    for B in dict.fromkeys((a.batch, 128, 64)):
        pool_b = make_pool(False, batch=B)
        loader = [pool_b[i % len(pool_b)] for i in range(k)]
        _, cur = run_epoch(args, model, opt, loader[:3], cur, 0.04, LOG, train=True)     # warm-up (shapes, workspaces)
        dt_t, cur = timed_epoch(loader, cur)
        state["pool"] = pool_b
        for i in range(3):
            step(nxt + i)
        dt_b, _ = region(nxt + 3, k)
        nxt += 3 + k
        res[f"batch_{B}"] = {"trainer_ms_per_step": round(dt_t / k * 1e3, 3), "bare_ms_per_step": round(dt_b / k * 1e3, 3),
                             "trainer_smiles_per_s": round(B * world * k / dt_t, 1), "ratio": round(dt_b / dt_t, 4)}
    # behind the loader: native tokenizer + collate per batch (gct_plus_amd.data.SmilesLoader), unconditioned types only
    # (the conditioned ones need property columns; the loop is the same)
    if nc == 0 and mtype == "vaetf":
        B = a.batch
        frame = synthetic_smiles_frame(B * k * world, seed=3)
        SRC, TRG, _ = data.get_fields(mtype, os.path.join(tmp, "utils"), frame["src"].tolist())
        vs, vt = synthetic.vocab_sizes(mtype)
        if len(SRC) <= vs and len(TRG) <= vt:
            loader = data.SmilesLoader(frame, SRC, TRG, mtype, [], B, rank, world, shuffle=False, seed=0, device=dev)
            warm = data.SmilesLoader(frame.iloc[:3 * B * world], SRC, TRG, mtype, [], B, rank, world, shuffle=False, seed=0, device=dev)
            _, cur = run_epoch(args, model, opt, warm, cur, 0.04, LOG, train=True)
            dt_l, cur = timed_epoch(loader, cur)
            res["behind_loader"] = {"batch": B, "steps": len(loader), "ms_per_step": round(dt_l / len(loader) * 1e3, 3),
                                    "smiles_per_s": round(B * world * len(loader) / dt_l, 1),
                                    "note": "SMILES strings -> native tokenizer + collate (padded to the batch's longest row, so "
                                            "batches are shorter than the staged 80-wide ones) -> run_epoch"}
        else:
            res["behind_loader"] = {"skipped": f"vocabulary {len(SRC)}/{len(TRG)} larger than the model's {vs}/{vt}"}
    LOG.removeHandler(fh)
    fh.close()
    return res


def replay_switch_children(rows=512, budget_s=200.0):
    """Only when graph replay was the slower mode: the decode comparison of tools/graph_probe.py repeated in CHILD
    processes (fresh HIP runtimes; this process only waits) under the runtime switches that change where a graph's
    kernel arguments live and how its nodes are issued, so that the line names the switch that makes the difference on
    this box.  Bounded: stops starting children after budget_s seconds."""
    here = os.path.dirname(os.path.abspath(__file__))
    probe = os.path.join(here, "tools", "graph_probe.py")
    if not os.path.exists(probe):
        return {"error": "tools/graph_probe.py is not next to bench.py"}
    out, t0 = [], time.perf_counter()
    for ks in ({"HIP_FORCE_DEV_KERNARG": "1"}, {"HIP_FORCE_DEV_KERNARG": "0"}, {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"},
               {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "1"}, {"DEBUG_HIP_GRAPH_BATCH_SIZE": "1"}):
        if time.perf_counter() - t0 > budget_s:
            out.append({"env": ks, "skipped": "time budget"})
            continue
        env = dict(os.environ)
        env.update(ks)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        try:
            p = subprocess.run([sys.executable, probe, "--stage", "knob", "--rows", str(rows)], env=env,
                               capture_output=True, text=True, timeout=90)
            line = [l for l in p.stdout.splitlines() if l.startswith("CHILD ")]
            rec = json.loads(line[-1][6:]) if line else {"error": (p.stderr or p.stdout)[-300:]}
        except Exception as exc:                                  # noqa: BLE001
            rec = {"error": repr(exc)}
        d = rec.get("decode", {}).get(str(rows))
        out.append({"env": ks, **({"ms_per_token_graph": d["wall_ms_per_token_graph"],
                                   "ms_per_token_eager": d["wall_ms_per_token_eager"]} if d else rec)})
    return out


# ------------------------------------------------------------------------------------ worker
def worker(a):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit(f"WORLD_SIZE={world} does not match --gpus {a.gpus}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cpu = a.selftest_cpu
    backend = "gloo" if (cpu or a.share_gpu) else a.backend
    if cpu:
        dev = torch.device("cpu")
    else:
        if a.share_gpu:
            local = 0
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from gct_plus_amd import synthetic
    mtype = a.model_type
    inner, opt, fwd_loss = (selftest_workload if cpu else hip_workload)(a, dev, world, rank)
    if not cpu:
        from gct_plus_amd import ops
        from gct_plus_amd.Train.trainer1 import warmup_lr
        main_mode = ops.gemm_get_mode()
    else:
        ops, main_mode = None, None
        warmup_lr = lambda i, d, w: 1e-3                                          # noqa: E731

    # synthetic MOSES-shaped pool, sharded like DistributedSampler, staged in HBM up front
    S = 20 if (a.tiny or cpu) else 80

    def make_pool(fixed_len, batch=None, mt=None):
        n_pool, batch = 4, batch or a.batch
        ds = synthetic.make_dataset(batch * n_pool * world, S, mt or mtype, seed=0, fixed_len=fixed_len)
        idx = synthetic.shard_indices(ds["src"].size(0), world, rank, epoch=0, seed=0, shuffle=False)
        shard = {k: v[idx] for k, v in ds.items()}
        return [{k: v.to(dev) for k, v in b.items()} for b in synthetic.batches(shard, batch)]

    pool = make_pool(a.fixed_len)
    state = {"pool": pool}

    state.update(fwd_loss=fwd_loss, opt=opt, dp_model=getattr(fwd_loss, "model", inner))

    host = {"s": 0.0, "n": 0}

    def step(i):
        t_h = time.perf_counter()
        try:
            return _step(i)
        finally:
            host["s"] += time.perf_counter() - t_h
            host["n"] += 1

    def _step(i):
        batch = state["pool"][i % len(state["pool"])]
        loss = state["fwd_loss"](batch)
        ahead = getattr(state["fwd_loss"], "prefetch", None)
        if ahead is not None:
            ahead(state["pool"][(i + 1) % len(state["pool"])])
        o = state["opt"]
        o.zero_grad(set_to_none=True)
        loss.backward()
        o.step()
        lr = warmup_lr(i + 1, 512, 8000)
        for g in o.param_groups:
            g["lr"] = lr
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        if not cpu:
            torch.cuda.synchronize()

    def reduce_max(dt):
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    def region(first, k):
        """EXACTLY k steps bracketed by barrier + device synchronise on both sides; max over ranks."""
        fence()
        t0 = time.perf_counter()
        last = None
        for i in range(first, first + k):
            last = step(i)
        fence()
        return reduce_max(time.perf_counter() - t0), last

    for i in range(a.warmup):
        step(i)
    fence()
    dpm = state["dp_model"] if world > 1 and hasattr(state.get("dp_model"), "diag_summary") else None
    if dpm is not None:
        dpm.diag = True
        dpm.diag_reset()
    host.update(s=0.0, n=0)
    if ops is not None:
        ops.HOST_BLOCKED_S[0] = 0.0
    if not cpu and not a.no_kernel_timing:
        ops.PROFILE = {}
        # only the roofline kernel is bracketed by events inside the timed region (all three GEMM
        # kinds cost ~2 % of the step in event overhead); --all-kernel-timing restores the rest
        ops.PROFILE_KINDS = None if a.all_kernel_timing else {"gemm_fwd"}
    dt, last = region(a.warmup, a.steps)
    host_main = {"ms_per_step": round(host["s"] / max(host["n"], 1) * 1e3, 3),
                 "blocked_in_readback_ms_per_step": round(ops.HOST_BLOCKED_S[0] / max(host["n"], 1) * 1e3, 3) if ops else None,
                 "note": "wall time this rank's Python spent inside the timed steps (ctypes launches, autograd, allocator) and "
                         "the part of it spent waiting in the step's one device->host read-back; a step is host-bound when "
                         "ms_per_step minus the wait approaches the step time"}
    prof = None
    if not cpu:
        prof, ops.PROFILE = ops.PROFILE, None
    final_loss = float(last.item()) / a.batch
    nxt = a.warmup + a.steps
    multi = None
    if world > 1:
        multi = multi_rank_report(a, dpm, inner, host, ops, dev, world, rank, cpu, step, region, nxt)
        nxt += 8

    # the same step on batches WITHOUT padding (every sample 80 tokens): none of the data-dependent
    # shortcuts (zero-gradient query tiles, zero rows of the decoder backward) can apply
    fixed = None
    if not cpu and not a.no_fixed_len_leg and not a.fixed_len:
        state["pool"] = make_pool(True)
        k2 = max(3, a.steps // 3)
        for i in range(2):
            step(nxt + i)
        dt2, _ = region(nxt + 2, k2)
        nxt += 2 + k2
        state["pool"] = pool
        fixed = {"steps": k2, "ms_per_step": round(dt2 / k2 * 1e3, 3), "value": round(a.batch * world * k2 / dt2, 1),
                 "note": "every sample 80 tokens (no padding): worst case of SURVEY.md 8(d)"}

    # the same step on the headline batches with EVERY decoder row computed in the forward pass (the reference's
    # forward: logits for padded target positions too, which its ignore_index loss then drops)
    dense = None
    if not cpu and not a.no_fixed_len_leg and not a.dense_decoder and not a.fixed_len:
        a.dense_decoder = True                    # fwd_loss reads the switch at every call
        k2 = max(3, a.steps // 3)
        for i in range(2):
            step(nxt + i)
        dt2, _ = region(nxt + 2, k2)
        nxt += 2 + k2
        a.dense_decoder = False
        dense = {"steps": k2, "ms_per_step": round(dt2 / k2 * 1e3, 3), "value": round(a.batch * world * k2 / dt2, 1),
                 "note": "forward pass over every decoder row (model.forward without loss_rows); same loss, same gradients"}

    # the same step with the GEMMs on the fp32 MFMA pipe (v_mfma_f32_32x32x2_f32), for reference: a short
    # further timed region (every rank runs it, so the collectives stay matched)
    alt = None
    if not cpu and main_mode == ops.GEMM_BF16X6 and not a.no_alt_mode:
        ops.gemm_set_mode(ops.GEMM_F32)
        k2 = max(3, a.steps // 3)
        for i in range(2):
            step(nxt + i)
        dt2, _ = region(nxt + 2, k2)
        nxt += 2 + k2
        ops.gemm_set_mode(main_mode)
        alt = {"gemm_arithmetic": "fp32 MFMA (v_mfma_f32_32x32x2_f32)", "steps": k2,
               "ms_per_step": round(dt2 / k2 * 1e3, 3), "value": round(a.batch * world * k2 / dt2, 1)}

    # Train.trainer1.run_epoch itself (per-step loss read-back, log line, LR write -- the reference's loop,
    # Train/trainer1.py:80-151) over staged batches at this run's batch size and at the reference scripts' 128 and 64
    # (Bashscript/train/train_vaetf.sh:10-18, train_scavaetf.sh:10-20), against the bare step of this file on the same
    # batches; and once behind the tokenizer + collate loader on synthetic SMILES strings
    tloop = None
    side_legs = world == 1 or a.all_legs
    if not cpu and not a.no_trainer_loop and not a.tiny and side_legs:
        tloop = trainer_loop_legs(a, inner, opt, state, step, region, fence, reduce_max, make_pool, dev, world, rank, nxt)
        nxt += 4000
        state["pool"] = pool

    # the other model types (BASELINE configs[2] = pvaetf, the 1-GPU point of configs[3] = scavaetf, pscavaetf): short legs
    # on MOSES-like batches and on unpadded ones, same step, same pool logic
    others = None
    if not cpu and not a.no_model_types and not a.tiny and side_legs:
        others = {}
        for mt in ("vaetf", "pvaetf", "scavaetf", "pscavaetf"):
            if mt == mtype:
                continue
            m2, o2, f2 = hip_workload(a, dev, world, rank, mtype=mt)
            state.update(fwd_loss=f2, opt=o2, pool=make_pool(False, mt=mt))
            for i in range(2):
                step(nxt + i)
            dtm, _ = region(nxt + 2, 7)
            state["pool"] = make_pool(True, mt=mt)
            for i in range(2):
                step(nxt + 9 + i)
            dtf, _ = region(nxt + 11, 5)
            nxt += 16
            others[mt] = {"steps": 7, "ms_per_step": round(dtm / 7 * 1e3, 3), "value": round(a.batch * world * 7 / dtm, 1),
                          "fixed_len_80": {"steps": 5, "ms_per_step": round(dtf / 5 * 1e3, 3),
                                           "value": round(a.batch * world * 5 / dtf, 1),
                                           "step_tflops_executed": round(FLOP_PER_SMILES_STEP[mt] * a.batch * world * 5 / dtf / 1e12, 2)}}
            del m2, o2, f2
            state.update(fwd_loss=fwd_loss, opt=opt, pool=pool)
            torch.cuda.empty_cache()
        others["note"] = (f"batch {a.batch}/GPU, dropout {a.dropout}, 7 timed steps after 2 warm-up on MOSES-like lengths, 5 after 2 "
                          "on unpadded batches; pvaetf = BASELINE configs[2] (encoder length 83, cross-attention keys 86), "
                          "scavaetf = the per-GPU point of configs[3]")

    dec = None
    if not cpu and not a.no_decode and not a.tiny:
        dec = decode_block(inner, a, dev, world, fence, reduce_max)

    if rank == 0:
        ms = dt / a.steps * 1e3
        value = a.batch * world * a.steps / dt
        launcher = os.environ.get("GCT_BENCH_LAUNCHER") or ("torch.distributed.run" if world > 1 else "single process")
        if cpu:
            out = {"metric": "bench.py launcher / step-loop self-test (CPU stand-in model, NOT a benchmark)",
                   "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": a.steps,
                   "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
                   "vs_baseline": None, "dtype": "f32", "data": "synthetic (self-test)",
                   "config": {"workload": f"self-test stand-in, batch {a.batch}/rank", "global_batch": a.batch * world,
                              "seq_len": S, "parallelism": f"dp{world}"},
                   "ranks": world, "backend": backend, "launcher": launcher,
                   "final_loss_per_sample": round(final_loss, 4)}
            if multi is not None:
                out["multi_rank"] = multi
            print(json.dumps(out), flush=True)
        else:
            roof = roofline(prof, dt, a.steps)
            out = {
                "metric": f"SMILES/sec training step ({mtype}, seq_len=80, d_model=512)",
                "value": round(value, 1), "unit": "SMILES/s", "n_gpus": world, "steps": a.steps,
                "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "gemm_arithmetic": ("bf16x6: fp32 operands split exactly into 3 bf16 pieces, 6 partial products, fp32 "
                                    "accumulate (error vs fp64 <= the fp32 fma chain's; GCT_GEMM_MODE=f32 selects "
                                    "v_mfma_f32_32x32x2_f32)") if ops.gemm_get_mode() == ops.GEMM_BF16X6
                                   else "fp32 MFMA (v_mfma_f32_32x32x2_f32)",
                "config": {"workload": f"{mtype} training step: 6+6 layers d_model=512 h=8 d_ff=2048 "
                                       f"latent=128, batch {a.batch}/GPU, seq_len=80 (T=81), dropout "
                                       f"{a.dropout}, CE+KL loss, fused Adam, fp32 (BASELINE configs["
                                       f"{3 if mtype == 'scavaetf' and world > 1 else 1}])"
                                       + (", TINY dims (launcher test)" if a.tiny else "")
                                       + (", every sample 80 tokens" if a.fixed_len else
                                          ", MOSES-like lengths N(35,8) padded to 80"),
                           "global_batch": a.batch * world, "seq_len": 80,
                           "parallelism": f"dp{world}"},
                "ranks": world, "backend": "rccl" if backend == "nccl" else "gloo (host-staged, test rig)",
                "launcher": launcher,
                "final_loss_per_sample": round(final_loss, 4),
                "roofline": roof,
                "host": host_main,
            }
            try:                                       # which box is this: the matrix rate it sustains (diagnostics library)
                from gct_plus_amd import graphdiag
                out["box"] = graphdiag.mfma_probe()
            except Exception as exc:                   # noqa: BLE001 -- calibration only, never a reason to lose the line
                out["box"] = {"error": repr(exc)}
            if not a.dense_decoder:
                out["rows_not_computed"] = (
                    "as Train/trainer1.run_epoch runs it: decoder rows whose target is <pad> (ignore_index in the loss) "
                    "and K/V of padded memory / source rows are not computed; loss, every gradient and the updated "
                    "parameters equal the dense step's (tests/test_model_gpu.py::test_full_size_batch_512_skip_ignored_"
                    "vs_oracle); `every_decoder_row` and `fixed_len_80` are the same step without the shortcut / on "
                    "batches where none applies")
            if dense is not None:
                out["every_decoder_row"] = dense
            if fixed is not None:
                out["fixed_len_80"] = fixed
                # 3 x forward flops per SMILES x the rate of the leg in which every row is computed: on
                # MOSES-length batches the step skips the rows that cannot reach the loss, so no such figure is quoted there
                out["step_tflops_executed_fixed_len_80"] = round(FLOP_PER_SMILES_STEP[mtype] * fixed["value"] / 1e12, 2)
            if alt is not None:
                out["same_step_fp32_mfma_gemms"] = alt
            if multi is not None:
                out["multi_rank"] = multi
            if others is not None:
                out["model_types"] = others
            if tloop is not None:
                out["trainer_loop"] = tloop
            if dec is not None:
                out["decode"] = dec
            if world == 1 and not a.no_cpu_baseline and not a.tiny:
                out["cpu_baseline"] = cpu_baseline()
            print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def roofline(prof, dt, steps=0):
    """`roofline` object for the dominant kernel from the live HIP-event brackets of ops._Timed."""
    if not prof:
        return None
    kern = {}
    klaunch = {k.split(":", 1)[1]: v for k, v in prof.items() if k.startswith("_x6_kernel_launches:")}
    for kind, recs in prof.items():
        if kind.startswith("_"):
            continue
        tsum = sum(e0.elapsed_time(e1) for _, e0, e1 in recs) * 1e-3
        fsum = sum(f for f, _, _ in recs)
        kern[kind] = {"launches": len(recs), "avg_us": round(tsum / len(recs) * 1e6, 1),
                      "tflops": round(fsum / tsum / 1e12, 2), "share_of_step": round(tsum / dt, 3)}
        if klaunch.get(kind):        # GEMM calls vs gemm_x6_kernel launches (tail-balanced calls make two)
            kern[kind]["kernel_launches"] = klaunch[kind]
            kern[kind]["avg_kernel_us"] = round(tsum / klaunch[kind] * 1e6, 1)
    x6 = "gemm_fwd[x6]" in kern
    dom = "gemm_fwd[x6]" if x6 else "gemm_fwd"
    if dom not in kern:
        return None
    # HBM bytes per launch come from separate rocprofv3 --pmc passes (they cannot be collected by this run);
    # the newest committed summary is quoted together with the commit it was measured at
    traffic, tsrc = None, None
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm_x6_traffic.json" if x6 else "r*_gemm_fwd_traffic.json")))
    if cands:
        rec = json.load(open(cands[-1]))
        traffic = rec.get("hbm_bytes_per_launch")
        tsrc = {"file": os.path.relpath(cands[-1], ROOT), "commit": rec.get("commit", "unrecorded (round 1)")}
    # algorithmic bytes of THIS run's forward-GEMM calls (operands read once, results written once) against what the
    # counters saw for the forward GEMM kernels of a step (every launch of the 128 x 256-tile and the 64 x 128-tile
    # forward kernels, from the same committed summary): > 1 = re-reads (operand panels fetched by more than one XCD's L2)
    alg = prof.get("_bytes:" + dom)
    ncalls = len(prof[dom]) if dom in prof else 0
    nl = klaunch.get(dom) or ncalls
    alg_per_launch = round(alg / nl) if (alg and nl) else None
    ratio, per_step = None, None
    if cands and alg and ncalls and steps:
        allk = rec.get("all", {})
        psteps = (allk.get("adam_kernel") or {}).get("launches") or rec.get("steps_profiled")
        fw = [v for k, v in allk.items() if k.startswith("gemm_x6_kernel<0>") or k.startswith("gemm_x6s_kernel<0>")
              or (not x6 and k.startswith("gemm_f32_fast_kernel"))]
        if psteps and fw:
            per_step = {"counters_bytes": round(sum(v["launches"] * v["hbm_bytes_per_launch"] for v in fw) / psteps),
                        "algorithmic_bytes": round(alg / steps)}
            ratio = round(per_step["counters_bytes"] / per_step["algorithmic_bytes"], 3)
    if x6:
        # bf16 MFMA pipe, six bf16 partial products per fp32 product: the fp32-equivalent
        # ceiling of the kernel is the dense bf16 peak / 6
        peak = PEAK_BF16_MFMA_TFLOPS / 6.0
        return {"kernel": "gemm_x6_kernel<FWD> (nn.Linear forward, exact 3-way bf16 split of both fp32 "
                          "operands, 6 partial products on v_mfma_f32_16x16x32_bf16, fp32 accumulate)",
                "bound": "mfma", "achieved": kern[dom]["tflops"], "peak": round(peak, 1),
                "unit": "TFLOP/s", "frac": round(kern[dom]["tflops"] / peak, 4),
                "mfma_utilisation": round(6 * kern[dom]["tflops"] / PEAK_BF16_MFMA_TFLOPS, 4),
                "traffic": traffic, "traffic_algorithmic": alg_per_launch, "traffic_ratio": ratio,
                "traffic_forward_gemms_per_step": per_step, "traffic_source": tsrc,
                "traffic_source_commit": tsrc["commit"] if tsrc else None,
                "avg_launch_us": kern[dom].get("avg_kernel_us", kern[dom]["avg_us"]),
                "launches": kern[dom].get("kernel_launches", kern[dom]["launches"]),
                "gemm_calls": kern[dom]["launches"], "avg_call_us": kern[dom]["avg_us"],
                "note": "avg_launch_us = time of all forward GEMM calls / bf16x6 forward kernel launches (a "
                        "tail-balanced call launches the 128x256-tile kernel for the full rounds and the 64x128-tile "
                        "kernel -- or a K-split launch plus a small fix-up kernel -- for the tail rows, all "
                        "inside the bracket); achieved/peak in fp32-equivalent (algorithmic) FLOP/s, peak = dense "
                        f"bf16 {PEAK_BF16_MFMA_TFLOPS} / 6 partial products, so frac = MFMA utilisation; on the pipe "
                        f"itself: {round(6 * kern[dom]['tflops'], 1)} of {PEAK_BF16_MFMA_TFLOPS} bf16 TFLOP/s; "
                        f"the fp32 MFMA pipe these GEMMs ran on before peaks at {PEAK_F32_MFMA_TFLOPS}",
                "kernels": kern}
    return {"kernel": "gemm_f32_fast_kernel<true,true> (nn.Linear forward, every shape of the step)",
            "bound": "mfma", "achieved": kern[dom]["tflops"], "peak": PEAK_F32_MFMA_TFLOPS,
            "unit": "TFLOP/s", "frac": round(kern[dom]["tflops"] / PEAK_F32_MFMA_TFLOPS, 4),
            "traffic": traffic, "traffic_algorithmic": alg_per_launch, "traffic_ratio": ratio,
            "traffic_forward_gemms_per_step": per_step, "traffic_source": tsrc, "traffic_source_commit": tsrc["commit"] if tsrc else None,
            "avg_launch_us": kern[dom]["avg_us"], "launches": kern[dom]["launches"], "kernels": kern}


def main():
    argv = sys.argv[1:]
    a = parse_args(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_workers(a, argv))          # the parent never imports torch.cuda / touches a GPU
    worker(a)


if __name__ == "__main__":
    main()
