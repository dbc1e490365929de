"""Robustness sweep over batch sizes and model types: three training steps (dropout 0.1, Adam) with the data-dependent
shortcuts on (decoder forward and backward over the live rows, visible-rows cross-attention K/V, tail balancing as the sizes fall) and
again with them off; the per-step losses must agree (the shortcuts are exact: same arithmetic on the live rows, same
dropout bits; only summation splits differ).  python tools/batch_sweep.py [--batches 24,64,...]"""
import argparse, sys, torch
sys.path.insert(0, ".")
from gct_plus_amd import engine, synthetic
from gct_plus_amd.Model import forward_propagation, model_dict
from gct_plus_amd.Train.trainer1 import loss_function
from gct_plus_amd.optim import FusedAdam
dev = torch.device("cuda", 0)
DROPOUT = 0.1
FIXED_LEN = False
DIMS = dict(N=6, d_model=512, dff=2048, h=8, latent_dim=128)


def run(mtype, B, compact, S=80):
    engine.COMPACT_BWD = engine.COMPACT_KV = engine.COMPACT_FWD = engine.COMPACT_ENC_KV = compact
    engine._SEED.update(base=None, ctr=0)          # same dropout / eps streams in both runs
    vs, vt = synthetic.vocab_sizes(mtype)
    nc = synthetic.n_conds(mtype)
    torch.manual_seed(1)
    model = model_dict[mtype](vs, vt, dropout=DROPOUT, nconds=nc, use_cond2dec=False, use_cond2lat=True,
                              **DIMS).to(dev).train()
    opt = FusedAdam(model.parameters(), lr=1e-4, betas=(0.9, 0.98), eps=1e-9, model=model)
    ds = synthetic.make_dataset(B * 3, S, mtype, seed=B, fixed_len=FIXED_LEN)
    losses = []
    torch.manual_seed(7)
    for batch in synthetic.batches(ds, B):
        batch = {k: v.to(dev) for k, v in batch.items()}
        prop, mol, mu, lv, _ = forward_propagation[mtype](model, batch, synthetic.PAD_ID, False, skip_ignored=True)
        ys = batch["trg"][:, 1:].contiguous().view(-1)
        ys_cond = batch["dconds"].unsqueeze(2).contiguous().view(-1, nc, 1) if nc else None
        loss, _, _, _ = loss_function(0.04, prop, mol, ys_cond, ys, mu, lv, False, synthetic.PAD_ID)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss.item()) / B)
    torch.cuda.synchronize()
    return losses



def compare(mtype, B, S=80):
    """max relative difference of the three per-sample losses, shortcuts on vs off"""
    keep = (engine.COMPACT_BWD, engine.COMPACT_KV, engine.COMPACT_FWD, engine.COMPACT_ENC_KV)
    try:
        on, off = run(mtype, B, True, S), run(mtype, B, False, S)
    finally:
        engine.COMPACT_BWD, engine.COMPACT_KV, engine.COMPACT_FWD, engine.COMPACT_ENC_KV = keep    # leave the module as found
    assert all(x == x for x in on + off), (mtype, B, on, off)
    return max(abs(x - y) / max(abs(y), 1e-9) for x, y in zip(on, off)), on, off


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="24,64,100,128,192,256,320,384,448,512")
    ap.add_argument("--model-types", default="vaetf,pscavaetf")
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--seqs", default="80", help="maximum SMILES lengths (source rows; the decoder sees one more)")
    ap.add_argument("--fixed-len", action="store_true", help="every sample at the maximum length (no padding)")
    ap.add_argument("--dims", default="", help="N,d_model,h,dff,latent (default 6,512,8,2048,128)")
    a = ap.parse_args()
    DROPOUT, FIXED_LEN = a.dropout, a.fixed_len
    if a.dims:
        n_, d_, h_, f_, l_ = (int(x) for x in a.dims.split(","))
        DIMS.update(N=n_, d_model=d_, h=h_, dff=f_, latent_dim=l_)
    bad = 0
    for mtype in a.model_types.split(","):
        for B, S in [(int(x), int(y)) for x in a.batches.split(",") for y in a.seqs.split(",")]:
            rel, on, off = compare(mtype, B, S)
            ok = rel < 2e-5
            bad += not ok
            print(f"{mtype:10s} B={B:4d} S={S:3d}  losses {['%.4f' % x for x in on]}  off {['%.4f' % x for x in off]}  max rel diff "
                  f"{rel:.1e}  {'ok' if ok else 'MISMATCH'}", flush=True)
    print("sweep", "FAILED" if bad else "ok")
    sys.exit(1 if bad else 0)
