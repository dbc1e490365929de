"""Host-side time of one training step (Python + ctypes + launches): the full layer count (6+6) at a width where the
GPU work is negligible, so the wall time per step is the host's.  python tools/host_floor.py"""
import sys, time, torch
sys.path.insert(0, ".")
from gct_plus_amd import synthetic
from gct_plus_amd.Model import forward_propagation, model_dict
from gct_plus_amd.Train.trainer1 import loss_function
from gct_plus_amd.optim import FusedAdam
dev = torch.device("cuda", 0)
mtype = "vaetf"
vs, vt = synthetic.vocab_sizes(mtype)
torch.manual_seed(1)
model = model_dict[mtype](vs, vt, dropout=0.1, nconds=0, use_cond2dec=False, use_cond2lat=True,
                          N=6, d_model=64, dff=128, h=4, latent_dim=16).to(dev).train()
opt = FusedAdam(model.parameters(), lr=1e-4, betas=(0.9, 0.98), eps=1e-9, model=model)
ds = synthetic.make_dataset(64 * 4, 80, mtype, seed=0, fixed_len=False)
pool = [{k: v.to(dev) for k, v in b.items()} for b in synthetic.batches(ds, 64)]
def step(i):
    batch = pool[i % len(pool)]
    prop, mol, mu, lv, _ = forward_propagation[mtype](model, batch, synthetic.PAD_ID, False)
    ys = batch["trg"][:, 1:].contiguous().view(-1)
    loss, _, _, _ = loss_function(0.04, prop, mol, None, ys, mu, lv, False, synthetic.PAD_ID)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
for i in range(5):
    step(i)
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for i in range(n):
    step(i)
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"6+6 layers at d_model=64, B=64: {1e3*(t1-t0)/n:.1f} ms/step wall (host-bound)")
