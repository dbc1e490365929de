"""norm_bwd with / without the fused dropout_bwd output vs the separate dropout_bwd pass (same process, same box)."""
import sys, time, torch
sys.path.insert(0, ".")
from gct_plus_amd import ops
dev = "cuda"
M, d = 40960, 512
x = torch.randn(M, d, device=dev); dy = torch.randn(M, d, device=dev); res = torch.randn(M, d, device=dev)
al = torch.ones(d, device=dev); be = torch.zeros(d, device=dev)
y, mean, rstd = ops.norm_fwd(x, al, be, 1e-6)
da, db = torch.empty(d, device=dev), torch.empty(d, device=dev)
out, gd = torch.empty(M, d, device=dev), torch.empty(M, d, device=dev)
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
a = timeit(lambda: ops.norm_bwd(dy, x, al, mean, rstd, da, db, dres=res, out=out))
b = timeit(lambda: ops.norm_bwd(dy, x, al, mean, rstd, da, db, dres=res, out=out, drop=(gd, 0.1, 7, 3)))
c = timeit(lambda: ops.dropout_bwd(out, 0.1, 7, 3))
ref = ops.dropout_bwd(out, 0.1, 7, 3)
print(f"norm_bwd {a:.1f} us   norm_bwd+drop {b:.1f} us   separate dropout_bwd {c:.1f} us   fused == separate: {bool(torch.equal(ref, gd))}")
