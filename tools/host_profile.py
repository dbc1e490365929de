"""cProfile of the host side of the training step (same narrow 6+6-layer model as tools/host_floor.py):
python tools/host_profile.py [n_lines]   -> cumulative and own-time tables"""
import cProfile, pstats, sys, io, torch
sys.path.insert(0, ".")
sys.argv = sys.argv[:1] + sys.argv[1:]
import runpy
ns = runpy.run_path("tools/host_floor.py")
step = ns["step"]
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    step(i)
torch.cuda.synchronize()
pr.disable()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 45
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(n)
    print(s.getvalue()[:12000])
