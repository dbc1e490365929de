"""Per-kernel median durations from a rocprofv3 --kernel-trace CSV.
python tools/trace_summary.py <dir> [substring] [--phases <kernel substring that starts a new phase>]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
sub = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
starter = sys.argv[sys.argv.index("--phases") + 1] if "--phases" in sys.argv else None
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
phases, prev = [{}], None
for r in rows:
    n = r["Kernel_Name"]
    if sub not in n:
        continue
    short = n.replace("(anonymous namespace)::", "").replace("void ", "")[:44]
    if starter and starter in n and prev is not None and starter not in prev:
        phases.append({})
    phases[-1].setdefault(short, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    prev = n
for i, ph in enumerate(phases):
    if starter:
        print("phase", i)
    for n, ds in ph.items():
        ds = sorted(ds)
        print("  %-46s %5d  median %8.1f us" % (n, len(ds), ds[len(ds) // 2] / 1000))
