"""Per-kernel breakdown of ONE late decode step from a rocprofv3 --kernel-trace CSV of tools/decode_bench.py.
python tools/decode_step_profile.py <dir>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [i for i, r in enumerate(rows) if "select_token" in r["Kernel_Name"]]
a, b = sel[-3], sel[-2]
seg = rows[a + 1:b + 1]
d = collections.defaultdict(list)
for r in seg:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:34]
    d[(n, int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
tot = 0
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print("%-36s grid %8d  x%3d  avg %7.1f us  sum %7.1f" % (k[0], k[1], len(v), sum(v) / len(v), sum(v)))
    tot += sum(v)
print("kernel sum us %.1f   span us %.1f" % (tot, (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1000))
