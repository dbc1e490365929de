"""Mean counter values per kernel from rocprofv3 --pmc counter_collection CSVs.
python tools/pmc_generic.py <dir> [kernel substring]"""
import csv, glob, sys, collections
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if sub in n:
            short = n.replace("(anonymous namespace)::", "").replace("void ", "")[:40]
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-40s n=%4d mean %.4g" % (c, len(v), sum(v) / len(v)))
