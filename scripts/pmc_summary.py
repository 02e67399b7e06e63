"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter."""
import csv, sys, collections, glob
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if not any(s in k for s in sys.argv[2:] or [""]):
        continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s n=%3d mean %.4g" % (c, len(v), sum(v) / len(v)))
