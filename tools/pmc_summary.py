"""Summarise a rocprofv3 --pmc csv: per kernel name, mean of each counter.  usage: pmc_summary.py <dir> [name-filter ...]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0][:60]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in acc.items():
    if not any(k in name for k in (sys.argv[2:] or ["k_eval", "k_solve"])): continue
    print(name)
    for c, v in sorted(cs.items()):
        print("   %-28s n=%3d mean=%.4g" % (c, len(v), sum(v) / len(v)))
