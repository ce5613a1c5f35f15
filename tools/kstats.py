"""Average kernel durations (us) of rocprofv3 --kernel-trace --stats runs kept under gpurun_out/<tag>/: python tools/kstats.py TAG [TAG ...] [filter]"""
import csv, glob, os, sys
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
tags = [a for a in sys.argv[1:] if os.path.isdir(os.path.join(root, a))]
flt = [a for a in sys.argv[1:] if a not in tags]
cols = []
for t in tags:
    f = sorted(glob.glob(os.path.join(root, t, "trace", "*", "*kernel_stats.csv")))[-1]
    cols.append({r["Name"].split("(")[0].replace("void ", ""): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f))})
names = sorted(set().union(*cols), key=lambda k: -max(c.get(k, 0) for c in cols))
print("%-40s" % "kernel" + "".join("%12s" % t[-12:] for t in tags))
for k in names:
    if flt and not any(x in k for x in flt): continue
    print("%-40s" % k[:40] + "".join("%12.2f" % c.get(k, float("nan")) for c in cols))
print("%-40s" % "sum" + "".join("%12.1f" % sum(c.values()) for c in cols))
