"""Post-process a rocprofv3 kernel trace: per-kernel mean duration and mean idle gap before each kernel."""
import csv, glob, os, sys
from collections import defaultdict
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
dur, gap = defaultdict(list), defaultdict(list)
prev_end = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void r360::", "").replace("r360::", "")[:28]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[name].append(e - s)
    if prev_end is not None: gap[name].append(s - prev_end)
    prev_end = e
for n in dur:
    d = dur[n][len(dur[n])//4:]; g = gap[n][len(gap[n])//4:] or [0]
    print("%-30s n=%4d dur(us) mean=%.2f min=%.2f | gap-before(us) mean=%.2f min=%.2f" % (n, len(dur[n]), sum(d)/len(d)/1e3, min(d)/1e3, sum(g)/len(g)/1e3, min(g)/1e3))
