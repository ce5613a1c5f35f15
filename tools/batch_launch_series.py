"""Per-launch durations of the 16-slot batch pass k_eval_b in a rocprofv3 kernel trace of tools/prof_hbm_legs.py batch (the trace_batch leg of
tools/collect_profiles.sh): the series in launch order with the idle gap in front of every launch, per call of rgbd360_forced_iters_batch.
    python tools/batch_launch_series.py gpurun_out/<tag>/trace_batch"""
import csv, glob, os, sys
from collections import defaultdict

f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
ks = defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void r360::", "")
    if "k_eval_b" in n:
        ks[n].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for n, v in sorted(ks.items()):
    v.sort()
    d = [(e - s) / 1e3 for s, e in v]
    print("%s: %d launches, min %.1f mean %.1f max %.1f us" % (n, len(d), min(d), sum(d) / len(d), max(d)))
    call, prev_end = [], None
    for (s, e), us in zip(v, d):
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        if prev_end is not None and gap > 1000.0:          # a new call of the entry (host work in between)
            print("   call: " + " ".join(call))
            call = []
        call.append("%.0f%s" % (us, "" if gap < 1.0 or prev_end is None else "(+%.0f)" % gap))
        prev_end = e
    print("   call: " + " ".join(call))
