"""Timeline of the last alignment in a rocprofv3 kernel trace: python tools/timeline.py <trace dir>"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# split into alignments at k_level_init
starts = [i for i, r in enumerate(rows) if "k_level_init" in r["Kernel_Name"]]
seg = rows[starts[-1]:]
t0 = int(seg[0]["Start_Timestamp"])
prev = t0
tot_busy = 0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void r360::", "").replace("r360::", "")[:22]
    print("%8.1f  +%6.1f gap  %6.1f us  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, name))
    tot_busy += e - s
    prev = e
print("span %.1f us, busy %.1f us, kernels %d" % ((prev - t0) / 1e3, tot_busy / 1e3, len(seg)))
