"""Prints the newest rocprofv3 kernel-stats summary that tools/prof_frame360.sh left under gpurun_out/f360_<W>/.  python tools/f360_stats.py [W ...]"""
import csv, glob, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for w in (sys.argv[1:] or ["2048", "4096"]):
    fs = glob.glob(os.path.join(root, "gpurun_out", f"f360_{w}", "trace", "*", "*kernel_stats.csv"))
    if not fs:
        continue
    f = max(fs, key=os.path.getmtime)
    print(w, os.path.relpath(f, root))
    tot = 0.0
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0]
        print("  %-32s calls %3s avg %8.1f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3))
        if "rocclr" not in n:
            tot += float(r["AverageNs"]) / 1e3
    print("  sum of our kernels per frame: %.1f us" % tot)
