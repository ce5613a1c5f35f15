"""One line per profiled Frame360 run under gpurun_out/<tag>/ (tools/prof_frame360.sh): kernel time per frame, launches per frame, per-kernel averages.
    python tools/f360_summary.py TAG [TAG ...]"""
import csv, glob, os, sys
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
for tag in sys.argv[1:]:
    f = sorted(glob.glob(os.path.join(root, tag, "trace", "*", "*kernel_stats.csv")), key=os.path.getmtime)[-1]
    tot, n, out = 0.0, 0.0, []
    for r in csv.DictReader(open(f)):
        nm = r["Name"].split("(")[0].replace("void ", "").replace("f360::", "").replace("k_f360_", "")
        calls, avg = int(r["Calls"]), float(r["AverageNs"]) / 1e3
        tot += avg * calls / 3
        n += calls / 3
        out.append("%s %.1f" % (nm, avg) + ("x%d" % (calls // 3) if calls > 3 else ""))
    print(tag, "total %.1f us per frame, %.0f launches |" % (tot, n), "; ".join(out))
