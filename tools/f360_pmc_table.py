"""One table per profiled Frame360 run (tools/collect_frame360.sh): per kernel the rocprofv3 average duration, the HBM-side bytes of a
launch -- (2 x FETCH_SIZE + WRITE_SIZE) KB, the gfx950 correction of MI355X_MICROARCH.md -- and what that is in TB/s, vector instructions
per launch and the share of the kernel's duration the chip's SIMDs would need to issue them (4 cycles each at 2.4 GHz over 1024 SIMDs).
    python tools/f360_pmc_table.py gpurun_out/<tag>/f360_<W> <W>"""
import csv, glob, os, sys
from collections import defaultdict

d, W = sys.argv[1], int(sys.argv[2])
n_px = W * (W // 2)


def short(name):
    return name.split("(")[0].replace("void ", "").replace("f360::", "")


def counters(sub):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


stats = {}
for r in csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))):
    stats[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
fetch, write, sq, sq2 = counters("pmc_fetch"), counters("pmc_write"), counters("pmc_sq"), counters("pmc_sq2")
frames = 3          # tools/prof_frame360.py runs the call three times
tot = 0.0
print("Frame360 chain at %d x %d (tools/prof_frame360.py %d 0.03 40 0): per kernel, averages over the profiled launches" % (W, W // 2, W))
print("%-34s %5s %9s %10s %8s %10s %8s %8s %8s" % ("kernel", "calls", "avg us", "HBM MB", "TB/s", "VALU inst", "issue%", "LDS inst", "wait%"))
for name, (calls, us) in sorted(stats.items(), key=lambda kv: -kv[1][1] * kv[1][0]):
    if not name.startswith("k_f360") and not name.startswith("k_sphere"):
        continue
    tot += us * calls / frames
    fb = (2.0 * fetch.get(name, {}).get("FETCH_SIZE", float("nan")) + write.get(name, {}).get("WRITE_SIZE", float("nan"))) * 1024.0
    valu = sq.get(name, {}).get("SQ_INSTS_VALU", float("nan"))
    issue_us = valu / 1024.0 * 4.0 / 2.4e3
    waves = sq.get(name, {})
    wait = 100.0 * waves.get("SQ_WAIT_INST_ANY", float("nan")) / max(waves.get("SQ_WAVE_CYCLES", float("nan")), 1.0)
    lds = sq2.get(name, {}).get("SQ_INSTS_LDS", float("nan"))
    print("%-34s %5d %9.2f %10.1f %8.2f %10.3g %8.0f %8.3g %8.0f" % (name[:34], calls // frames, us, fb / 1e6, fb / (us * 1e-6) / 1e12, valu, 100.0 * issue_us / us, lds, wait))
bytes_alg = (14 + 24 + 16) * n_px
print("chain: %.1f us of kernels per frame; SURVEY 8d bytes (14 + 24 + 16 B/px) = %.1f MB -> %.1f us at 8 TB/s -> %.3f of the HBM roof" % (tot, bytes_alg / 1e6, bytes_alg / 8e12 * 1e6, bytes_alg / 8e12 * 1e6 / tot))
