"""Copies what a tools/collect_profiles.sh run left under gpurun_out/<tag>/ into profiles/ under the round's name, and rebuilds
profiles/traffic_latest.json (the PMC-derived fabric-side bytes per launch that bench.py quotes as roofline.traffic).
    python tools/publish_profiles.py <tag> <round-name>         e.g.  python tools/publish_profiles.py r04b r04"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")


def last(pattern):
    f = sorted(glob.glob(os.path.join(src, pattern), recursive=True))
    return f[-1] if f else None


shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, rnd + "_bench.json"))
shutil.copy(last("trace/**/*kernel_stats.csv"), os.path.join(dst, rnd + "_bench_kernel_stats.csv"))
for leg in ("rotating", "4k", "batch"):
    f = last("trace_%s/**/*kernel_stats.csv" % leg)
    if f:
        shutil.copy(f, os.path.join(dst, "%s_hbm_%s_kernel_stats.csv" % (rnd, leg)))
log = os.path.join(ROOT, "gpurun_out", "collect_%s.log" % tag)
if os.path.exists(log):
    lines = open(log).read().split("\n")
    body = [l for l in lines if not l.startswith("{\"metric\"")]
    open(os.path.join(dst, rnd + "_pmc_and_trace_summary.txt"), "w").write(
        "# tools/collect_profiles.sh %s on one MI355X box: per-kernel durations and gaps of `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 "
        "--no-cpu-baseline --no-4k --no-native-multi --no-rotating --no-sequence` (tools/trace_gaps.py), the PMC passes of tools/prof_eval.py and tools/prof_hbm_legs.py "
        "(each counter set in its own run; FETCH_SIZE / WRITE_SIZE in KB, bytes = 2 x FETCH_SIZE + WRITE_SIZE for 16 B/lane streams, MI355X_MICROARCH.md), and the kernel traces "
        "of the HBM-fed legs (rotating over 8 copies of the pair, 4096 x 2048, 16-slot batch) beside the HIP-event figures of the same runs.\n" % tag + "\n".join(body))


def pmc_mean(d, counter):
    acc = {}
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def pick(means, needle):
    for k, v in means.items():
        if needle in k:
            return v
    return None


out = {"_how": "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024; FETCH_SIZE / WRITE_SIZE = mean over the dispatches of rocprofv3 --pmc (a separate pass per counter) in "
               "tools/collect_profiles.sh; the x2 is the gfx950 correction of MI355X_MICROARCH.md (HBM section) for 16 B/lane streams; the 12-byte gathers are an uncalibrated "
               "width, so treat the figure as +-10%. At 2048x1024 the working set is Infinity-Cache resident: these are fabric-side requests, not necessarily HBM array reads. "
               "Kernel: k_eval_fs (the product launch: solve prologue + pass). 4096x2048 runs the recompute form of the source stream (8 B per source pixel).",
       "collected": "round %s, tools/collect_profiles.sh %s -> tools/publish_profiles.py" % (rnd, tag)}
fetch, write, fetch4 = pmc_mean("pmc_fetch", "FETCH_SIZE"), pmc_mean("pmc_write", "WRITE_SIZE"), pmc_mean("pmc_fetch_4k", "FETCH_SIZE")
for size, fm, npx in (("2048x1024", fetch, 2048 * 1024), ("4096x2048", fetch4, 4096 * 2048)):
    for m, name, bpp_rec, bpp_rc in ((0, "PHOTO_CONSISTENCY", 28, 20), (2, "PHOTO_DEPTH", 40, 32)):
        f, w = pick(fm, "k_eval_fs<%d" % m), pick(write, "k_eval_fs<%d" % m)
        if f is None or w is None:
            continue
        bpp = bpp_rc if npx >= 4 * 1024 * 1024 else bpp_rec
        b = (2 * f + w) * 1024
        out["%s_%s" % (size, name)] = {"kernel": "k_eval_fs<%d>" % m, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_per_launch": int(b),
                                        "algorithmic_bytes": bpp * npx, "bytes_per_pixel": bpp, "ratio": b / (bpp * npx)}
json.dump(out, open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
print(json.dumps({k: (v["ratio"] if isinstance(v, dict) else None) for k, v in out.items()}, indent=0))
