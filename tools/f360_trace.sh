#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of the Frame360 chain (tools/prof_frame360.py) at one size, no counters.
# usage: bash tools/f360_trace.sh <tag> [width]      (writes gpurun_out/<tag>/f360_<width>/kernel_stats.csv)
TAG=${1:-r06}
W=${2:-4096}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG/f360_$W
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/prof_frame360.py $W 0.03 40 0 > $OUT/trace.log 2>&1 || exit 1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -n 1) $OUT/kernel_stats.csv
cd $R
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "f360" in n:
        tot += float(r["AverageNs"])
        print("%-40s %8.2f us" % (n.split("(")[0].replace("void ", "").replace("f360::", "")[:40], float(r["AverageNs"]) / 1e3))
print("chain: %.1f us" % (tot / 1e3))
PY
