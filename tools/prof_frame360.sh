#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace of the Frame360 stage pipeline; usage: bash tools/prof_frame360.sh [width]
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=${1:-2048}
OUT=$R/gpurun_out/f360_$W
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/prof_frame360.py $W > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cut -d, -f1-5 $f | head -24
