#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace of the Frame360 stage pipeline; usage: bash tools/prof_frame360.sh [width [angular_threshold [min_inliers [frame 0|1 [tag [colour 0|1]]]]]]
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=${1:-2048}
TAG=${5:-f360_$W}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/prof_frame360.py $W ${2:-0.03} ${3:-40} ${4:-0} ${6:-0} > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cut -d, -f1-5 $f | head -24
