#!/bin/bash
# Runs on the GPU box (via gpurun): rows per wave of the normal-map sweep (RGBD360_SWEEP_SEG, a debug-build knob) against the sweep's time.
# needs rgbd360_amd/lib/librgbd360_hip_dbg.so (python tools/ab_libs.py build dbg=-DRGBD360_DEBUG_KNOBS).  usage: bash tools/sweep_seg_ab.sh <tag> <width> SEG [SEG ...]
TAG=$1; W=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
export RGBD360_LIB=$R/rgbd360_amd/lib/librgbd360_hip_dbg.so
for SEG in "$@"; do
    if [ "$SEG" = auto ]; then unset RGBD360_SWEEP_SEG; else export RGBD360_SWEEP_SEG=$SEG; fi
    bash $R/tools/f360_trace.sh ${TAG}_seg$SEG $W > $R/gpurun_out/${TAG}_seg${SEG}_$W.txt 2>&1 || { echo "$SEG: trace failed"; continue; }
    echo "seg $SEG ($W): $(grep -E "normals_sweep|chain" $R/gpurun_out/${TAG}_seg${SEG}_$W.txt | tr -s ' ' | tr '\n' ';')"
done
