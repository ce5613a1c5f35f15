#!/bin/bash
# Runs on the GPU box (via gpurun): the Frame360 kernel trace (tools/f360_trace.sh) for several builds of the library
# (rgbd360_amd/lib/librgbd360_hip_<NAME>.so from tools/ab_libs.py build; 'default' = the product library), one line per kernel of interest.
# usage: bash tools/f360_ab.sh <tag> <width> <kernel-regex> NAME [NAME ...]
TAG=$1; W=$2; PAT=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
for NAME in "$@"; do
    if [ "$NAME" = default ]; then unset RGBD360_LIB; else export RGBD360_LIB=$R/rgbd360_amd/lib/librgbd360_hip_$NAME.so; fi
    bash $R/tools/f360_trace.sh ${TAG}_$NAME $W > $R/gpurun_out/${TAG}_${NAME}_$W.txt 2>&1 || { echo "$NAME: trace failed"; continue; }
    echo "$NAME ($W): $(grep -E "$PAT|chain" $R/gpurun_out/${TAG}_${NAME}_$W.txt | tr -s ' ' | tr '\n' ';')"
done
