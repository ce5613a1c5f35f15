#!/bin/bash
# Runs on the GPU box: kernel trace of the Frame360 chain with both moment kernels (RGBD360_F360_MOMENTS=0 per-region wave sums,
# 1 run-based segmented scan) in the one-region regime (0.03 rad) and the fragmented one (0.015 rad, rotated frame).
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in 0 1; do
  export RGBD360_F360_MOMENTS=$v
  bash $R/tools/prof_frame360.sh 2048 0.03 40 0 ab_mom_${v}_one > /dev/null 2>&1
  bash $R/tools/prof_frame360.sh 2048 0.015 40 1 ab_mom_${v}_many > /dev/null 2>&1
  bash $R/tools/prof_frame360.sh 4096 0.03 40 0 ab_mom_${v}_one4k > /dev/null 2>&1
  for t in one many one4k; do
    f=$(find $R/gpurun_out/ab_mom_${v}_$t/trace -name "*kernel_stats.csv" | head -1)
    echo "variant $v $t: $(grep planes $R/gpurun_out/ab_mom_${v}_$t/trace.log) $(grep k_f360_moments $f | cut -d, -f2-4 | tr '\n' ' ')"
  done
done
