#!/bin/bash
# Runs on the GPU box: kernel time of k_f360_normals_sweep for several rows-per-wave settings.  usage: bash tools/normals_seg_sweep.sh [width] [segs...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=${1:-2048}; shift
cd /tmp; export TMPDIR=/tmp
for SEG in ${@:-16 20 27 38}; do
  OUT=$R/gpurun_out/seg_${W}_$SEG
  mkdir -p $OUT
  RGBD360_SWEEP_SEG=$SEG rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/prof_frame360.py $W 0.03 40 0 > /dev/null 2>&1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "seg $SEG: $(grep normals_sweep $f | sed 's/(.*)//' | cut -d, -f1-4)"
done
