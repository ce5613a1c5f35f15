#!/bin/bash
# kernel timeline of one rgbd360_frame_planes_dev call with a registered colour image: bash tools/f360_colour_timeline.sh [W]   (on the GPU box)
W=${1:-4096}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fpc
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/fpc -- python3 $R/tools/frame_planes_call_perf.py $W 0.03 0 1 > /dev/null 2>&1
python3 $R/tools/occ_timeline.py show /tmp/fpc
