#!/bin/bash
# Runs on the GPU box: SQ counters of the Frame360 stage kernels (own pass, no trace domains).  usage: bash tools/pmc_frame360.sh [width] [kernel-name filter ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=${1:-2048}; shift
OUT=$R/gpurun_out/pmc_f360_$W
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq -- python3 $R/tools/prof_frame360.py $W 0.03 40 0 > $OUT/sq.log 2>&1
python3 $R/tools/pmc_summary.py $OUT/sq ${@:-normals sphere_cloud}
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq2 -- python3 $R/tools/prof_frame360.py $W 0.03 40 0 > $OUT/sq2.log 2>&1
python3 $R/tools/pmc_summary.py $OUT/sq2 ${@:-normals sphere_cloud}
