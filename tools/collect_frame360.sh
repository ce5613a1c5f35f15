#!/bin/bash
# Runs on the GPU box (via gpurun): the Frame360 chain (rows a13-a15) at one size -- rocprofv3 kernel trace + stats of tools/prof_frame360.py,
# then HBM bytes (FETCH_SIZE, WRITE_SIZE) and SQ issue counters per kernel, every counter set in its OWN run without a trace domain.
# usage: bash tools/collect_frame360.sh <tag> [width]      (writes gpurun_out/<tag>/f360_<width>/)
TAG=${1:-r06}
W=${2:-4096}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG/f360_$W
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
ARGS="$W 0.03 40 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/prof_frame360.py $ARGS > $OUT/trace.log 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/prof_frame360.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/tools/prof_frame360.py $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- python3 $R/tools/prof_frame360.py $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq2 -- python3 $R/tools/prof_frame360.py $ARGS > $OUT/pmc_sq2.log 2>&1
cd $R
python3 tools/f360_pmc_table.py $OUT $W > $OUT/summary.txt
cat $OUT/summary.txt
