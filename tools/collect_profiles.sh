#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel trace of the same command + PMC passes.
# usage: bash tools/collect_profiles.sh <tag>
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-4k --no-native-multi --no-rotating --no-sequence --no-live-traffic > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/prof_eval.py 2048 1024 10 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/tools/prof_eval.py 2048 1024 10 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- python3 $R/tools/prof_eval.py 2048 1024 10 > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_tcc -- python3 $R/tools/prof_eval.py 2048 1024 10 > $OUT/pmc_tcc.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_4k -- python3 $R/tools/prof_eval.py 4096 2048 10 > $OUT/pmc_fetch_4k.log 2>&1
# the HBM-fed legs of the bench line, one regime per profiler run (tools/prof_hbm_legs.py): kernel trace + stats, and FETCH_SIZE / WRITE_SIZE
# on the rotating leg (each counter set in its own run, never together with a trace)
for leg in rotating 4k batch; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$leg -- python3 $R/tools/prof_hbm_legs.py $leg 10 > $OUT/trace_$leg.log 2>&1
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_rotating -- python3 $R/tools/prof_hbm_legs.py rotating 4 > $OUT/pmc_fetch_rotating.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_rotating -- python3 $R/tools/prof_hbm_legs.py rotating 4 > $OUT/pmc_write_rotating.log 2>&1
# the 16-slot batch pass k_eval_b (configs[3]'s kernel): bytes and issue counters, each set in its own run
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_batch -- python3 $R/tools/prof_hbm_legs.py batch 4 > $OUT/pmc_fetch_batch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_batch -- python3 $R/tools/prof_hbm_legs.py batch 4 > $OUT/pmc_write_batch.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq_batch -- python3 $R/tools/prof_hbm_legs.py batch 4 > $OUT/pmc_sq_batch.log 2>&1
cd $R
cat $OUT/bench.json
python3 tools/trace_gaps.py $OUT/trace
for d in pmc_fetch pmc_write pmc_sq pmc_tcc pmc_fetch_4k pmc_fetch_rotating pmc_write_rotating pmc_fetch_batch pmc_write_batch pmc_sq_batch; do echo "== $d"; cat $OUT/$d.log | grep "avg us"; python3 tools/pmc_summary.py $OUT/$d; done
for leg in rotating 4k batch; do echo "== trace_$leg"; cat $OUT/trace_$leg.log | grep "HIP events"; python3 tools/trace_gaps.py $OUT/trace_$leg | grep k_eval; cp $(ls $OUT/trace_$leg/*/*kernel_stats.csv | tail -1) $OUT/hbm_${leg}_kernel_stats.csv; done
