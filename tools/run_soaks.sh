#!/bin/bash
# Runs on the GPU box (via gpurun): the ten randomised soaks (tests/tools/*_soak.py) with a seed of the caller's, one after the other.
# usage: bash tools/run_soaks.sh <tag> <seed> [scale]      (writes gpurun_out/<tag>/soak_<name>_<seed>.txt)
TAG=${1:-r06}
SEED=${2:-1}
SC=${3:-1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
rc=0
for spec in fused:48 hull:18 engine:15 normals:10 pinhole_occ:40 planes:24 align:40 pinhole_align:40 sensor_stages:30 rig_dense:24; do
    name=${spec%%:*}; n=$(( ${spec##*:} * SC ))
    echo "== ${name}_soak $n trials, seed $SEED"
    timeout -k 10 900 python3 tests/tools/${name}_soak.py $n $SEED > $OUT/soak_${name}_${SEED}.txt 2>&1 || { rc=1; echo "FAILED: ${name}"; }
    tail -n 2 $OUT/soak_${name}_${SEED}.txt
done
exit $rc
