mkdir -p gpurun_out/r05t
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "occl" > gpurun_out/r05t/tests.log 2>&1; tail -3 gpurun_out/r05t/tests.log
for i in 1 2; do
echo base; RGBD360_LIB=$PWD/rgbd360_amd/lib/librgbd360_hip_base.so timeout -k 10 200 python tools/occ_perf.py 2>&1 | grep occlusion
echo new; timeout -k 10 200 python tools/occ_perf.py 2>&1 | grep occlusion
done
