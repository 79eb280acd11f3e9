export TMPDIR=/tmp
mkdir -p gpurun_out
( ./tools/scratch/metric_phase.bin 100 256 0 1e-3 1 10 1 ) > gpurun_out/fast_phase_traj.txt 2>&1
grep -v "^    wave" gpurun_out/fast_phase_traj.txt | tail -12 | cut -c1-200
( ./tools/scratch/metric_phase.bin 100 256 0 1e-3 1 ) > gpurun_out/fast_phase.txt 2>&1
HTA_RMHMC_FUSED=0 timeout 600 python tools/bench_with_lib.py hamiltorch_amd/libhamiltorch_amd.so --workload cfg3 --traj 20 --steps 5 --warmup 1 > gpurun_out/fast_bench.txt 2>&1; tail -1 gpurun_out/fast_bench.txt | cut -c1-300
timeout 1200 python -m pytest tests/test_gpu_rmhmc.py -x -q -m gpu ${TESTSEL} > gpurun_out/fast_tests.txt 2>&1; tail -3 gpurun_out/fast_tests.txt
