#!/bin/bash
# round 5, call b: tracebacks of the fp64 vglobal failures of r05a, the new kernels' parity tests, A/B of rmhmc_uvc2d
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r05b
timeout 300 python -m pytest tests/test_gpu_rmhmc.py -q -x --tb=short -k "test_metric_eval_vs_oracle and dtype1" > ${O}_metric.txt 2>&1; tail -30 ${O}_metric.txt
timeout 300 python -m pytest tests/test_gpu_rmhmc.py -q -x --tb=short -k "test_softabs_dmetric_vs_oracle and 100" > ${O}_dmetric.txt 2>&1; tail -15 ${O}_dmetric.txt | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_rmhmc.py -q --tb=short -k "uvc2d or beyond_the_round_4 or explicit_leapfrog_and_hamiltonian" > ${O}_new.txt 2>&1; tail -40 ${O}_new.txt | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_hmc.py -q --tb=short -k "starved or eigenbasis_route_equals or sample_fused_vs_oracle or generic_callback or wave_eigenbasis" > ${O}_hmc.txt 2>&1; tail -30 ${O}_hmc.txt | cut -c1-300
timeout 300 python tools/ab_rmhmc.py 1024:rmhmc_uvc2d=0 1024:- 512:rmhmc_uvc2d=0 512:- 768:rmhmc_uvc2d=0 768:- 1536:rmhmc_uvc2d=0 1536:- 256:- 256:rmhmc_uv_g=2 384:rmhmc_uvc2d=0 384:- > ${O}_ab.txt 2>&1; cat ${O}_ab.txt | cut -c1-250
