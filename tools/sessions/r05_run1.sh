#!/bin/bash
# round 5, GPU session 1: the wide sweep with k_wy_update, KKT tests, start for C5, bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_slsqp_core.py -m gpu -x -q -s -k "wide or recovers or sweep_forms or first_subproblem" > gpurun_out/r05_t1.log 2>&1
echo "t1 rc $?" >> gpurun_out/r05_t1.log
timeout 900 python -m pytest tests/test_gpu_solve.py -m gpu -q -s -k "kkt and not largest" > gpurun_out/r05_t2.log 2>&1
echo "t2 rc $?" >> gpurun_out/r05_t2.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_edge_problems.py -m gpu -q -k "lgl or edge or r1" > gpurun_out/r05_t3.log 2>&1
echo "t3 rc $?" >> gpurun_out/r05_t3.log
timeout 1200 python tools/make_start_launch4.py > gpurun_out/r05_start.log 2>&1
timeout 900 python bench.py > gpurun_out/r05_bench1.json 2> gpurun_out/r05_bench1.err
tail -3 gpurun_out/r05_t1.log gpurun_out/r05_t2.log gpurun_out/r05_t3.log gpurun_out/r05_start.log
