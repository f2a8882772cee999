#!/bin/bash
# round 5, GPU session 12: after the short-row default - the SQP tests, then the bench line as the driver runs it
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -s KILL 1500 python -m pytest tests/test_slsqp_core.py tests/test_gpu_solve.py -m gpu -q -x --deselect "tests/test_slsqp_core.py::test_gpu_first_subproblem_of_the_baseline_configurations[launch4]" 2>&1 | tail -4 > gpurun_out/r05_t12.log
cat gpurun_out/r05_t12.log
/usr/bin/time -v timeout -s KILL 900 python bench.py > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err
grep -E "Elapsed|Maximum resident" gpurun_out/r05_bench_final.err
cut -c1-300 gpurun_out/r05_bench_final.json
