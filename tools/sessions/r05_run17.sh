#!/bin/bash
# round 5, GPU session 17: warm-start removals spread over the grid - bits, the SQP suite, C3 kernel statistics and leg
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -s KILL 1500 python -m pytest tests/test_slsqp_core.py tests/test_gpu_solve.py -m gpu -q -x --deselect "tests/test_slsqp_core.py::test_gpu_first_subproblem_of_the_baseline_configurations[launch4]" 2>&1 | tail -6 > gpurun_out/r05_t17.log
cat gpurun_out/r05_t17.log | cut -c1-300
for v in 1 0 1 0; do
  OGSQP_WARM_SPREAD=$v tools/sqp_kstats.sh polar_tsto 10 r05_spread_$v 2>&1 | grep "k_rows_decide\|^polar" | cut -c1-260
done
OGSQP_WARM_SPREAD=1 tools/sqp_kstats.sh launch4 60 r05_spread_c5_1 2>&1 | grep "k_rows_decide\|^launch4" | cut -c1-260
OGSQP_WARM_SPREAD=0 tools/sqp_kstats.sh launch4 60 r05_spread_c5_0 2>&1 | grep "k_rows_decide\|^launch4" | cut -c1-260
