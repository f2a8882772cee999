#!/bin/bash
# round 6, call 13+: the flag-synchronised LQ panel, steps of tuning: subset of the QP tests, panel trace, kernel statistics
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_slsqp_core.py -x -q -m gpu -k "lq_sweep or first_subproblem_of_the_baseline or random_qps or goldens_at_baseline" 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version\|^$" | tail -3
rm -f tools/_build/libogsqp_trace.so
bash tools/sqp_trace.sh polar_tsto 2 > /dev/null
grep "panel16 at" gpurun_out/sqp_trace_polar_tsto.log | head -3
bash tools/sqp_kstats.sh polar_tsto 10 r06_run12_sqp_polar_tsto 2>&1 | grep -v "Opened result" | grep "k_lq\|sqp_solve\|major" | head -9
bash tools/sqp_kstats.sh low_thrust 10 r06_run12_sqp_low_thrust 2>&1 | grep -v "Opened result" | grep "k_lq\|sqp_solve\|major" | head -9
