#!/bin/bash
# round 5, GPU session 19: after the last two changes of the core (removals spread, T of the small finish through LDS)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -s KILL 900 python -m pytest tests/test_slsqp_core.py -m gpu -q -x -k "wide or sweep_forms or in_block or recovers" 2>&1 | tail -3
tools/sqp_kstats.sh launch4 60 r05_sqp_launch4_60 2>&1 | grep "^k_\|^launch4" | head -14 | cut -c1-200
for i in 1 2; do timeout -s KILL 600 python tests/perf/solve_timing.py launch4 --sqp-core hip 2>/dev/null | tail -1 | cut -c1-330; done
timeout -s KILL 600 python tests/perf/solve_timing.py polar_tsto --sqp-core hip --maxiter 400 2>/dev/null | tail -1 | cut -c1-400
s=$(date +%s)
timeout -s KILL 900 python bench.py > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err
echo "bench.py wall $(( $(date +%s) - s )) s"
cut -c1-300 gpurun_out/r05_bench_final.json
