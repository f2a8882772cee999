#!/bin/bash
# round 5, GPU session 2: streamed rows vs register kernels, KKT vs ftol, C5 kernel statistics, bench with the measured chain
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_slsqp_core.py -m gpu -x -q -k "streamed_rows or random_qps or warm_started or replays_scipy_iterates" > gpurun_out/r05_t4.log 2>&1
echo "t4 rc $?" >> gpurun_out/r05_t4.log
timeout 900 python tools/kkt_study.py polar_tsto --maxiter 400 --ftol 1e-6,1e-8,1e-10 --save-x > gpurun_out/r05_kkt_polar_tsto.jsonl 2> gpurun_out/r05_kkt.err
timeout 300 python tools/kkt_study.py low_thrust --ftol 1e-6,1e-8,1e-10 > gpurun_out/r05_kkt_low_thrust.jsonl 2>> gpurun_out/r05_kkt.err
tools/sqp_kstats.sh launch4 60 r05_sqp_launch4_60 > gpurun_out/r05_sqp_launch4_60.txt 2>&1
OGSQP_ROWS=reg tools/sqp_kstats.sh launch4 60 r05_sqp_launch4_60_rowsreg > gpurun_out/r05_sqp_launch4_60_rowsreg.txt 2>&1
OGSQP_ROWS=stage tools/sqp_kstats.sh launch4 60 r05_sqp_launch4_60_stage > gpurun_out/r05_sqp_launch4_60_stage.txt 2>&1
tools/sqp_kstats.sh polar_tsto 10 r05_sqp_polar_tsto > gpurun_out/r05_sqp_polar_tsto.txt 2>&1
timeout 900 python bench.py --no-solve > gpurun_out/r05_bench2.json 2> gpurun_out/r05_bench2.err
tail -3 gpurun_out/r05_t4.log; cat gpurun_out/r05_kkt_*.jsonl | cut -c1-400; tail -16 gpurun_out/r05_sqp_launch4_60.txt
