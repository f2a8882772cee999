#!/bin/bash
# round 5, GPU session 3: the fused in-block update, C5 kernel statistics with it, KKT with exact Jacobians at C3
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_slsqp_core.py -m gpu -x -q -k "in_block or wide or sweep_forms or recovers" > gpurun_out/r05_t5.log 2>&1
echo "t5 rc $?" >> gpurun_out/r05_t5.log
tools/sqp_kstats.sh launch4 60 r05_sqp_launch4_60_inblock > gpurun_out/r05_sqp_launch4_60_inblock.txt 2>&1
timeout 600 python tools/kkt_study.py polar_tsto --maxiter 400 --ftol 1e-6,1e-9 --jacobian exact > gpurun_out/r05_kkt_polar_tsto_exact.jsonl 2> gpurun_out/r05_kkt_exact.err
tail -3 gpurun_out/r05_t5.log; cut -c1-600 gpurun_out/r05_kkt_polar_tsto_exact.jsonl; tail -16 gpurun_out/r05_sqp_launch4_60_inblock.txt
