#!/bin/bash
# round 5, GPU session 9: the two new re-authored examples on the GPU, the C5 replay of SciPy's golden, streamed rows after the default change
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -s KILL 1200 python -m pytest tests -m gpu -q -k "goddard_1knot or polar_ssto or goldens_at_baseline_sizes or streamed_rows or largest" --durations=5 > gpurun_out/r05_t9.log 2>&1
echo "t9 rc $?" >> gpurun_out/r05_t9.log
grep -v "^$" gpurun_out/r05_t9.log | tail -14 | cut -c1-300
