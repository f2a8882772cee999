#!/bin/bash
# round 5, GPU session 5: the whole GPU suite, then the bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/r05_gputests.log 2>&1
echo "suite rc $?" >> gpurun_out/r05_gputests.log
timeout 300 python tools/kkt_study.py low_thrust --maxiter 400 --ftol 1e-8 > gpurun_out/r05_kkt_low_thrust_400.jsonl 2>/dev/null
timeout 900 python bench.py > gpurun_out/r05_bench3.json 2> gpurun_out/r05_bench3.err
tail -30 gpurun_out/r05_gputests.log
