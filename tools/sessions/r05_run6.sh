#!/bin/bash
# round 5, GPU session 6: the tests the suite did not reach, the XCD probe
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 120 tools/_build/xcd_probe > gpurun_out/r05_xcd_probe.txt 2>&1
timeout -s KILL 1500 python -m pytest tests/test_gpu_solve.py -m gpu -q -s -k "largest or beyond or tighter" > gpurun_out/r05_t6.log 2>&1
echo "t6 rc $?" >> gpurun_out/r05_t6.log
cat gpurun_out/r05_xcd_probe.txt; grep -v "^$" gpurun_out/r05_t6.log | tail -12 | cut -c1-400
