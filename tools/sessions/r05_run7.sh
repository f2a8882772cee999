#!/bin/bash
# round 5, GPU session 7: the panel's exchange in one trip - wide tests, C5 wall-clock A/B, the C5 check of the suite
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -s KILL 900 python -m pytest tests/test_slsqp_core.py -m gpu -x -q -k "wide or sweep_forms or recovers" > gpurun_out/r05_t7.log 2>&1
echo "t7 rc $?" >> gpurun_out/r05_t7.log
out=gpurun_out/r05_ab2_launch4.jsonl
: > $out
run() { tag=$1; shift; env "$@" timeout -s KILL 600 python tests/perf/solve_timing.py launch4 --sqp-core hip 2>/dev/null | tail -1 | sed "s/^{/{\"variant\": \"$tag\", /" >> $out; }
run default X=1
run default_again X=1
timeout -s KILL 900 python -m pytest tests/test_gpu_solve.py -m gpu -q -s -k "largest" > gpurun_out/r05_t7b.log 2>&1
echo "t7b rc $?" >> gpurun_out/r05_t7b.log
tools/sqp_kstats.sh launch4 60 r05_sqp_launch4_60_poll > gpurun_out/r05_sqp_launch4_60_poll.txt 2>&1
tail -3 gpurun_out/r05_t7.log; cut -c1-330 $out; grep -v "^$" gpurun_out/r05_t7b.log | tail -4 | cut -c1-500; tail -8 gpurun_out/r05_sqp_launch4_60_poll.txt
