#!/bin/bash
# usage (GPU box): tools/r04_explore.sh - the GPU suite (log kept), then wall-clock-to-convergence probes of C3/C4/C5
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r04_gputests_first.log
out=gpurun_out/r04_solve_probe.jsonl
: > $out
runh() { timeout 1200 python tests/perf/solve_timing.py "$@" --sqp-core hip 2>gpurun_out/last_err.log | tail -1 >> $out; tail -3 gpurun_out/last_err.log | cut -c1-300 >> $out.err; }
runh polar_tsto
runh polar_tsto --maxiter 400
runh low_thrust
runh low_thrust --maxiter 1000 --max-restarts 3
runh launch4 --maxiter 40 --max-restarts 1
cat gpurun_out/r04_gputests_first.log | tail -25
cut -c1-420 $out
