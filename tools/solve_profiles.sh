#!/bin/bash
# usage (GPU box): tools/solve_profiles.sh <round> - wall-clock-to-convergence lines with the HIP SQP core
# (tests/perf/solve_timing.py) into gpurun_out/<round>_solve_timing_hip.jsonl
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${1:-r01}_solve_timing_hip.jsonl
: > $out
run() { timeout 900 python $R/tests/perf/solve_timing.py "$@" --sqp-core hip 2>/dev/null | tail -1 >> $out; }
run brachistochrone
run goddard
run polar_tsto_shipped
run polar_tsto
run low_thrust
run polar_tsto --maxiter 400
cat $out | cut -c1-300
