#!/bin/bash
# usage (GPU box): tools/solve_profiles.sh <round> - wall-clock-to-convergence lines (tests/perf/solve_timing.py):
# SciPy's Fortran core driven by the GPU callbacks (host API, PCIe inclusive) into gpurun_out/<round>_solve_timing.jsonl,
# the HIP SQP core into gpurun_out/<round>_solve_timing_hip.jsonl
R=${GRAFT_REPO_ROOT:-/root/repo}
rnd=${1:-r02}
out=$R/gpurun_out/${rnd}_solve_timing.jsonl
: > $out
run() { timeout 900 python $R/tests/perf/solve_timing.py "$@" 2>/dev/null | tail -1 >> $out; }
run brachistochrone
run goddard
run polar_tsto_shipped --max-restarts 4
out=$R/gpurun_out/${rnd}_solve_timing_hip.jsonl
: > $out
runh() { timeout 900 python $R/tests/perf/solve_timing.py "$@" --sqp-core hip 2>/dev/null | tail -1 >> $out; }
runh brachistochrone
runh goddard
runh polar_tsto_shipped
runh polar_tsto
runh low_thrust
cat $R/gpurun_out/${rnd}_solve_timing.jsonl $out | cut -c1-330
