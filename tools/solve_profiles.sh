#!/bin/bash
# usage (GPU box): tools/solve_profiles.sh <round> - wall-clock-to-convergence lines (tests/perf/solve_timing.py):
# SciPy's Fortran core driven by the GPU callbacks (host API, PCIe inclusive) into gpurun_out/<round>_solve_timing.jsonl,
# the HIP SQP core into gpurun_out/<round>_solve_timing_hip.jsonl (reference defaults - maxiter 25 per restart - unless
# the line says otherwise; the C3 line with --maxiter 400 runs to exit mode 0; C5: the first major iterations)
R=${GRAFT_REPO_ROOT:-/root/repo}
rnd=${1:-r04}
mkdir -p $R/gpurun_out
out=$R/gpurun_out/${rnd}_solve_timing.jsonl
: > $out
run() { timeout 900 python $R/tests/perf/solve_timing.py "$@" --sqp-core scipy 2>/dev/null | tail -1 >> $out; }
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
runh polar_tsto --maxiter 400
runh low_thrust
runh low_thrust --maxiter 1000 --max-restarts 3
# C5 (round 4: a well-posed problem, the wide LQ sweep): the reference's defaults, then to exit mode 0
runh launch4
runh launch4 --maxiter 3000 --max-restarts 3 --time-limit 900
# what a new problem shape pays before its first sweep (forced rebuild of its kernel module)
cold=$R/gpurun_out/${rnd}_cold_start.jsonl
: > $cold
for w in goddard polar_tsto low_thrust launch4; do
    timeout 900 python $R/tests/perf/solve_timing.py $w --cold-start 2>/dev/null | tail -1 >> $cold
done
cat $R/gpurun_out/${rnd}_solve_timing.jsonl $out $cold | cut -c1-330
