#!/bin/bash
# usage (GPU box): tools/sqp_quick.sh [tag] - the SQP core's timing lines used while tuning it: C3 40 x 25 iterations,
# C3 to exit mode 0, C4, C5 (tests/perf/solve_timing.py, HIP core) and bench.py's SQP leg, into gpurun_out/sqp_quick_<tag>.log
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-x}
mkdir -p $R/gpurun_out
out=$R/gpurun_out/sqp_quick_${tag}.log
: > $out
runh() { timeout 600 python $R/tests/perf/solve_timing.py "$@" --sqp-core hip 2>/dev/null | tail -1 | cut -c1-520 >> $out; }
runh polar_tsto
runh polar_tsto --maxiter 400
runh low_thrust
runh launch4 --max-restarts 1 --maxiter 7
timeout 600 python $R/bench.py --steps 50 --warmup 5 --reps 3 --no-cpu-baseline --no-cold-start --sqp-reference-iterations 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d.get('sqp') or {}
print('bench sqp:', {k:s.get(k) for k in ('ms_per_major_iteration','qp_s','active_set_iterations','parity_checked','first_qp_step_error_rel','first_qp_active_set_changes')})" >> $out
cat $out
