#!/bin/bash
# usage (GPU box): tools/wide_ahead_check.sh - the wide LQ sweep with its block reflectors applied on side streams
# (default) against the one-stream order (OGSQP_WIDE_AHEAD=0): tests, then 40 major iterations of C5 each way
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_slsqp_core.py -m gpu -q -x -p no:cacheprovider -k "wide or launch4 or recover or lost or first" 2>&1 | grep -a "passed\|failed\|Error\|error" | head -5
for a in 1 0; do
  OGSQP_WIDE_AHEAD=$a timeout 900 python tests/perf/solve_timing.py launch4 --sqp-core hip --maxiter 40 --max-restarts 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ahead=$a', 'qp_solves', d['qp_solves'], 'ms/QP %.2f' % (1e3*d['t_qp_s']/d['qp_solves']), 'changes', d['active_set_iterations'], 'cost', d['cost'], 'nit', d['major_iterations'])"
done
