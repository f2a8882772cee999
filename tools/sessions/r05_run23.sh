#!/bin/bash
# round 5, GPU session 23: the host API's scatter run by run - parity tests that go through it, then its time per sweep
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -s KILL 1200 python -m pytest tests/test_gpu_parity.py tests/test_random_layouts.py tests/test_edge_problems.py tests/test_cabi_and_solve.py -m gpu -q -x 2>&1 | tail -3
python - <<'PY'
import time, numpy as np
from opengoddard_amd import _native, problems
from opengoddard_amd.engine import HipEngine
for name in ("goddard", "polar_tsto", "low_thrust", "launch4"):
    prob, obj = problems.build(name)
    eng = HipEngine(prob, obj)
    lb = np.array([-np.inf if b[0] is None else b[0] for b in prob.bounds]); ub = np.array([np.inf if b[1] is None else b[1] for b in prob.bounds])
    x = np.clip(prob.p, lb, ub); h = _native.fd_step(x, lb, ub)
    F1, J1 = eng.sweep_stacked(x, h)
    for _ in range(5): F2, J2 = eng.sweep_persistent(x, h)
    assert np.array_equal(J1, J2) and np.array_equal(F1, F2)
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps): eng.sweep_persistent(x, h)
    print("%-12s host_api_ms_per_sweep %.4f" % (name, (time.perf_counter() - t0) / reps * 1e3), flush=True)
    eng.close()
PY
