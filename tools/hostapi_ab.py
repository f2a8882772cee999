import sys, time, os, numpy as np
sys.path.insert(0, '/root/repo')
from opengoddard_amd import _native, problems
from opengoddard_amd.engine import HipEngine
import subprocess
print(open('/sys/kernel/mm/transparent_hugepage/enabled').read().strip() if os.path.exists('/sys/kernel/mm/transparent_hugepage/enabled') else 'no thp file')
for name in ["polar_tsto", "low_thrust"]:
  for alloc in ("pinned", "numpy"):
    prob, obj = problems.build(name)
    eng = HipEngine(prob, obj)
    if alloc == "numpy":
        _native.pinned_matrix_saved = _native.pinned_matrix
        _native.pinned_matrix = lambda r, c: None
    lb = np.array([-np.inf if b[0] is None else b[0] for b in prob.bounds]); ub = np.array([np.inf if b[1] is None else b[1] for b in prob.bounds])
    x0 = np.clip(prob.p, lb, ub); h = _native.fd_step(x0, lb, ub)
    for _ in range(12): eng.sweep_persistent(x0, h)
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps): eng.sweep_persistent(x0, h)
    dt = (time.perf_counter() - t0) / reps * 1e3
    F0, JT = eng.sweep_persistent(x0, h)
    F1, J1 = eng.sweep_stacked(x0, h)
    print(name, alloc, "host_api_ms_per_sweep %.4f" % dt, "path", eng.host_path, "equal", bool(np.array_equal(JT, J1) and np.array_equal(F0, F1)), flush=True)
    eng.close()
    if alloc == "numpy":
        _native.pinned_matrix = _native.pinned_matrix_saved
