"""usage (GPU box): python tools/hostapi_timing.py [workload ...] - ms per host-pointer sweep (PCIe included) into the
engine's persistent host matrix, and which path served it (OGPSX_HOST=mapped|staged|copyx, OGPSX_TIMING=1 for the breakdown)."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from opengoddard_amd import _native, problems
from opengoddard_amd.engine import HipEngine
for name in (sys.argv[1:] or ["polar_tsto"]):
    prob, obj = problems.build(name)
    eng = HipEngine(prob, obj)
    lb = np.array([-np.inf if b[0] is None else b[0] for b in prob.bounds]); ub = np.array([np.inf if b[1] is None else b[1] for b in prob.bounds])
    x0 = np.clip(prob.p, lb, ub); h = _native.fd_step(x0, lb, ub)
    for _ in range(10): eng.sweep_persistent(x0, h)
    reps = 256
    t0 = time.perf_counter()
    for _ in range(reps): eng.sweep_persistent(x0, h)
    print(name, "host_api_ms_per_sweep %.4f" % ((time.perf_counter() - t0) / reps * 1e3), "path", eng.host_path, flush=True)
    eng.close()
