#!/usr/bin/env python3
"""Development aid (GPU box): run one solve with the HIP SQP core, print one line per QP subproblem (status,
active-set changes, finiteness of the factor and of the multipliers) and save the inputs of the first
subproblems whose answer is not finite to gpurun_out/qp_case_<k>.npz for replay against oracle/slsqp_np.py.

    OGSQP_GI=coop python tests/perf/coop_debug.py polar_tsto_shipped 40 1e-6 exact
"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch                                                     # noqa: E402
from opengoddard_amd import _sqp_native, problems, sqp           # noqa: E402

name, maxiter, ftol, jac = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), sys.argv[4]
out_dir = os.path.join(ROOT, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
solve_dev = _sqp_native.QpCore.solve_dev
device_jacobian_init = sqp.DeviceJacobian.__init__
state = {"count": 0, "saved": 0, "jacobian": None}


def traced_init(self, engine):
    device_jacobian_init(self, engine)
    state["jacobian"] = self


def traced_solve(self, d_jt, ld, g, c, dl, du, augmented=False, rho=100.0, stream=0):
    Z = self.get_factor()
    d, mult, bm, status, iters = res = solve_dev(self, d_jt, ld, g, c, dl, du, augmented, rho, stream)
    state["count"] += 1
    finite = bool(np.all(np.isfinite(d)) and np.all(np.isfinite(mult)))
    print("[qp %d] augmented %d rho %g status %d changes %d |d| %.3e finite: step+multipliers %s factor %s bounds %s" % (
        state["count"], augmented, rho, status, iters, float(np.max(np.abs(d))) if finite else np.nan, finite,
        bool(np.all(np.isfinite(Z))), bool(np.all(np.isfinite(bm)))), file=sys.stderr, flush=True)
    if not finite and state["saved"] < 3:
        torch.cuda.synchronize()
        JT = state["jacobian"].d_JT.cpu().numpy().reshape(self.n, ld)
        np.savez(os.path.join(out_dir, "qp_case_%d.npz" % state["saved"]), Z=Z, g=np.asarray(g), c=np.asarray(c),
                 dl=np.asarray(dl), du=np.asarray(du), augmented=augmented, rho=rho, ld=ld, status=status, JT=JT,
                 d=d, mult=mult, bm=bm)
        state["saved"] += 1
    return res


sqp.DeviceJacobian.__init__ = traced_init
_sqp_native.QpCore.solve_dev = traced_solve
prob, obj = problems.build(name)
prob.maxIterator = 1
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    prob.solve(obj, maxiter=maxiter, ftol=ftol, sqp_core="hip", jacobian=jac)
r = prob.last_result
print("status", r.status, "nit", r.nit, "fun", r.fun, file=sys.stderr)
