#!/usr/bin/env python3
"""Development aid: run one solve with the HIP SQP core and record every QP subproblem whose answer differs
between the cooperative and the single-workgroup active-set kernels (inputs to gpurun_out/qp_case_*.npz).

    OGSQP_GI=coop python tests/perf/coop_debug.py polar_tsto_shipped 40 1e-6 exact
"""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                                     # noqa: E402
from opengoddard_amd import _sqp_native, problems               # noqa: E402

name, maxiter, ftol, jac = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), sys.argv[4]
out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
orig = _sqp_native.QpCore.solve_dev
count = {"n": 0, "saved": 0}


def wrapped(self, d_jt, ld, g, c, dl, du, augmented=False, rho=100.0, stream=0):
    Z = self.get_factor()
    res = orig(self, d_jt, ld, g, c, dl, du, augmented, rho, stream)
    d, mult, bm, status, iters = res
    count["n"] += 1
    bad = not np.all(np.isfinite(d)) or not np.all(np.isfinite(mult))
    print("[qp %d] aug %d rho %g status %d iters %d |d| %.3e finite %s" % (
        count["n"], augmented, rho, status, iters, float(np.max(np.abs(d))) if np.all(np.isfinite(d)) else np.nan,
        not bad), file=sys.stderr, flush=True)
    print("        finite Z %s bm %s mult %s  |Z|max %.3e" % (bool(np.all(np.isfinite(Z))), bool(np.all(np.isfinite(bm))),
          bool(np.all(np.isfinite(mult))), float(np.nanmax(np.abs(Z)))), file=sys.stderr, flush=True)
    if (bad or not np.all(np.isfinite(Z))) and count["saved"] < 3:
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, "qp_case_%d.npz" % count["saved"]), Z=Z, g=np.asarray(g), c=np.asarray(c),
                 dl=np.asarray(dl), du=np.asarray(du), augmented=augmented, rho=rho, ld=ld, status=status,
                 JT=HOLD["jac"].d_JT.cpu().numpy().reshape(self.n, ld), d=d, mult=mult, bm=bm)
        count["saved"] += 1
    return res


HOLD = {}
from opengoddard_amd import sqp                                  # noqa: E402
orig_dj = sqp.DeviceJacobian.__init__


def dj_init(self, engine):
    orig_dj(self, engine)
    HOLD["jac"] = self


sqp.DeviceJacobian.__init__ = dj_init
_sqp_native.QpCore.solve_dev = wrapped
prob, obj = problems.build(name)
prob.maxIterator = 1
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    prob.solve(obj, maxiter=maxiter, ftol=ftol, sqp_core="hip", jacobian=jac)
r = prob.last_result
print("status", r.status, "nit", r.nit, "fun", r.fun, file=sys.stderr)
