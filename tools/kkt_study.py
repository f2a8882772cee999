#!/usr/bin/env python3
"""GPU box: how stationary is the point SLSQP's ``ftol`` test stops at?  ``Problem.solve`` (HIP SQP core) on one workload
for a list of ``ftol`` values; per value one JSON line: exit mode, wall-clock, major iterations, cost and the KKT residuals
by the oracle (oracle/kkt.py).  SLSQP's exit test is on the CHANGE of the cost and the size of the step
(``abs(f - f0) < acc or norm(s) < acc`` with the constraint violation below ``acc``: Kraft's report, SciPy's
``slsqp_optmz.f``; the reference passes ``ftol`` 1e-6, ``optimize.py:735``), not on the gradient of the Lagrangian.

    python tools/kkt_study.py polar_tsto --maxiter 400 --ftol 1e-6,1e-8,1e-10 [--save-x]
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                    # noqa: E402
from opengoddard_amd import problems                  # noqa: E402
from oracle import kkt                                # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("workload")
ap.add_argument("--maxiter", type=int, default=None)
ap.add_argument("--ftol", default="1e-6,1e-8,1e-10")
ap.add_argument("--restarts", type=int, default=None)
ap.add_argument("--time-limit", type=float, default=240.0)
ap.add_argument("--save-x", action="store_true")
ap.add_argument("--jacobian", default="fd", choices=["fd", "exact"])
a = ap.parse_args()
for ftol in [float(v) for v in a.ftol.split(",")]:
    prob, obj = problems.build(a.workload)
    if a.restarts is not None:
        prob.maxIterator = a.restarts
    opts = {"ftol": ftol, "sqp_core": "hip", "jacobian": a.jacobian}
    if a.maxiter is not None:
        opts["maxiter"] = a.maxiter
    t0 = time.perf_counter()

    class OutOfTime(Exception):
        pass

    def after():
        if time.perf_counter() - t0 > a.time_limit and prob.last_result.status != 0:
            raise OutOfTime

    stopped = False
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        try:
            prob.solve(obj, after, **opts)
        except OutOfTime:
            stopped = True
    wall = time.perf_counter() - t0
    res = prob.last_result
    k = kkt.residuals(prob, obj, res.x, prob._engine.m_eq)
    tm = prob.sqp_timings
    print(json.dumps({"workload": a.workload, "ftol": ftol, "jacobian": a.jacobian, "maxiter": a.maxiter, "exit_mode": int(res.status),
                      "stopped_by_time_limit": stopped, "wall_s": wall, "restarts": buf.getvalue().count("---- iteration"),
                      "qp_solves": int(sum(t["qp_solves"] for t in tm)), "cost": float(res.fun),
                      "kkt": {key: k[key] for key in ("kkt", "feasibility", "stationarity", "stationarity_2norm",
                                                      "stationarity_floor_signs_free", "stationarity_floor_2norm", "dual",
                                                      "complementarity", "active_inequalities", "inequalities_at_zero",
                                                      "variables_on_bounds", "largest_multiplier", "gradient_scale")}}),
          flush=True)
    if a.save_x:
        np.savez_compressed(os.path.join(ROOT, "gpurun_out", "kkt_x_%s_%g.npz" % (a.workload, ftol)), x=res.x)
    prob._engine.close()
