#!/usr/bin/env python3
"""Measurement script (lives under tests/ because its CPU leg uses the test-only oracle).

Wall-clock to SLSQP convergence, split into callback time and SciPy's SLSQP core
(second half of BASELINE.json's metric).

    python tests/perf/solve_timing.py goddard [--engine hip|oracle] [--max-restarts N]

``--engine oracle`` runs the same solve with the NumPy restatement of the reference path (CPU) so
that both columns come from one machine.  Prints one JSON line.
"""
import argparse, contextlib, io, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from opengoddard_amd import optimize as og, problems


class Timed:
    def __init__(self, inner):
        self.inner, self.t_values, self.t_jac, self.n_values, self.n_jac = inner, 0.0, 0.0, 0, 0
        self._last_v = self._last_j = None

    def values(self, p):
        t = time.perf_counter(); out = self.inner.values(p); self.t_values += time.perf_counter() - t
        self.n_values += 1
        return out

    def jacobians(self, p, lb, ub):
        t = time.perf_counter(); out = self.inner.jacobians(p, lb, ub); self.t_jac += time.perf_counter() - t
        self.n_jac += 1
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("--engine", default="hip", choices=["hip", "oracle"])
    ap.add_argument("--max-restarts", type=int, default=None)
    ap.add_argument("--ftol", type=float, default=None)
    ap.add_argument("--maxiter", type=int, default=None)
    ap.add_argument("--jacobian", default="fd", choices=["fd", "exact"],
                    help="exact: forward-mode derivatives instead of SciPy's forward differences")
    ap.add_argument("--sqp-core", default="scipy", choices=["scipy", "hip"],
                    help="hip: QP subproblems on the GPU (include/ogsqp.h); needs --engine hip")
    ap.add_argument("--time-limit", type=float, default=None,
                    help="seconds: no further SLSQP restart is started after this much wall time (the line then says "
                         "converged false with what was reached; restarts are --maxiter iterations long)")
    ap.add_argument("--cold-start", action="store_true",
                    help="only measure what a NEW problem shape pays before its first sweep: trace + codegen + hipcc "
                         "(forced rebuild of the kernel module) + load, in a fresh process (bench.cold_start)")
    a = ap.parse_args()
    if a.cold_start:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
        import bench
        out = bench.cold_start(a.workload)
        out.update({"workload": a.workload, "first_solve_s": out.get("total_s")})
        print(json.dumps(out))
        return
    prob, obj = problems.build(a.workload)
    if a.max_restarts is not None:
        prob.maxIterator = a.max_restarts
    holder = {}

    def factory(p, o):
        if a.engine == "hip":
            from opengoddard_amd.engine import HipEngine
            inner = HipEngine(p, o)
        else:
            from oracle import np_path
            inner = np_path.NumpyEngine(p, o)
        holder["e"] = Timed(inner)
        return holder["e"]

    if a.sqp_core == "scipy":
        og.ENGINE_FACTORY = factory
    opts = {"ftol": a.ftol} if a.ftol is not None else ({"ftol": 1e-10} if a.workload == "goddard" else {})
    if a.maxiter is not None:
        opts["maxiter"] = a.maxiter
    opts["sqp_core"] = a.sqp_core
    opts["jacobian"] = a.jacobian
    buf = io.StringIO()
    t0 = time.perf_counter()

    class OutOfTime(Exception):
        pass

    def after_restart():
        if a.time_limit is not None and time.perf_counter() - t0 > a.time_limit and prob.last_result.status != 0:
            raise OutOfTime

    stopped = False
    with contextlib.redirect_stdout(buf):
        try:
            prob.solve(obj, after_restart, **opts)
        except OutOfTime:
            stopped = True
    wall = time.perf_counter() - t0
    out = buf.getvalue()
    if a.sqp_core == "hip":
        tm = prob.sqp_timings
        cb = sum(t["callbacks"] for t in tm)
        print(json.dumps({"workload": a.workload, "engine": "hip", "sqp_core": "hip", "jacobian": a.jacobian,
                          "n": int(prob.number_of_variables), "wall_s": wall, "t_callbacks_s": cb,
                          "t_qp_s": sum(t["qp"] for t in tm), "t_bfgs_s": sum(t["bfgs"] for t in tm),
                          "qp_solves": sum(t["qp_solves"] for t in tm),
                          "active_set_iterations": sum(t["qp_iterations"] for t in tm),
                          "t_driver_and_python_s": wall - cb - sum(t["qp"] + t["bfgs"] for t in tm),
                          "major_iterations": int(prob.last_result.nit), "exit_mode": int(prob.last_result.status),
                          "restarts": out.count("---- iteration"), "converged": "successfully" in out,
                          "stopped_by_time_limit": stopped, "recoveries": sum(t.get("recoveries", 0) for t in tm),
                          "sqp_core_used": getattr(prob, "sqp_core_used", None), "cost": float(prob.last_result.fun)}))
        return
    e = holder["e"]
    print(json.dumps({"workload": a.workload, "engine": a.engine, "n": int(prob.number_of_variables),
                      "wall_s": wall, "t_callbacks_s": e.t_values + e.t_jac,
                      "t_values_s": e.t_values, "t_jacobians_s": e.t_jac,
                      "t_slsqp_core_and_python_s": wall - e.t_values - e.t_jac,
                      "value_calls": e.n_values, "jacobian_calls": e.n_jac,
                      "restarts": out.count("---- iteration"), "converged": "successfully" in out,
                      "cost": float(np.atleast_1d(e.inner.values(prob.p)[0])[0])}))


if __name__ == "__main__":
    main()
