#!/usr/bin/env python3
"""Golden vectors for the SQP core at the BASELINE sizes: what SciPy 1.15.3's Fortran SLSQP does on C3 / C4.

Run in the build container only (it imports the reference from /root/reference, which never travels):

    python tools/make_golden_slsqp.py [--only NAME ...]

One major iteration of SciPy's SLSQP costs 16 s at C3 (n = 1442) and 25 s at C4 (n = 2001), so the replay tests
of ``tests/test_slsqp_core.py`` cannot run SciPy at these sizes on every test run (they do for n <= 282); its
first major iterations are captured here instead, twice:

``*_twin``  SciPy driven by this repo's callbacks (the compiled CPU twin of the generated code, forward
            differences with SciPy's step rule) - the inputs the replay on the GPU box reproduces bit for bit.
            Stored: the points at which SciPy asks for the constraint Jacobian (= its accepted iterates),
            ``status / nit / nfev / njev / fun``.
``*_ref``   the **reference itself**: ``OpenGoddard.optimize.Problem.solve`` of /root/reference
            (``optimize.py:723-749``: callbacks without ``jac=``, SciPy differencing them) on the same problem
            definition, the iterate after every major iteration taken from SciPy's ``callback=`` hook.
            The two runs see Jacobians that differ by forward-difference rounding (1e-8 relative), so their
            iterates agree to about 1e-6 of the step, not to the bit; the test states that bound.

Nothing is written under /root/reference.
"""
import argparse
import os
import sys
import time
import warnings

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
warnings.filterwarnings("ignore")

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REF)
sys.path.append(REPO)
sys.path.append(os.path.join(REPO, "tests"))
sys.path.append(os.path.join(REPO, "tools"))

import numpy as np                      # noqa: E402
import scipy                            # noqa: E402
import scipy.optimize as sciopt         # noqa: E402

import make_golden                      # noqa: E402  (capture(): the reference's call site intercepted)
import OpenGoddard.optimize as ref      # noqa: E402

assert ref.__file__.startswith(REF), ref.__file__

# (C5: one major iteration of SciPy's Fortran core at n = 6148 is ~10 minutes of O(n^3) work - two are what is affordable)
CASES = {"polar_tsto": dict(maxiter=4, ftol=1e-6), "low_thrust": dict(maxiter=3, ftol=1e-6),
         "launch4": dict(maxiter=2, ftol=1e-6)}


def run_twin(name, maxiter, ftol, checkpoint=None):
    """``checkpoint(k, data)``: called when SciPy asks for the constraint Jacobian at its k-th accepted iterate, with
    the golden a ``maxiter = k + 1`` run would have produced.  (SciPy hands the Fortran core ``itermx = maxiter - 1``
    (``_slsqp_py.py``); the core evaluates the gradients at iterate k, does the BFGS update, increments ``iter`` to
    k + 1 and only then compares it with the limit -> ``mode = 9`` - so at that moment ``nit = njev = k + 1``,
    ``nfev`` = the objective evaluations so far and ``fun`` = the objective at that iterate are exactly what the
    shorter run reports;
    ``--check-checkpoints`` verifies that on a small configuration.)  One major iteration at C5 is 1.6 hours of the
    Fortran core: a run that is cut short still leaves a usable golden behind."""
    from test_slsqp_core import Callbacks
    cb = Callbacks(name)
    t = time.time()
    if checkpoint is not None:
        plain_fun, plain_jac = cb.fun, cb.jac

        def marking_jac(x, mark=False):
            out = plain_jac(x, mark)
            k = len(cb.iterates) - 1
            if mark and k >= 1 and k < maxiter - 1 and not any(np.array_equal(x, it) for it in cb.iterates[:-1]):
                # (SciPy asks ScalarFunction for f(x) once per new point; the eq/ineq lambdas call cb.fun too, so the
                #  objective's own evaluations are counted in objective() below, not here)
                checkpoint(k, dict(iterates_twin=np.array(cb.iterates), x_twin=np.array(x), status_twin=np.int64(9),
                                   nit_twin=np.int64(k + 1), nfev_twin=np.int64(objective_calls[0]), njev_twin=np.int64(k + 1),
                                   fun_twin=np.float64(plain_fun(x)[0]), x0=np.array(cb.prob.p, dtype=float),
                                   lb=cb.lb, ub=cb.ub, m_eq=np.int64(cb.meq), maxiter=np.int64(k + 1)))
                print("  checkpoint after major iteration %d (%.0f s)" % (k, time.time() - t), flush=True)
            return out

        cb.jac = marking_jac
        objective_calls = [0]
        res = scipy_counting_objective(cb, maxiter, ftol, objective_calls)
    else:
        res = cb.scipy(maxiter, ftol)
    print("  twin-driven SciPy: status %d nit %d nfev %d njev %d fun %.10g (%.0f s)" % (
        res.status, res.nit, res.nfev, res.njev, res.fun, time.time() - t), flush=True)
    return dict(iterates_twin=np.array(cb.iterates), x_twin=np.array(res.x), status_twin=np.int64(res.status),
                nit_twin=np.int64(res.nit), nfev_twin=np.int64(res.nfev), njev_twin=np.int64(res.njev),
                fun_twin=np.float64(res.fun), x0=np.array(cb.prob.p, dtype=float), lb=cb.lb, ub=cb.ub,
                m_eq=np.int64(cb.meq))


def scipy_counting_objective(cb, maxiter, ftol, counter):
    """``Callbacks.scipy`` (tests/test_slsqp_core.py:70) with the objective's evaluations counted (= ``res.nfev``)."""
    meq = cb.meq

    def objective(x):
        counter[0] += 1
        return cb.fun(x)[0]

    cons = [{"type": "eq", "fun": lambda x: cb.fun(x)[1][:meq], "jac": lambda x: cb.jac(x, True)[1][:meq]},
            {"type": "ineq", "fun": lambda x: cb.fun(x)[1][meq:], "jac": lambda x: cb.jac(x)[1][meq:]}]
    with np.errstate(all="ignore"):
        return sciopt.minimize(objective, cb.prob.p.copy(), jac=lambda x: cb.jac(x)[0],
                               bounds=list(zip(cb.lb, cb.ub)), constraints=cons, method="SLSQP",
                               options={"maxiter": maxiter, "ftol": ftol})


def run_reference(name, maxiter, ftol):
    from opengoddard_amd import problems
    holder = {}

    def run():
        prob, obj = problems.build(name, api=ref)
        holder["prob"] = prob
        prob.solve(obj)

    got = make_golden.capture(run)
    iterates = []
    t = time.time()
    res = sciopt.minimize(got["fun"], got["x0"].copy(), args=got["args"], bounds=got["bounds"],
                          constraints=got["constraints"], jac=got["jac"], method="SLSQP",
                          callback=lambda xk: iterates.append(np.array(xk, dtype=float)),
                          options={"disp": False, "maxiter": maxiter, "ftol": ftol})
    print("  the reference (Problem.solve's own call): status %d nit %d nfev %d fun %.10g (%.0f s)" % (
        res.status, res.nit, res.nfev, res.fun, time.time() - t), flush=True)
    return dict(iterates_ref=np.array(iterates), x_ref=np.array(res.x), status_ref=np.int64(res.status),
                nit_ref=np.int64(res.nit), nfev_ref=np.int64(res.nfev), fun_ref=np.float64(res.fun))


def check_checkpoints(name, maxiter):
    """Checkpoint k of one long run == the result of a separate ``maxiter = k`` run, field for field."""
    ftol = CASES.get(name, dict(ftol=1e-6))["ftol"]
    kept = {}
    run_twin(name, maxiter, ftol, checkpoint=lambda k, d: kept.__setitem__(k, d))
    for k in sorted(kept):
        short = run_twin(name, k + 1, ftol)
        for key, v in short.items():
            assert np.array_equal(np.asarray(v), np.asarray(kept[k][key])), (k, key, v, kept[k][key])
        print("  checkpoint %d == maxiter=%d run (%s)" % (k, k + 1, ", ".join(sorted(short))), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--maxiter", type=int, default=None, help="override the case's number of major iterations")
    ap.add_argument("--no-reference", action="store_true",
                    help="only the twin-driven SciPy run (the GPU replay uses nothing else; halves the hours C5 takes)")
    ap.add_argument("--check-checkpoints", metavar="NAME", default=None,
                    help="verify on a (small) configuration that checkpoint k equals the golden of a maxiter = k run")
    a = ap.parse_args()
    if a.check_checkpoints:
        return check_checkpoints(a.check_checkpoints, a.maxiter or 4)
    for name, opts in CASES.items():
        if a.only and name not in a.only:
            continue
        print("golden slsqp:", name, opts, flush=True)
        data = dict(maxiter=np.int64(opts["maxiter"]), ftol=np.float64(opts["ftol"]),
                    scipy_version=np.array(scipy.__version__))
        if a.maxiter is not None:
            opts = dict(opts, maxiter=a.maxiter)
            data["maxiter"] = np.int64(a.maxiter)
        base = dict(ftol=np.float64(opts["ftol"]), scipy_version=np.array(scipy.__version__))
        path = os.path.join(OUT, "slsqp_%s.npz" % name)

        def checkpoint(k, partial, base=base, path=path):
            np.savez_compressed(path + ".tmp.npz", **dict(base, **partial))
            os.replace(path + ".tmp.npz", path)

        data.update(run_twin(name, checkpoint=checkpoint if a.no_reference else None, **opts))
        if a.no_reference:
            np.savez_compressed(os.path.join(OUT, "slsqp_%s.npz" % name), **data)
            continue
        data.update(run_reference(name, **opts))
        k = min(len(data["iterates_ref"]), len(data["iterates_twin"]) - 1)
        for i in range(k):
            step = np.max(np.abs(data["iterates_twin"][i + 1] - data["iterates_twin"][i]))
            diff = np.max(np.abs(data["iterates_ref"][i] - data["iterates_twin"][i + 1]))
            print("  iterate %d: |twin-driven - reference| = %.3e, step %.3e" % (i + 1, diff, step), flush=True)
        np.savez_compressed(os.path.join(OUT, "slsqp_%s.npz" % name), **data)


if __name__ == "__main__":
    main()
