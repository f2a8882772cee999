"""End-to-end on the GPU: Problem.solve driving SciPy SLSQP with HIP-evaluated callbacks and
Jacobians (G5 of SURVEY.md section 8(c): converged objectives, not iteration counts)."""
import time

import numpy as np
import pytest

from opengoddard_amd import problems

pytestmark = pytest.mark.gpu


def test_brachistochrone_converges_to_sqrt_pi(capsys):
    prob, obj = problems.build("brachistochrone")
    prob.solve(obj)
    out = capsys.readouterr().out
    assert "Optimization terminated successfully" in out
    # default ftol is 1e-6, and the iteration path depends on last-ulp rounding
    assert abs(prob.time_final(-1) - np.sqrt(np.pi)) < 2e-5      # reference run: 1.77245410898455
    eng = prob._engine
    assert eng.n_sweeps > 0 and eng.n_values > 0
    eng.close()


def test_goddard_converges_to_literature_altitude(capsys):
    prob, obj = problems.build("goddard")
    t0 = time.perf_counter()
    prob.solve(obj, ftol=1e-10)
    wall = time.perf_counter() - t0
    out = capsys.readouterr().out
    assert "Optimization terminated successfully" in out
    h_end = prob.states_all_section(0)[-1]
    assert abs(h_end - 1.0128342) < 1e-6                        # reference: 1.0128341906902811
    eng = prob._engine
    print("goddard: %.2f s wall, %d sweeps, %d evaluations" % (wall, eng.n_sweeps, eng.n_values))
    eng.close()


def test_solve_records_q13_like_the_reference():
    """After a gradient request the reference leaves prob.p at the last FD column's input:
    p == x except p[-1] = x[-1] + h[-1] (quirk Q13)."""
    prob, obj = problems.build("goddard")
    prob.maxIterator = 1
    prob.solve(obj, maxiter=2, sqp_core="scipy")          # SciPy's core asking the engine's callbacks
    eng = prob._engine
    (_, _, _), h = eng._jac
    x = np.frombuffer(eng._jac_key, dtype=np.float64)
    assert np.array_equal(prob.p[:-1], x[:-1]) and prob.p[-1] == x[-1] + h[-1]
    eng.close()
    # the HIP SQP core (the default at this size since round 4) leaves the same trace: its last linearisation's x with
    # the last column's step on the last entry
    prob, obj = problems.build("goddard")
    prob.maxIterator = 1
    prob.solve(obj, maxiter=2)
    assert prob.sqp_core_used == "hip"
    jac = prob._engine._sqp_cache[0]
    assert np.array_equal(prob.p[:-1], prob.last_result.x[:-1])
    assert prob.p[-1] == prob.last_result.x[-1] + jac.last_step[-1]
    prob._engine.close()


def test_devices_and_the_hip_sqp_core_work_together(monkeypatch, capsys):
    """``Problem.solve(devices=[...])`` with the default SQP core at a size BASELINE shards (C4, split in 4).
    The FD columns of every sweep are split over the listed devices (peer mode lets the 1-GPU box list device 0 four times:
    four sub-handles, four streams, the real exchange), the QP core reads device d0's replica in place - and the run is
    the single-device run, bit for bit: the sharded matrix is the same matrix, so every iterate is the same."""
    import warnings
    monkeypatch.setenv("OGPSX_GATHER", "peer")

    def run(devices):
        prob, obj = problems.build("low_thrust")
        prob.maxIterator = 1
        with warnings.catch_warnings():
            warnings.simplefilter("error", RuntimeWarning)          # no "devices= is not used" warning any more
            prob.solve(obj, maxiter=8, devices=devices)
        capsys.readouterr()
        eng = prob._engine
        jac = eng._sqp_cache[0]
        out = (prob.last_result.x.copy(), prob.p.copy(), prob.last_result.nit, prob.last_result.nfev,
               prob.last_result.fun, prob.sqp_core_used, jac.sharded_over)
        eng.close()
        return out

    one = run(None)
    four = run([0, 0, 0, 0])
    assert one[5] == four[5] == "hip" and one[6] == 1 and four[6] == 4
    assert np.array_equal(one[0], four[0]) and np.array_equal(one[1], four[1])
    assert one[2:5] == four[2:5]
    # the exact-Jacobian mode is a single-device kernel: with devices= it keeps the one-device buffers (no sharding, no
    # failure), and the iterates are those of the run without devices=
    def run_exact(devices):
        prob, obj = problems.build("goddard")
        prob.maxIterator = 1
        prob.solve(obj, maxiter=6, devices=devices, jacobian="exact", sqp_core="hip")
        capsys.readouterr()
        out = (prob.last_result.x.copy(), prob.last_result.nit, prob._engine._sqp_cache[0].sharded_over)
        prob._engine.close()
        return out
    a, b = run_exact(None), run_exact([0, 0])
    assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2] == 1


@pytest.mark.parametrize("name,maxiter", [("launch4", 11), ("low_thrust", 11)])
def test_the_redefined_baseline_problems_descend_without_a_non_finite_number(name, maxiter, capsys):
    """C5's round-1 form overflowed to NaN between the 5th and the 8th major iteration in every core.  The
    problems as redefined in round 4 (C5: two-stage minimum-effort ascent, C4: 3-D minimum-energy transfer): the first
    ten major iterations of the default solve keep every iterate, the cost, its gradient and every constraint value
    finite and move the iterate (profiles/r04_solve_timing_hip.jsonl has the runs to exit mode 0)."""
    from opengoddard_amd.engine import HipEngine
    prob, obj = problems.build(name)
    prob.maxIterator = 1
    x0 = prob.p.copy()
    prob.solve(obj, maxiter=maxiter)
    capsys.readouterr()
    res = prob.last_result
    assert prob.sqp_core_used == "hip" and res.status == 9 and res.nit >= maxiter - 1
    assert np.all(np.isfinite(res.x)) and np.isfinite(res.fun) and np.all(np.isfinite(res.jac))
    eng = prob._engine
    F0, F1 = eng.eval_stacked(x0), eng.eval_stacked(res.x)
    assert np.all(np.isfinite(F0)) and np.all(np.isfinite(F1))
    assert not np.array_equal(res.x, np.clip(x0, *[np.array(v, dtype=float) for v in zip(*[(-np.inf if b[0] is None else b[0],
                                                                                            np.inf if b[1] is None else b[1])
                                                                                           for b in prob.bounds])]))
    eng.close()


# ------------------------------------------------------------------ converged optima against the oracle
KKT_BOUND = 1e-5


def _kkt_of_a_solve(name, options, capsys, start=None):
    """``Problem.solve`` with the default (HIP) SQP core to exit mode 0, then the KKT residuals of the reference's NLP at
    the returned point - every value from the NumPy restatement of the reference path (oracle/kkt.py on
    oracle/np_path.py: the Problem's own callbacks, central differences, least-squares multipliers).  SciPy's Fortran
    core cannot finish these sizes (8 s .. 1.6 h per major iteration), so this is the independent statement that the
    point IS an optimum of the problem ``optimize.py:723-749`` hands to SLSQP."""
    from oracle import kkt
    prob, obj = problems.build(name)
    if start is not None:
        prob.p = np.array(start, dtype=float)
    t0 = time.perf_counter()
    prob.solve(obj, **options)
    wall = time.perf_counter() - t0
    capsys.readouterr()
    res = prob.last_result
    assert prob.sqp_core_used == "hip"
    assert res.status == 0, (res.status, res.message)
    k = kkt.residuals(prob, obj, res.x, prob._engine.m_eq)
    print("%s: exit mode 0 after %.1f s, cost %.9g (oracle %.9g); KKT by the oracle: %s" % (
        name, wall, res.fun, k["cost"], {key: k[key] for key in (
            "kkt", "feasibility", "stationarity", "dual", "complementarity", "active_inequalities",
            "variables_on_bounds")}))
    prob._engine.close()
    return res, k, wall


@pytest.mark.parametrize("name,options", [("polar_tsto", {"maxiter": 400}), ("low_thrust", {})])
def test_converged_optimum_satisfies_the_oracles_kkt_conditions(name, options, capsys):
    """C3 (``maxiter=400`` per restart) and C4 (the reference's defaults) from their own initial guesses."""
    res, k, wall = _kkt_of_a_solve(name, options, capsys)
    assert abs(res.fun - k["cost"]) <= 1e-9 * max(1.0, abs(k["cost"]))       # the GPU's cost IS the reference path's
    assert k["feasibility"] <= 1e-6 and k["dual"] <= KKT_BOUND and k["complementarity"] <= KKT_BOUND
    assert k["stationarity"] <= KKT_BOUND, k
    assert k["kkt"] <= KKT_BOUND


def test_converged_optimum_of_the_largest_configuration_satisfies_the_oracles_kkt_conditions(capsys):
    """C5 (n = 6148): ~2 600 major iterations from its own guess (profiles/r0*_solve_timing_hip.jsonl), so the test
    starts from a late iterate of that very solve (tests/golden/start_launch4.npz, tools/make_start_launch4.py - an input
    made by this package's solver; the verdict below is the oracle's) with a fresh quasi-Newton matrix."""
    import os
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "start_launch4.npz"))
    res, k, wall = _kkt_of_a_solve("launch4", {"maxiter": 3000}, capsys, start=G["x"])
    assert abs(res.fun - k["cost"]) <= 1e-9 * max(1.0, abs(k["cost"]))
    assert k["feasibility"] <= 1e-6 and k["dual"] <= KKT_BOUND and k["complementarity"] <= KKT_BOUND
    assert k["stationarity"] <= KKT_BOUND, k
    assert res.fun <= float(G["cost_there"]) + 1e-9                            # and it went down from the start
