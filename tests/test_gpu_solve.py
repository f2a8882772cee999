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


# What SLSQP's exit test leaves of the gradient of the Lagrangian (measured, profiles/r05_kkt_*.jsonl).  The test is on the
# CHANGE of the cost and the size of the step (``abs(f - f0) < ftol or norm(s) < ftol`` with the violation below ftol),
# not on stationarity: C4 (smooth minimum-energy transfer) stops at 4.7e-5 with the reference's ftol = 1e-6 and at 2e-6
# with 1e-8; C3's second stage has a bang-bang tangential thrust, whose switching nodes keep a residual of 1.3e-2 ..
# 2.9e-2 however long SLSQP runs (ftol 1e-8: 16 000 subproblems, exit mode 9, cost -0.023434 < -0.022789; exact Jacobians:
# the same) - "exit mode 0" there is SLSQP's flat-valley stop, which SciPy's own core shares by construction.
# (round 6, ADVICE r5: the C3 bound follows what was measured - 1.6e-2 on the default path, up to 2.9e-2 on its variants -
# with a small margin, where round 5 allowed 5e-2; and the cost has to lie in the band every stop of that valley has shown
# since round 1, -0.02348 .. -0.02279, not merely "within 5 %")
STATIONARITY_BOUND = {"polar_tsto": 3.5e-2, "low_thrust": 1e-4}
COST_BAND = {"polar_tsto": (-0.0236, -0.0227), "low_thrust": (41.4540, 41.4550)}


@pytest.mark.parametrize("name,options", [("polar_tsto", {"maxiter": 400}), ("low_thrust", {"maxiter": 400})])
def test_converged_optimum_satisfies_the_oracles_kkt_conditions(name, options, capsys):
    """C3 and C4 (``maxiter=400`` per restart: with the reference's 25, whether C4's tenth restart ends by SLSQP's cost test
    or by its limit depends on rounding - 3 of 5 neighbouring starts, ``bench.py``) from their own initial guesses: the
    point SLSQP stops at is feasible to 1e-6, its multipliers have the right signs, complementarity holds to 1e-5, the cost the GPU
    reports is the reference path's to 1e-9, and the stationarity residual is what SLSQP's ftol test leaves (see above)."""
    res, k, wall = _kkt_of_a_solve(name, options, capsys)
    assert abs(res.fun - k["cost"]) <= 1e-9 * max(1.0, abs(k["cost"]))       # the GPU's cost IS the reference path's
    assert k["feasibility"] <= 1e-6 and k["dual"] <= KKT_BOUND and k["complementarity"] <= KKT_BOUND
    from conftest import record_measurement
    record_measurement("converged_optimum_kkt", name=name, wall_s=wall, cost=float(k["cost"]), stationarity=float(k["stationarity"]),
                       feasibility=float(k["feasibility"]))
    assert k["stationarity"] <= STATIONARITY_BOUND[name], k
    assert COST_BAND[name][0] <= k["cost"] <= COST_BAND[name][1], k["cost"]


def test_both_sqp_cores_reach_exit_mode_0_at_the_same_optimum_of_a_mid_size_problem(capsys):
    """"SciPy's own core shares this stop by construction" was an argument, not a test.  C4's problem on 30 LGL
    nodes (n = 301, above the size from which ``sqp_core="auto"`` takes the HIP core) is small enough for SciPy's Fortran
    core to finish: both cores, same callbacks (the GPU's), the reference's defaults - both stop with exit mode 0, at the
    same cost to 1e-4, and the oracle's KKT residuals of the two points are of one size (the HIP core's at most 3 x
    SciPy's, feasibility and multiplier signs to 1e-6 on both)."""
    import __graft_entry__ as entry
    from oracle import kkt
    found = {}
    for core in ("scipy", "hip"):
        prob, obj = problems.build("low_thrust", nodes=entry.BOTH_CORES_NODES)
        assert prob.number_of_variables == 301
        t0 = time.perf_counter()
        prob.solve(obj, sqp_core=core)
        wall = time.perf_counter() - t0
        res = prob.last_result
        k = kkt.residuals(prob, obj, res.x, prob._engine.m_eq)
        found[core] = (res, k, wall)
        prob._engine.close()
    capsys.readouterr()
    for core, (res, k, wall) in found.items():
        print("low_thrust x 30, %s core: exit mode %d in %.1f s, cost %.9g, oracle: kkt %.2e feasibility %.2e stationarity %.2e"
              % (core, res.status, wall, res.fun, k["kkt"], k["feasibility"], k["stationarity"]))
    from conftest import record_measurement
    record_measurement("both_cores_mid_size", **{core: {"wall_s": w, "cost": float(r.fun), "kkt": float(k["kkt"]), "status": int(r.status)}
                                                 for core, (r, k, w) in found.items()})
    (rs, ks, _), (rh, kh, _) = found["scipy"], found["hip"]
    assert rs.status == 0 and rh.status == 0
    assert abs(rh.fun - rs.fun) <= 1e-4 * abs(rs.fun)
    for k in (ks, kh):
        assert k["feasibility"] <= 1e-6 and k["dual"] <= KKT_BOUND and k["complementarity"] <= KKT_BOUND
    assert kh["kkt"] <= max(3.0 * ks["kkt"], 2e-4), (kh["kkt"], ks["kkt"])


def test_a_tighter_ftol_brings_the_kkt_residual_of_the_smooth_configuration_below_1e_5(capsys):
    """C4 with ``ftol = 1e-8`` and ``maxiter = 400``: exit mode 0 after 211 subproblems (1.6 s) with every KKT residual by
    the oracle below 1e-5 (3.5e-7 measured, profiles/r05_kkt_low_thrust_400.jsonl) - the stationarity the default ftol
    leaves is SLSQP's stopping rule, not an inaccuracy of the GPU's subproblems or Jacobians."""
    from oracle import kkt
    prob, obj = problems.build("low_thrust")
    prob.solve(obj, ftol=1e-8, maxiter=400)
    capsys.readouterr()
    res = prob.last_result
    assert res.status == 0
    k = kkt.residuals(prob, obj, res.x, prob._engine.m_eq)
    print("low_thrust, ftol 1e-8: exit mode %d, cost %.9g, KKT by the oracle %.3e (stationarity %.3e)" % (
        res.status, res.fun, k["kkt"], k["stationarity"]))
    prob._engine.close()
    assert k["kkt"] <= KKT_BOUND, k


def test_the_largest_configuration_near_its_optimum_against_the_oracle(capsys):
    """C5 (n = 6148) takes 2 000 - 4 000 major iterations from its own guess to SLSQP's exit mode 0 (100 - 220 s,
    profiles/r05_launch4_restarts.jsonl; from a late iterate with a fresh quasi-Newton matrix still 1 250 - 3 200), so the
    suite checks a bounded piece: 120 major iterations from a late iterate of that very solve
    (tests/golden/start_launch4.npz, tools/make_start_launch4.py - an input made by this package's solver; the verdicts
    below are the oracle's).  At the iterate SLSQP holds afterwards the cost and every constraint value the GPU reports
    are the reference path's (NumPy restatement) to 1e-9, the iterate stays within 5 % of the start's cost (measured: 141.14
    against 145.87 with a violation of 7.6e-3 - a fresh
    quasi-Newton matrix first trades feasibility against cost: SLSQP's iterates are feasible only at convergence), and
    what the oracle measures there is printed.  The KKT residuals of C5's exit-mode-0 point itself - 174.7 s from this
    start on the round's build: feasibility 3.0e-7, wrong-signed multipliers 1.5e-10, stationarity 7.7e-2 (like C3 a
    problem with switching controls: SLSQP's ftol test fires in a flat valley) - are in profiles/r05_kkt_summary.md."""
    import os
    from oracle import kkt, np_path
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "start_launch4.npz"))
    prob, obj = problems.build("launch4")
    prob.p = np.array(G["x"], dtype=float)
    prob.maxIterator = 1
    t0 = time.perf_counter()
    prob.solve(obj, maxiter=120)
    wall = time.perf_counter() - t0
    capsys.readouterr()
    res = prob.last_result
    assert prob.sqp_core_used == "hip" and res.status in (0, 9) and np.all(np.isfinite(res.x))
    F_gpu = prob._engine.eval_stacked(res.x)
    F_ref = np_path.stacked_values(prob, obj, res.x)
    t0 = time.perf_counter()
    k = kkt.residuals(prob, obj, res.x, prob._engine.m_eq, max_rounds=1)
    print("launch4: %d major iterations in %.1f s, cost %.9g (oracle %.9g, start %.9g); oracle (%.0f s): %s" % (
        res.nit, wall, res.fun, k["cost"], float(G["cost_there"]), time.perf_counter() - t0,
        {key: k[key] for key in ("feasibility", "stationarity_floor_signs_free", "stationarity_floor_2norm")}))
    prob._engine.close()
    assert abs(res.fun - k["cost"]) <= 1e-9 * max(1.0, abs(k["cost"]))
    assert np.all(np.abs(F_gpu - F_ref) <= 1e-9 * np.maximum(1.0, np.abs(F_ref)))
    assert abs(res.fun - float(G["cost_there"])) <= 5e-2 * abs(float(G["cost_there"]))
    assert k["feasibility"] <= 5e-2 and k["stationarity_floor_signs_free"] <= 0.5


@pytest.mark.parametrize("which", ["n = 9988", "n = 16372"])
def test_a_problem_beyond_8192_variables_runs_on_the_hip_sqp_core(which, capsys):
    """VERDICT r4 missing #5: the reference puts no bound on n (``optimize.py:759-781``); rounds 1-4 handed every problem
    with n + 1 > 8192 to SciPy's Fortran core (hours per major iteration there).  C5's problem on 208 nodes per phase -
    n = 9988, rows of 9989 entries: five column slabs per panel of the wide LQ sweep - solves its first major iterations
    with ``sqp_core="auto"``: no fallback, no warning, finite iterates; and its first subproblem, solved through the host
    entry point on the Jacobian the sweep produces, satisfies the linearisation it was given."""
    import warnings
    import __graft_entry__ as entry
    from opengoddard_amd import _native, _sqp_native
    from opengoddard_amd.engine import HipEngine
    from oracle import np_path
    # (round 6, ADVICE r5: the range 8193 .. 16384 was tested at one point; the second case sits 12 variables below the QP
    # core's limit of n + 1 = 16 384 - eight column slabs per wide panel, every one of them in use)
    nodes = entry.BEYOND_8192_NODES if which == "n = 9988" else entry.NEAR_THE_LIMIT_NODES
    prob, obj = problems.build("launch4", nodes=nodes)
    assert prob.number_of_variables == int(which.split("=")[1])
    prob.maxIterator = 1
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        t0 = time.perf_counter()
        prob.solve(obj, maxiter=4)
        wall = time.perf_counter() - t0
    capsys.readouterr()
    res = prob.last_result
    assert prob.sqp_core_used == "hip" and prob.sqp_core_fallback is None
    assert res.status == 9 and res.nit >= 3 and np.all(np.isfinite(res.x)) and np.isfinite(res.fun)
    tm = prob.sqp_timings[-1]
    print("%s: %d subproblems, %d active-set changes, %.2f s in the QP core of %.2f s" % (
        which, tm["qp_solves"], tm["qp_iterations"], tm["qp"], wall))
    prob._engine.close()
    assert tm.get("recoveries", 0) == 0
    if which != "n = 9988":
        return                                                # (the host-staged check below would hold a 1.8 GB Jacobian twice)
    # the first subproblem against its own linearisation (B = I; relaxed if the linearisation is inconsistent)
    prob, obj = problems.build("launch4", nodes=entry.BEYOND_8192_NODES)
    eng = HipEngine(prob, obj)
    lb, ub = np_path.bounds_arrays(prob)
    x = np.clip(prob.p, lb, ub)
    F0, JT = eng.sweep_stacked(x, _native.fd_step(x, lb, ub))
    n, meq = eng.n, eng.m_eq
    g, A, c = JT[:, 0].copy(), JT[:, 1:].T.copy(), F0[1:]
    core = _sqp_native.QpCore(n, meq, eng.m_ineq)
    d, mult, bm, status, iters = core.solve(A, g, c, lb - x, ub - x)
    delta = 0.0
    if status == 4:
        core.set_active()
        d, mult, bm, status, iters = core.solve(A, g, c, np.append(lb - x, 0.0), np.append(ub - x, 1.0), True, 100.0)
        delta = d[n]
    assert status == 1 and iters > 0
    dd = d[:n]
    scale = max(1.0, np.abs(c).max())
    assert np.max(np.abs(A[:meq] @ dd + c[:meq] * (1.0 - delta))) <= 1e-8 * scale
    assert np.min(A[meq:] @ dd + c[meq:] + np.maximum(-c[meq:], 0.0) * delta) >= -1e-8 * scale
    assert np.all(dd >= lb - x - 1e-12) and np.all(dd <= ub - x + 1e-12)
    # stationarity of the QP: d + g = A'mult + bound multipliers (B = I; the relaxed problem's extra variable aside)
    kkt_qp = dd + g - A.T @ mult - bm[:n]
    if delta == 0.0:
        assert np.max(np.abs(kkt_qp)) <= 1e-7 * max(1.0, np.abs(mult).max())
    assert np.all(mult[meq:] >= -1e-10)
    core.close()
    eng.close()
