"""SQP core: the reference's solver is SciPy's Fortran SLSQP.

CPU: the NumPy restatement (``oracle/slsqp_np.py``) is pinned against SciPy itself - its QP
against SciPy on random strictly convex QPs, its major iteration by replaying SciPy's own
iterates (identical inputs at every iteration, so the FD-noise amplification of a free-running
comparison is taken out) and by free-running convergence.

GPU: the HIP QP core (``include/ogsqp.h``) against that restatement on random QPs (including the
relaxed problem, incompatible and rank-deficient cases), its BFGS update, the replay of SciPy's
iterates with the GPU solving every QP, and ``Problem.solve(sqp_core="hip")`` against the SciPy
core.  Tolerances are floating-point ones (the QP solution is unique; the two methods round
differently): 1e-9 relative to the step unless stated."""
import os
import re

import numpy as np
import pytest
from scipy.optimize import minimize

from opengoddard_amd import _native, _sqp_native, codegen, problems
from oracle import np_path, slsqp_np, twin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------- helpers
def random_qp(rng, n, meq, mg, feasible=True):
    Z = rng.normal(size=(n, n)) / np.sqrt(n) + np.eye(n)
    g = rng.normal(size=n)
    C = rng.normal(size=(meq, n))
    xf = rng.normal(size=n) * 0.3
    c = -C @ xf
    G = rng.normal(size=(mg, n))
    h = -G @ xf + rng.uniform(0, 1, mg) * (rng.uniform(size=mg) < 0.7)
    lb = np.where(rng.uniform(size=n) < 0.5, xf - rng.uniform(0, 0.5, n), -np.inf)
    ub = np.where(rng.uniform(size=n) < 0.5, xf + rng.uniform(0, 0.5, n), np.inf)
    if not feasible:
        G = np.vstack([G, -G[:1]])
        h = np.concatenate([h, [-h[0] - 1.0]])          # a'd + h >= 0 and -a'd - h - 1 >= 0
    return Z, g, C, c, G, h, lb, ub


class Callbacks:
    """fun / jac of a registered problem through the compiled CPU twin, logging where SciPy asks
    for the constraint Jacobian (= its accepted iterates)."""

    def __init__(self, name):
        self.prob, self.obj = problems.build(name)
        P = codegen.trace_problem(self.prob, self.obj)
        self.twin = twin.Twin(self.prob, self.obj, program=P)
        self.lb, self.ub = np_path.bounds_arrays(self.prob)
        self.meq = P.m_eq
        self.iterates = []
        self._cache = {}

    def fun(self, x):
        F = self.twin.values(x)
        return F[0], F[1:]

    def jac(self, x, mark=False):
        key = x.tobytes()
        if mark:
            self.iterates.append(x.copy())
        if key not in self._cache:
            h = _native.fd_step(x, self.lb, self.ub)
            _, JT = self.twin.sweep(x, h)
            self._cache = {key: (JT[:, 0].copy(), JT[:, 1:].T.copy())}
        return self._cache[key]

    def scipy(self, maxiter, ftol):
        meq = self.meq
        cons = [{"type": "eq", "fun": lambda x: self.fun(x)[1][:meq], "jac": lambda x: self.jac(x, True)[1][:meq]},
                {"type": "ineq", "fun": lambda x: self.fun(x)[1][meq:], "jac": lambda x: self.jac(x)[1][meq:]}]
        with np.errstate(all="ignore"):
            return minimize(lambda x: self.fun(x)[0], self.prob.p.copy(), jac=lambda x: self.jac(x)[0],
                            bounds=list(zip(self.lb, self.ub)), constraints=cons, method="SLSQP",
                            options={"maxiter": maxiter, "ftol": ftol})


def replay(cb, maxiter, ftol, qp=None):
    """SciPy run, then the restatement forced onto SciPy's iterates -> (scipy result, ours, trace)."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = cb.scipy(maxiter, ftol)
    teacher = list(cb.iterates)
    trace = []
    ours = slsqp_np.slsqp(cb.fun, cb.jac, cb.prob.p.copy(), cb.lb, cb.ub, cb.meq, ftol=ftol, maxiter=maxiter,
                          trace=trace, teacher=teacher, qp=qp)
    return ref, ours, trace


# ------------------------------------------------------------------------------------------- CPU
def test_qp_restatement_matches_scipy_on_random_qps():
    """The QP has a unique solution: SciPy's LSQ chain and the Goldfarb-Idnani restatement must
    agree.  SciPy is run on the QP itself (a QP is solved by SLSQP's first subproblems)."""
    rng = np.random.default_rng(0)
    for _ in range(12):
        n = int(rng.integers(3, 25))
        meq = int(rng.integers(0, n // 2 + 1))
        mg = int(rng.integers(0, 2 * n))
        Z, g, C, c, G, h, lb, ub = random_qp(rng, n, meq, mg)
        B = np.linalg.inv(Z @ Z.T)
        d, lam, mu, mode, Zn, info = slsqp_np.qp_solve(Z, g, C, c, G, h, lb, ub)
        assert mode == 1
        cons = []
        if meq:
            cons.append({"type": "eq", "fun": lambda v: C @ v + c, "jac": lambda v: C})
        if mg:
            cons.append({"type": "ineq", "fun": lambda v: G @ v + h, "jac": lambda v: G})
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = minimize(lambda v: 0.5 * v @ B @ v + g @ v, np.clip(np.zeros(n), lb, ub), jac=lambda v: B @ v + g,
                           bounds=list(zip(lb, ub)), constraints=cons, method="SLSQP",
                           options={"ftol": 1e-15, "maxiter": 500})
        ours = 0.5 * d @ B @ d + g @ d
        assert ours <= ref.fun + 1e-9 * max(1.0, abs(ref.fun))          # at least as optimal ...
        assert np.max(np.abs(d - ref.x)) <= 1e-6                          # ... and the same point
        if meq:
            assert np.max(np.abs(C @ d + c)) <= 1e-10
        if mg:
            assert np.min(G @ d + h) >= -1e-10
        kkt = B @ d + g - C.T @ lam - G.T @ mu - info["bound_multipliers"]
        assert np.max(np.abs(kkt)) <= 1e-7 * max(1.0, np.abs(lam).max(initial=0.0), np.abs(mu).max(initial=0.0))
        assert np.all(mu >= 0)
        assert np.max(np.abs(Zn @ Zn.T - Z @ Z.T)) <= 1e-10 * np.abs(Z @ Z.T).max()   # same B, rotated factor


@pytest.mark.parametrize("name,maxiter,ftol,worst,typical", [
    ("goddard", 40, 1e-10, 1e-9, 1e-11), ("polar_tsto_shipped", 25, 1e-6, 1e-7, 1e-9),
    ("table_ascent", 12, 1e-6, 1e-5, 1e-7)])        # lookup-table problem: ill-conditioned subproblems
def test_slsqp_restatement_replays_scipy_iterates(name, maxiter, ftol, worst, typical):
    """Every accepted iterate of SciPy's SLSQP is reproduced from the previous one: QP step,
    multipliers (through the merit function and the BFGS update), line search, relaxed QP
    (polar_tsto_shipped and table_ascent start from an inconsistent linearisation)."""
    cb = Callbacks(name)
    ref, ours, trace = replay(cb, maxiter, ftol)
    assert (ours["status"], ours["nit"], ours["nfev"], ours["njev"]) == (ref.status, ref.nit, ref.nfev, ref.njev)
    mism = np.array([t["mismatch"] for t in trace if "mismatch" in t])
    step = np.array([t["step"] for t in trace if "mismatch" in t])
    assert len(mism) >= maxiter - 2
    assert np.all(mism <= worst * np.maximum(step, 1e-3)), (mism / np.maximum(step, 1e-3)).max()
    assert np.median(mism / np.maximum(step, 1e-12)) <= typical


def test_slsqp_restatement_converges_like_scipy():
    """Free-running on the brachistochrone (C1): same exit, same number of major iterations,
    function and gradient evaluations, same optimum to ftol."""
    cb = Callbacks("brachistochrone")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = cb.scipy(100, 1e-6)
    ours = slsqp_np.slsqp(cb.fun, cb.jac, cb.prob.p.copy(), cb.lb, cb.ub, cb.meq, ftol=1e-6, maxiter=100)
    assert ref.status == 0 and ours["status"] == 0
    assert (ours["nit"], ours["nfev"], ours["njev"]) == (ref.nit, ref.nfev, ref.njev)
    assert abs(ours["fun"] - ref.fun) <= 1e-6
    assert np.max(np.abs(ours["x"] - ref.x)) <= 1e-3
    assert ours["message"] == ref.message


def test_lapack_lq_and_warm_start_of_the_restatement_change_nothing():
    """``qp_solve(lq="lapack")`` (what makes the restatement usable at the BASELINE sizes) and ``warm=`` (the
    active-set method started from a given set of rows: the previous subproblem's, or arbitrary ones) reach the
    solution of the reflector-by-reflector, cold-started solve; started from the solution's own active set the
    method has nothing to do."""
    rng = np.random.default_rng(5)
    saved = 0
    for trial in range(16):
        n = int(rng.integers(3, 50))
        meq = int(rng.integers(0, n // 2 + 1))
        mg = int(rng.integers(0, 2 * n))
        Z, g, C, c, G, h, lb, ub = random_qp(rng, n, meq, mg)
        base = slsqp_np.qp_solve(Z, g, C, c, G, h, lb, ub)
        fast = slsqp_np.qp_solve(Z, g, C, c, G, h, lb, ub, lq="lapack")
        assert base[3] == fast[3] == 1 and base[5]["ldp_iterations"] == fast[5]["ldp_iterations"]
        assert np.max(np.abs(base[0] - fast[0])) <= 1e-10 * max(1.0, np.abs(base[0]).max())
        same = slsqp_np.qp_solve(Z, g, C, c, G, h, lb, ub, warm=base[5]["active"])
        assert same[5]["ldp_iterations"] == 0 and np.max(np.abs(same[0] - base[0])) <= 1e-10
        g2 = g + 0.2 * rng.normal(size=n)
        cold = slsqp_np.qp_solve(Z, g2, C, c, G, h, lb, ub)
        junk = [("l", 0), ("u", 0), ("u", n - 1)] + [("g", int(j)) for j in rng.permutation(mg)[:4]]
        for warm in (base[5]["active"], junk):
            got = slsqp_np.qp_solve(Z, g2, C, c, G, h, lb, ub, warm=warm)
            assert got[3] == 1
            assert np.max(np.abs(got[0] - cold[0])) <= 1e-9 * max(1.0, np.abs(cold[0]).max())
            assert np.max(np.abs(got[2] - cold[2])) <= 1e-7 * max(1.0, np.abs(cold[2]).max(initial=0.0))
            assert sorted(got[5]["active"]) == sorted(cold[5]["active"])
        saved += cold[5]["ldp_iterations"] - slsqp_np.qp_solve(Z, g2, C, c, G, h, lb, ub,
                                                               warm=base[5]["active"])[5]["ldp_iterations"]
    assert saved > 0                                          # started next to the solution it takes fewer changes


BASELINE_REPLAYS = [("polar_tsto", 1e-4, 1e-5), ("low_thrust", 1e-4, 1e-5)]


def golden_slsqp(name):
    path = os.path.join(ROOT, "tests", "golden", "slsqp_%s.npz" % name)
    return np.load(path)


def replay_golden(name, qp):
    """The restatement forced onto SciPy's iterates at a BASELINE size, SciPy's run taken from the golden file
    (tools/make_golden_slsqp.py: 16-25 s per major iteration of the Fortran core at these sizes)."""
    G = golden_slsqp(name)
    cb = Callbacks(name)
    assert np.array_equal(G["x0"], cb.prob.p) and np.array_equal(G["lb"], cb.lb) and int(G["m_eq"]) == cb.meq
    trace = []
    ours = slsqp_np.slsqp(cb.fun, cb.jac, cb.prob.p.copy(), cb.lb, cb.ub, cb.meq, ftol=float(G["ftol"]),
                          maxiter=int(G["maxiter"]), trace=trace, teacher=list(G["iterates_twin"]), qp=qp)
    return G, ours, trace


@pytest.mark.parametrize("name,worst,typical", BASELINE_REPLAYS[:1])     # (C4 as well on the GPU box: the CPU suite stays short)
def test_slsqp_restatement_replays_scipy_goldens_at_baseline_sizes(name, worst, typical):
    """C3 (n = 1442) and C4 (n = 2001): SciPy 1.15.3's first major iterations - relaxed QP for the inconsistent
    first linearisation, line search, BFGS - reproduced by the restatement, iterate for iterate, with the same
    status / nit / nfev / njev.  The first subproblems at these sizes are vertex solutions (as many active rows as
    the null space has dimensions) and ill-conditioned: a relative perturbation of 1e-12 of the Jacobian moves the
    step by 2e-4 (see the next test), so two exact solvers agree to 1e-7 .. 3e-5 of the step here (measured), not to
    1e-11 as on the small configurations."""
    G, ours, trace = replay_golden(name, lambda *a: slsqp_np.qp_solve(*a, lq="lapack"))
    assert (ours["status"], ours["nit"], ours["nfev"], ours["njev"]) == tuple(
        int(G[k]) for k in ("status_twin", "nit_twin", "nfev_twin", "njev_twin"))
    mism = np.array([t["mismatch"] for t in trace if "mismatch" in t])
    step = np.array([t["step"] for t in trace if "mismatch" in t])
    assert len(mism) >= int(G["maxiter"]) - 2
    assert np.all(mism <= worst * np.maximum(step, 1e-3)), (mism / np.maximum(step, 1e-3)).max()
    assert np.median(mism / np.maximum(step, 1e-12)) <= typical


def test_the_reference_and_the_twin_driven_scipy_runs_agree_as_far_as_conditioning_allows():
    """The golden file also holds the REFERENCE's run (``OpenGoddard.optimize.Problem.solve`` of /root/reference:
    its own callbacks, SciPy differencing them) next to SciPy driven by this repo's callbacks.  Their Jacobians
    differ by forward-difference rounding (1e-5 absolute on entries up to 2.9e3); the first QP at C3 amplifies a
    perturbation of 3e-9 to 2e-4 and of 3e-7 to 3e-2 of the step (measured with the restatement), so the two runs'
    first iterates agree to a few per cent of the step - which is what the goldens show - and drift apart from
    there.  This is a property of the problem, not of either implementation: it is why every other test feeds
    both sides the same iterate."""
    G = golden_slsqp("polar_tsto")
    twin_it, ref_it = G["iterates_twin"], G["iterates_ref"]
    assert np.array_equal(twin_it[0], np.clip(G["x0"], G["lb"], G["ub"]))
    step = np.max(np.abs(twin_it[1] - twin_it[0]))
    assert np.max(np.abs(ref_it[0] - twin_it[1])) <= 0.05 * step
    assert int(G["status_ref"]) == int(G["status_twin"]) == 9 and int(G["nit_ref"]) == int(G["nit_twin"])
    # the amplification, on the restatement: the first subproblem with the Jacobian perturbed by 1e-12 relative
    cb = Callbacks("polar_tsto")
    x = twin_it[0]
    n, meq = x.size, cb.meq
    g, A = cb.jac(x)
    c = cb.fun(x)[1]
    extra = np.concatenate([-c[:meq], np.maximum(-c[meq:], 0.0)])
    Za = np.eye(n + 1)
    Za[n, n] = 1.0 / 100.0
    lo, hi = np.append(cb.lb - x, 0.0), np.append(cb.ub - x, 1.0)

    def relaxed(Amat):
        Aa = np.hstack([Amat, extra[:, None]])
        return slsqp_np.qp_solve(Za, np.append(g, 0.0), Aa[:meq], c[:meq], Aa[meq:], c[meq:], lo, hi, lq="lapack")

    base = relaxed(A)                   # (the plain QP is inconsistent here: mode 4, asserted by the GPU tests)
    noise = 1e-12 * np.abs(A).max() * np.random.default_rng(0).standard_normal(A.shape) * (A != 0)
    moved = relaxed(A + noise)
    assert base[3] == moved[3] == 1
    shift = np.max(np.abs(base[0][:n] - moved[0][:n]))
    assert 1e-6 <= shift <= 1e-2, shift


def test_sqp_library_exports_every_declared_symbol():
    with open(os.path.join(ROOT, "include", "ogsqp.h")) as fh:
        text = re.sub(r"/\*.*?\*/", "", fh.read(), flags=re.S)
    declared = set(re.findall(r"\b(og_[a-z_]+)\s*\(", text))
    assert declared == set(_sqp_native.SIGNATURES), declared ^ set(_sqp_native.SIGNATURES)
    lib = _sqp_native.lib()
    for name in declared:
        assert getattr(lib, name) is not None


def test_sqp_core_fails_loudly_without_a_gpu():
    if _native.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_sqp_native.SqpNativeError, match="no HIP device"):
        _sqp_native.QpCore(5, 2, 3)


# ------------------------------------------------------------------------------------------- GPU
def gpu_qp(core_box):
    """Adapter: the HIP core behind the signature of ``slsqp_np.qp_solve``."""
    def solve(Z, g, C, c, G, h, lb, ub):
        n_all = Z.shape[0]
        meq, mg = C.shape[0], G.shape[0]
        core = core_box.get((meq, mg))
        augmented = core is not None and n_all == core.n + 1
        if core is None:
            core = core_box[(meq, mg)] = _sqp_native.QpCore(n_all, meq, mg)
        n = core.n
        A = np.vstack([C, G])
        cc = np.concatenate([c, h])
        if augmented:
            assert np.allclose(A[:, n], np.concatenate([-c, np.maximum(-h, 0.0)]))
            core.set_factor(Z[:n, :n])
            d, mult, bm, status, iters = core.solve(A[:, :n], g[:n], cc, lb, ub, True, 1.0 / Z[n, n])
            Znew = Z
        else:
            core.set_factor(Z)
            d, mult, bm, status, iters = core.solve(A, g, cc, lb, ub)
            Znew = core.get_factor()
        return d, mult[:meq], mult[meq:], status, Znew, {"ldp_iterations": iters, "bound_multipliers": bm}
    return solve


UPDATES = ["rows", "single", "coop"]      # OGSQP_GI: the row-parallel method in rotated coordinates (default), the older two


@pytest.mark.gpu
@pytest.mark.parametrize("update", UPDATES)
def test_gpu_qp_matches_restatement_on_random_qps(update, monkeypatch):
    """All three implementations of the active-set update: the row-parallel method in rotated coordinates
    (k_rows_decide / k_rows_apply, the default - the restatement's own formulation), one workgroup (k_gi_iter)
    and the cooperative multi-workgroup kernel (k_gi_coop); identical iteration counts."""
    monkeypatch.setenv("OGSQP_GI", update)
    rng = np.random.default_rng(1)
    for trial in range(24):
        n = int(rng.integers(3, 140))
        meq = int(rng.integers(0, n // 2 + 1))
        mg = int(rng.integers(0, 2 * n))
        Z, g, C, c, G, h, lb, ub = random_qp(rng, n, meq, mg)
        d, lam, mu, mode, Zn, info = slsqp_np.qp_solve(Z, g, C, c, G, h, lb, ub)
        core = _sqp_native.QpCore(n, meq, mg)
        core.set_factor(Z)
        dd, mult, bm, status, iters = core.solve(np.vstack([C, G]), g, np.concatenate([c, h]), lb, ub)
        assert (status, iters) == (mode, info["ldp_iterations"]) == (1, iters)
        scale = max(1.0, np.abs(d).max())
        assert np.max(np.abs(dd - d)) <= 1e-11 * scale
        mscale = max(1.0, np.abs(lam).max(initial=0.0), np.abs(mu).max(initial=0.0))
        assert np.abs(mult - np.concatenate([lam, mu])).max(initial=0.0) <= 1e-9 * mscale
        assert np.max(np.abs(bm - info["bound_multipliers"])) <= 1e-9 * mscale
        Zg = core.get_factor()
        assert np.max(np.abs(Zg @ Zg.T - Z @ Z.T)) <= 1e-12 * np.abs(Z @ Z.T).max()
        core.close()


@pytest.mark.gpu
@pytest.mark.parametrize("update", UPDATES)
def test_gpu_qp_relaxed_incompatible_and_singular_cases(update, monkeypatch):
    monkeypatch.setenv("OGSQP_GI", update)
    rng = np.random.default_rng(2)
    n, meq, mg = 30, 8, 25
    Z, g, C, c, G, h, lb, ub = random_qp(rng, n, meq, mg, feasible=False)
    mg = G.shape[0]
    A = np.vstack([C, G])
    cc = np.concatenate([c, h])
    assert slsqp_np.qp_solve(Z, g, C, c, G, h, lb, ub)[3] == 4
    core = _sqp_native.QpCore(n, meq, mg)
    core.set_factor(Z)
    status = core.solve(A, g, cc, lb, ub)[3]
    assert status == _sqp_native.QP_INCOMPATIBLE
    assert np.array_equal(core.get_factor(), Z)                     # a failed solve leaves the factor alone
    # the relaxed problem of slsqp label 140-150 is solvable and agrees with the restatement
    rho = 100.0
    Za = np.zeros((n + 1, n + 1))
    Za[:n, :n] = Z
    Za[n, n] = 1.0 / rho
    extra = np.concatenate([-c, np.maximum(-h, 0.0)])
    Aa = np.hstack([A, extra[:, None]])
    lo, hi = np.append(lb, 0.0), np.append(ub, 1.0)
    d, lam, mu, mode, _, info = slsqp_np.qp_solve(Za, np.append(g, 0.0), Aa[:meq], c, Aa[meq:], h, lo, hi)
    core.set_active()
    dd, mult, bm, status, iters = core.solve(A, g, cc, lo, hi, True, rho)
    assert mode == 1 and status == 1 and iters == info["ldp_iterations"]
    assert 0.0 < dd[n] <= 1.0
    assert np.max(np.abs(dd - d)) <= 1e-10 * max(1.0, np.abs(d).max())
    assert np.max(np.abs(mult - np.concatenate([lam, mu]))) <= 1e-8 * max(1.0, np.abs(mu).max())
    assert np.array_equal(core.get_factor(), Z)                     # relaxed solves never touch the factor
    core.close()
    # an equality row without any gradient: "Singular matrix C in LSQ subproblem"
    C2 = np.vstack([C, np.zeros((1, n))])
    c2 = np.concatenate([c, [0.25]])
    assert slsqp_np.qp_solve(Z, g, C2, c2, G[:5], h[:5], lb, ub)[3] == 6
    core = _sqp_native.QpCore(n, meq + 1, 5)
    core.set_factor(Z)
    assert core.solve(np.vstack([C2, G[:5]]), g, np.concatenate([c2, h[:5]]), lb, ub)[3] == _sqp_native.QP_SINGULAR_C
    core.close()
    # an equality that repeats another one is dropped (zero component, zero multiplier): the step satisfies
    # every constraint, the repeated one included, and costs no less than the QP without the repetition (the
    # one null-space direction the dependency frees is not searched); a contradicting one is "singular C"
    B = np.linalg.inv(Z @ Z.T)
    for rhs_shift, expect in ((0.0, 1), (0.5, 6)):
        C3 = np.vstack([C, 2.0 * C[2:3]])
        c3 = np.concatenate([c, [2.0 * c[2] + rhs_shift]])
        ref = slsqp_np.qp_solve(Z, g, C3, c3, G[:5], h[:5], lb, ub)
        core = _sqp_native.QpCore(n, meq + 1, 5)
        core.set_factor(Z)
        dd, mult, bm, status, _ = core.solve(np.vstack([C3, G[:5]]), g, np.concatenate([c3, h[:5]]), lb, ub)
        assert status == ref[3] == expect
        if expect == 1:
            assert np.max(np.abs(dd - ref[0])) <= 1e-10 * max(1.0, np.abs(ref[0]).max())
            assert np.max(np.abs(C3 @ dd + c3)) <= 1e-10 and np.min(G[:5] @ dd + h[:5]) >= -1e-10
            assert np.all(dd >= lb - 1e-12) and np.all(dd <= ub + 1e-12)
            base = slsqp_np.qp_solve(Z, g, C, c, G[:5], h[:5], lb, ub)[0]
            cost = lambda v: 0.5 * v @ B @ v + g @ v
            assert cost(dd) >= cost(base) - 1e-9 * max(1.0, abs(cost(base)))
            assert mult[meq] == 0.0
        core.close()
    # no inequality at all, and no constraint at all
    for meq0 in (6, 0):
        core = _sqp_native.QpCore(n, meq0, 0)
        core.set_factor(Z)
        d, lam, mu, mode, _, info = slsqp_np.qp_solve(Z, g, C[:meq0], c[:meq0], G[:0], h[:0], lb, ub)
        dd, mult, bm, status, _ = core.solve(C[:meq0], g, c[:meq0], lb, ub)
        assert status == mode == 1 and np.max(np.abs(dd - d)) <= 1e-11 * max(1.0, np.abs(d).max())
        core.close()


@pytest.mark.gpu
@pytest.mark.parametrize("trsv", ["chain", "block"])
def test_gpu_redundant_equality_in_the_blocked_triangular_solves(trsv, monkeypatch):
    """More than 128 equalities: the triangular solves run blocked (64 x 64 inverse blocks; one chained launch, or one
    launch per block).  An equality that repeats an earlier one leaves a vanishing pivot inside a block - that block
    goes pivot by pivot: the repetition is dropped (zero multiplier, every constraint satisfied, the restatement's
    step), a contradicting repetition is "singular matrix C" - with the dependent row in the first, a middle and
    the last block."""
    if trsv == "block":
        monkeypatch.setenv("OGSQP_TRSV", "block")
    rng = np.random.default_rng(31)
    n, meq, mg = 320, 200, 40
    Z, g, C, c, G, h, lb, ub = random_qp(rng, n, meq, mg)
    for where, src in ((30, 3), (100, 70), (199, 150)):
        for shift, expect in ((0.0, 1), (0.5, 6)):
            C3, c3 = C.copy(), c.copy()
            C3[where] = 2.0 * C[src]
            c3[where] = 2.0 * c[src] + shift
            ref = slsqp_np.qp_solve(Z, g, C3, c3, G, h, lb, ub, lq="lapack")
            core = _sqp_native.QpCore(n, meq, mg)
            core.set_factor(Z)
            dd, mult, bm, status, _ = core.solve(np.vstack([C3, G]), g, np.concatenate([c3, h]), lb, ub)
            assert status == ref[3] == expect, (where, shift)
            if expect == 1:
                assert np.max(np.abs(dd - ref[0])) <= 1e-9 * max(1.0, np.abs(ref[0]).max())
                assert np.max(np.abs(C3 @ dd + c3)) <= 1e-9 and np.min(G @ dd + h) >= -1e-9
                assert mult[where] == 0.0
            core.close()


@pytest.mark.gpu
def test_gpu_bfgs_update_matches_restatement():
    rng = np.random.default_rng(3)
    n = 57
    Z = rng.normal(size=(n, n)) / np.sqrt(n) + np.eye(n)
    B = np.linalg.inv(Z @ Z.T)
    core = _sqp_native.QpCore(n, 3, 4)
    for damped in (False, True):
        s = rng.normal(size=n)
        eta = B @ s + (0.1 if not damped else -2.0) * rng.normal(size=n)
        core.set_factor(Z)
        assert core.bfgs(s, eta, B @ s) is False
        want = slsqp_np.bfgs_factor_update(Z, s, eta, B @ s)
        got = core.get_factor()
        assert np.max(np.abs(got - want)) <= 1e-12 * np.abs(want).max()
    core.set_factor(Z)
    assert core.bfgs(np.zeros(n), eta, np.zeros(n)) is True         # undefined update -> caller resets
    assert np.array_equal(core.get_factor(), Z)
    core.reset()
    assert np.array_equal(core.get_factor(), np.eye(n))
    core.close()


@pytest.mark.gpu
@pytest.mark.parametrize("update", UPDATES)
@pytest.mark.parametrize("name,maxiter,ftol", [("goddard", 30, 1e-10), ("polar_tsto_shipped", 20, 1e-6)])
def test_gpu_qp_replays_scipy_iterates(name, maxiter, ftol, update, monkeypatch):
    """SciPy's iterates reproduced with every QP subproblem solved by the HIP core (either
    active-set kernel)."""
    monkeypatch.setenv("OGSQP_GI", update)
    cb = Callbacks(name)
    cores = {}
    ref, ours, trace = replay(cb, maxiter, ftol, qp=gpu_qp(cores))
    for core in cores.values():
        core.close()
    assert (ours["status"], ours["nit"], ours["nfev"], ours["njev"]) == (ref.status, ref.nit, ref.nfev, ref.njev)
    mism = np.array([t["mismatch"] for t in trace if "mismatch" in t])
    step = np.array([t["step"] for t in trace if "mismatch" in t])
    assert np.all(mism <= 1e-6 * np.maximum(step, 1e-3)), (mism / np.maximum(step, 1e-3)).max()
    assert np.median(mism / np.maximum(step, 1e-12)) <= 1e-9


# (round 6, VERDICT r5 #5: the GPU replay's bounds follow what it measures - gpurun_out/test_measurements.jsonl, copied to
# profiles/r06_test_measurements.jsonl: worst mismatch / step 2.9e-5 at C3, 4.2e-11 at C4, 5.7e-5 at C5 - with a margin of
# 3-4 x, where round 5 allowed 1e-4 / 1e-4 / 2e-3; the referee puts either solver at 1e-5 .. 4e-5 of C5's first step)
GPU_REPLAYS = [("polar_tsto", 1e-4, 1e-5), ("low_thrust", 1e-9, 1e-10), ("launch4", 2e-4, 2e-4)]


@pytest.mark.gpu
@pytest.mark.parametrize("name,worst,typical", GPU_REPLAYS)
def test_gpu_qp_replays_scipy_goldens_at_baseline_sizes(name, worst, typical, monkeypatch):
    """C3, C4 and (round 4) C5: SciPy's golden iterates reproduced with every QP subproblem - the relaxed ones included -
    solved by the HIP core in its default configuration (row-parallel active-set method, warm-started from the previous
    subproblem's active rows; at C5 the wide LQ sweep).  Same bounds as the restatement's own replay at C3 / C4: the
    conditioning of these subproblems sets them (test_slsqp_restatement_replays_scipy_goldens_at_baseline_sizes); C5's
    golden holds SciPy's first two major iterations (1.6 hours each for the Fortran core) and its first subproblem is
    the worst conditioned of the three (the referee puts either solver at 1e-5 .. 4e-5 of its step)."""
    if not os.path.exists(os.path.join(ROOT, "tests", "golden", "slsqp_%s.npz" % name)):
        pytest.skip("no SciPy golden for %s in this tree" % name)
    monkeypatch.delenv("OGSQP_GI", raising=False)
    monkeypatch.delenv("OGSQP_WARM", raising=False)
    cores = {}
    G, ours, trace = replay_golden(name, gpu_qp(cores))
    for core in cores.values():
        core.close()
    assert (ours["status"], ours["nit"], ours["nfev"], ours["njev"]) == tuple(
        int(G[k]) for k in ("status_twin", "nit_twin", "nfev_twin", "njev_twin"))
    mism = np.array([t["mismatch"] for t in trace if "mismatch" in t])
    step = np.array([t["step"] for t in trace if "mismatch" in t])
    assert len(mism) >= int(G["maxiter"]) - 2
    from conftest import record_measurement
    record_measurement("gpu_qp_replays_scipy_goldens", name=name, worst_ratio=float((mism / np.maximum(step, 1e-3)).max()),
                       bound=worst, subproblems=int(len(mism)))
    assert np.all(mism <= worst * np.maximum(step, 1e-3)), (mism / np.maximum(step, 1e-3)).max()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["polar_tsto", "low_thrust", "launch4"])
def test_gpu_first_subproblem_of_the_baseline_configurations(name, monkeypatch):
    """C3, C4, C5: the first QP subproblem (B = I) on the Jacobian the sweep kernel produces, default kernels,
    against the restatement (LAPACK LQ): same exit mode (the linearisation is inconsistent at all three: mode 4,
    then the relaxed problem with rho = 100), same number of active-set changes, step and multipliers to the
    accuracy the conditioning of these vertex solutions allows (1e-12 relative noise in A moves the C3 step by
    2e-4: two exact solvers agree to ~1e-7)."""
    from opengoddard_amd.engine import HipEngine
    monkeypatch.delenv("OGSQP_GI", raising=False)
    prob, obj = problems.build(name)
    eng = HipEngine(prob, obj)
    lb, ub = np_path.bounds_arrays(prob)
    x = np.clip(prob.p, lb, ub)
    F0, JT = eng.sweep_stacked(x, _native.fd_step(x, lb, ub))
    n, meq = eng.n, eng.m_eq
    g, A, c = JT[:, 0].copy(), JT[:, 1:].T.copy(), F0[1:]
    ref = slsqp_np.qp_solve(np.eye(n), g, A[:meq], c[:meq], A[meq:], c[meq:], lb - x, ub - x, lq="lapack")
    core = _sqp_native.QpCore(n, meq, eng.m_ineq)
    d, mult, bm, status, iters = core.solve(A, g, c, lb - x, ub - x)
    assert status == ref[3]
    if status != 1:
        assert status == 4
        Za = np.eye(n + 1)
        Za[n, n] = 1.0 / 100.0
        extra = np.concatenate([-c[:meq], np.maximum(-c[meq:], 0.0)])
        Aa = np.hstack([A, extra[:, None]])
        lo, hi = np.append(lb - x, 0.0), np.append(ub - x, 1.0)
        ref = slsqp_np.qp_solve(Za, np.append(g, 0.0), Aa[:meq], c[:meq], Aa[meq:], c[meq:], lo, hi, lq="lapack")
        core.set_active()
        d, mult, bm, status, iters = core.solve(A, g, c, lo, hi, True, 100.0)
        assert status == ref[3] == 1
    assert abs(iters - ref[5]["ldp_iterations"]) <= max(2, ref[5]["ldp_iterations"] // 100), (iters, ref[5]["ldp_iterations"])
    scale = max(1.0, np.abs(ref[0]).max())
    pairwise = np.max(np.abs(d - ref[0])) / scale
    mscale = max(1.0, np.abs(ref[1]).max(initial=0.0), np.abs(ref[2]).max(initial=0.0))
    mult_err = np.max(np.abs(mult - np.concatenate([ref[1], ref[2]]))) / mscale
    # the referee (VERDICT r3 next #5b): which of the two is nearer the exact step?  oracle/qp_referee.py solves the
    # subproblem on the reported active set by iterative refinement with np.longdouble residuals (accurate to
    # cond x 5e-20); both solvers' distances to that step are compared - the HIP core may be at most 10 x as far
    # from it as the restatement with LAPACK's LQ is
    from oracle import qp_referee
    mg = eng.m_ineq
    ref_active = canonical_ids(ref[5]["active"], mg)
    hip_active = sorted(int(v) for v in core.get_active())
    if d.size > n:
        Zr, gr, Ar, lor, hir = Za, np.append(g, 0.0), Aa, lo, hi
    else:
        Zr, gr, Ar, lor, hir = np.eye(n), g, A, lb - x, ub - x
    dist, d_star, rinfo = qp_referee.distances(Zr, gr, Ar, c, lor, hir, meq, ref_active,
                                               {"hip": d, "restatement": ref[0]})
    if hip_active != ref_active:
        # The solution of a strictly convex QP is unique, its active set is not where a multiplier vanishes: at C5 the
        # two solvers may stop with a handful of different (degenerate) rows.  Then each is measured against the
        # refined solution on its OWN set, and the two refined solutions must be one and the same step.
        differing = set(hip_active) ^ set(ref_active)
        assert name == "launch4" and len(differing) <= max(4, len(ref_active) // 200), sorted(differing)
        dist2, d_star2, rinfo2 = qp_referee.distances(Zr, gr, Ar, c, lor, hir, meq, hip_active, {"hip": d})
        same = float(np.max(np.abs(d_star2 - d_star)) / max(1.0, float(np.abs(d_star).max())))
        print("          active sets differ in %d rows; the two refined steps differ by %.3e" % (len(differing), same))
        assert same <= 0.1 * max(dist["hip"], dist["restatement"]) + 1e-9
        assert rinfo2["min_multiplier_of_inequalities"] >= -1e-7
        dist["hip"] = min(dist["hip"], dist2["hip"])
    print("referee %s: hip %.3e restatement %.3e of the step (%d active rows; residuals %s; least multiplier %.3e)" % (
        name, dist["hip"], dist["restatement"], rinfo["active_rows"],
        ", ".join("%.1e" % v for v in rinfo["residual_history"][:4]), rinfo["min_multiplier_of_inequalities"] or 0.0))
    # the refinement has converged: its last sweeps move the step by far less than either solver's distance
    assert max(rinfo["step_moved"][-3:]) <= 1e-3 * min(dist["hip"], dist["restatement"]) + 1e-15, rinfo["step_moved"]
    assert rinfo["min_multiplier_of_inequalities"] is None or rinfo["min_multiplier_of_inequalities"] >= -1e-7
    # the HIP core may be at most twice as far from the exact step as the restatement with LAPACK's LQ (round 5: the
    # wide sweep's T factor is refined and its slice sums compensated; round 4 tolerated 10 x at every size)
    assert dist["hip"] <= 2.0 * max(dist["restatement"], 1e-12), dist
    # the two solvers against each other: no further apart than their distances to the exact step allow
    print("          pairwise step %.3e, multipliers %.3e" % (pairwise, mult_err))
    assert pairwise <= 1.05 * (dist["hip"] + dist["restatement"]) + 1e-12
    if name == "launch4":
        # C5's first subproblem is the worst conditioned of the three (either solver 1e-5 .. 4e-5 from the exact step):
        # absolute bounds from the referee's distances, multipliers (one more solve with the same triangle) within 100 x
        assert dist["hip"] <= 1e-4 and mult_err <= max(1e-4, 100.0 * (dist["hip"] + dist["restatement"]))
    else:
        # C3 / C4: the absolute bounds of rounds 1-3 (a regression up to 1e-4 must not pass here)
        assert pairwise <= 1e-6 and mult_err <= 1e-4
    # what does not depend on the conditioning: the step is feasible for the linearisation and a KKT point
    dd = d[:n]
    delta = d[n] if d.size > n else 0.0
    assert np.max(np.abs(A[:meq] @ dd + c[:meq] * (1.0 - delta))) <= 1e-8 * max(1.0, np.abs(c).max())
    assert np.min(A[meq:] @ dd + c[meq:] + np.maximum(-c[meq:], 0.0) * delta) >= -1e-8 * max(1.0, np.abs(c).max())
    core.close()
    eng.close()


def canonical_ids(active, mg):
    """``info["active"]`` of the restatement -> the numbering of og_qp_get_active."""
    return sorted(j if kind == "g" else mg + 2 * j + (1 if kind == "u" else 0) for kind, j in active)


@pytest.mark.gpu
def test_gpu_warm_started_active_set_reaches_the_same_solution():
    """The default active-set method keeps the rows active at a solution and starts the next solve on the handle
    from them (rows appended to the LQ sweep, negative multipliers dropped): a perturbed problem solved warm equals
    the restatement's cold solve, takes fewer changes than its own cold solve, and reports the solution's active
    set; an unchanged problem takes no change at all; set_active() / arbitrary rows work as well."""
    import os
    if os.environ.get("OGSQP_WARM") == "0" or os.environ.get("OGSQP_GI") in ("single", "coop", "old"):
        pytest.skip("the warm start is switched off in this environment")
    rng = np.random.default_rng(11)
    saved = 0
    for trial in range(12):
        n = int(rng.integers(8, 120))
        meq = int(rng.integers(0, n // 2 + 1))
        mg = int(rng.integers(1, 2 * n))
        Z, g, C, c, G, h, lb, ub = random_qp(rng, n, meq, mg)
        A, cc = np.vstack([C, G]), np.concatenate([c, h])
        first = slsqp_np.qp_solve(Z, g, C, c, G, h, lb, ub)
        core = _sqp_native.QpCore(n, meq, mg)
        core.set_factor(Z)
        d, mult, bm, status, iters = core.solve(A, g, cc, lb, ub)
        assert status == 1 and iters == first[5]["ldp_iterations"]
        assert sorted(core.get_active().tolist()) == canonical_ids(first[5]["active"], mg)
        core.set_factor(Z)                                   # the solve left Z Q: the same B, start from Z again
        d2, _, _, status, iters = core.solve(A, g, cc, lb, ub)
        assert status == 1 and iters == 0 and np.max(np.abs(d2 - d)) <= 1e-11 * max(1.0, np.abs(d).max())
        g2 = g + 0.2 * rng.normal(size=n)
        cold = slsqp_np.qp_solve(Z, g2, C, c, G, h, lb, ub)
        warm_ref = slsqp_np.qp_solve(Z, g2, C, c, G, h, lb, ub, warm=first[5]["active"])
        core.set_factor(Z)
        d3, mult3, bm3, status, iters3 = core.solve(A, g2, cc, lb, ub)
        assert status == cold[3] == 1
        assert np.max(np.abs(d3 - cold[0])) <= 1e-10 * max(1.0, np.abs(cold[0]).max())
        mscale = max(1.0, np.abs(cold[1]).max(initial=0.0), np.abs(cold[2]).max(initial=0.0))
        assert np.max(np.abs(mult3 - np.concatenate([cold[1], cold[2]]))) <= 1e-8 * mscale
        assert sorted(core.get_active().tolist()) == canonical_ids(cold[5]["active"], mg)
        assert abs(iters3 - warm_ref[5]["ldp_iterations"]) <= 2
        saved += cold[5]["ldp_iterations"] - iters3
        junk = rng.permutation(mg + 2 * n)[:min(6, n - meq)].astype(np.int32)
        core.set_active(junk)
        core.set_factor(Z)
        d4, _, _, status, _ = core.solve(A, g2, cc, lb, ub)
        assert status == 1 and np.max(np.abs(d4 - cold[0])) <= 1e-10 * max(1.0, np.abs(cold[0]).max())
        core.close()
    assert saved > 0


@pytest.mark.gpu
@pytest.mark.parametrize("n", [150, 300, 600, 860, 1100, 1380, 1530, 1800])
def test_gpu_lq_panel_at_every_row_length(n):
    """The 16-reflector panel holds its rows in groups of 256 columns (E = 1 .. 8 groups: a row pair per wavefront and no
    barrier between reflectors up to E = 6, the lane-group form with two barriers beyond): one subproblem per E whose
    sweep starts in that class and walks down through the shorter ones, with a number of equalities that is not a
    multiple of 16 (a last panel of 5 reflectors), an equality that repeats an earlier one inside a panel (no reflector
    is built from it: zero multiplier) - against the restatement's step with LAPACK's LQ."""
    rng = np.random.default_rng(1000 + n)
    meq, mg = 16 * (n // 40) + 5, 12
    Z, g, C, c, G, h, lb, ub = random_qp(rng, n, meq, mg)
    lb[:], ub[:] = -np.inf, np.inf                       # (the sweep is what is tested: a short active-set phase)
    where, src = meq // 2 + 3, meq // 2 - 9             # (both in the same panel or in neighbouring ones)
    C[where], c[where] = -1.5 * C[src], -1.5 * c[src]
    ref = slsqp_np.qp_solve(Z, g, C, c, G, h, lb, ub, lq="lapack")
    core = _sqp_native.QpCore(n, meq, mg)
    core.set_factor(Z)
    dd, mult, bm, status, iters = core.solve(np.vstack([C, G]), g, np.concatenate([c, h]), lb, ub)
    assert status == ref[3] == 1
    assert np.max(np.abs(dd - ref[0])) <= 1e-9 * max(1.0, np.abs(ref[0]).max())
    assert np.max(np.abs(C @ dd + c)) <= 1e-9 * max(1.0, np.abs(c).max()) and np.min(G @ dd + h) >= -1e-9
    assert mult[where] == 0.0
    # the same subproblem again on the same handle: the same bits (nothing in the panel depends on timing)
    core.set_factor(Z)
    core.set_active()
    d2, _, _, status2, iters2 = core.solve(np.vstack([C, G]), g, np.concatenate([c, h]), lb, ub)
    assert status2 == 1 and iters2 == iters and np.array_equal(d2, dd)
    core.close()


@pytest.mark.gpu
def test_gpu_lq_sweep_forms_agree(monkeypatch):
    """(Also the three forms of the blocked triangular solves, OGSQP_TRSV.)  The three forms of the LQ sweep - 16-reflector panels with the next panel factored during the trailing update
    (default: k_lq_step16, head workgroups + panel workgroup inside one launch), the same panels as separate
    launches (OGSQP_LQ=16) and round 2's 8-reflector panels (OGSQP_LQ=8) - are the same Householder sweep with sums
    in different orders: the same step to rounding, the same active set and number of changes, on random problems
    with several panels and on the baseline configuration's first subproblem shape (n = 500, 300 equalities)."""
    rng = np.random.default_rng(23)
    cases = [(int(rng.integers(40, 130)), None, None) for _ in range(6)] + [(500, 300, 240)]
    for n, meq, mg in cases:
        meq = int(rng.integers(n // 3, 2 * n // 3)) if meq is None else meq
        mg = int(rng.integers(1, n)) if mg is None else mg
        Z, g, C, c, G, h, lb, ub = random_qp(rng, n, meq, mg)
        A, cc = np.vstack([C, G]), np.concatenate([c, h])
        results = {}
        for form in ("ahead", "16", "8", "trsv-block", "trsv-single"):
            monkeypatch.delenv("OGSQP_LQ", raising=False)
            monkeypatch.delenv("OGSQP_TRSV", raising=False)
            if form in ("16", "8"):
                monkeypatch.setenv("OGSQP_LQ", form)
            elif form.startswith("trsv-"):               # the triangular solves: one chained launch (default) / per block / one workgroup
                monkeypatch.setenv("OGSQP_TRSV", form[5:])
            core = _sqp_native.QpCore(n, meq, mg)            # the switch is read when the handle is made
            core.set_factor(Z)
            d, mult, bm, status, iters = core.solve(A, g, cc, lb, ub)
            results[form] = (d, mult, status, iters, sorted(core.get_active().tolist()))
            # the same handle again: the counters of the look-ahead start over with every sweep
            core.set_factor(Z)
            core.set_active()
            d2, _, _, status2, iters2 = core.solve(A, g, cc, lb, ub)
            assert status2 == status and iters2 == iters and np.array_equal(d2, d)
            core.close()
        d0, m0, s0, i0, a0 = results["ahead"]
        assert s0 == 1
        for form in ("16", "8", "trsv-block", "trsv-single"):
            d, mult, status, iters, active = results[form]
            assert (status, iters, active) == (s0, i0, a0), (n, meq, mg, form)
            assert np.max(np.abs(d - d0)) <= 1e-11 * max(1.0, np.abs(d0).max())
            assert np.max(np.abs(mult - m0)) <= 1e-9 * max(1.0, np.abs(m0).max())


@pytest.mark.gpu
def test_gpu_wide_sweep_agrees_with_the_old_kernels(monkeypatch):
    """Rows longer than the 2048 entries one workgroup's registers hold (C5: 6149) take the wide sweep since round 4
    (csrc/ogsqp_lqwide.h: a 16-reflector panel whose columns are split over 2-4 workgroups that exchange their partial
    products through HBM, 64-reflector block reflectors applied by GEMMs); ``OGSQP_WIDE=0`` keeps round 2's
    8-reflector kernels on those rows.  Same Householder sweep, sums in another order: same status, active set and
    number of changes, the step to rounding - on random problems whose rows start at 2 and at 3 slabs (the sweep
    hands over to the look-ahead kernels where the rows have become short enough; the second case ends inside the wide
    part, with a partial last block and a partial last panel), with a redundant equality row in a wide panel, and twice
    on one handle."""
    rng = np.random.default_rng(31)
    for n, meq, mg, redundant in ((2300, 700, 150, False), (4200, 1930, 120, True)):
        Z, g, C, c, G, h, lb, ub = random_qp(rng, n, meq, mg)
        if redundant:                                   # row 100 repeats a combination of rows 3 and 40: dropped, consistent
            C[100] = 0.5 * C[3] - 2.0 * C[40]
            c[100] = 0.5 * c[3] - 2.0 * c[40]
        A, cc = np.vstack([C, G]), np.concatenate([c, h])
        results = {}
        for form in ("wide", "old"):
            monkeypatch.delenv("OGSQP_WIDE", raising=False)
            if form == "old":
                monkeypatch.setenv("OGSQP_WIDE", "0")
            core = _sqp_native.QpCore(n, meq, mg)
            core.set_factor(Z)
            d, mult, bm, status, iters = core.solve(A, g, cc, lb, ub)
            results[form] = (d, mult, status, iters, sorted(core.get_active().tolist()))
            core.set_factor(Z)
            core.set_active()
            d2, _, _, status2, iters2 = core.solve(A, g, cc, lb, ub)
            assert status2 == status and iters2 == iters and np.array_equal(d2, d)      # bit-reproducible
            assert core.recoveries() == 0
            core.close()
        d0, m0, s0, i0, a0 = results["old"]
        d1, m1, s1, i1, a1 = results["wide"]
        assert s0 == s1 == 1 and (i0, a0) == (i1, a1), (n, i0, i1)
        assert np.max(np.abs(d1 - d0)) <= 1e-10 * max(1.0, np.abs(d0).max())
        assert np.max(np.abs(m1 - m0)) <= 1e-8 * max(1.0, np.abs(m0).max())
        # feasibility and stationarity do not care which kernels ran
        assert np.max(np.abs(C @ d1 + c)) <= 1e-9 * max(1.0, np.abs(c).max())


@pytest.mark.gpu
def test_gpu_wide_sweep_on_side_streams_gives_the_bits_of_the_one_stream_order(monkeypatch):
    """The block reflectors of the wide sweep are applied to the rest of C Z and to Z on two streams of the handle's own
    while the caller's stream factors the next block (events order what depends on what); ``OGSQP_WIDE_AHEAD=0`` runs
    the same kernels on the same pieces one after the other on the caller's stream.  Disjoint rows, same arithmetic:
    the step, the multipliers and the change count must be the same BITS - a missing dependency between the streams
    would show here (several subproblems per handle, so that consecutive sweeps overlap too)."""
    rng = np.random.default_rng(77)
    n, meq, mg = 4300, 2000, 100
    Z, g, C, c, G, h, lb, ub = random_qp(rng, n, meq, mg)
    A, cc = np.vstack([C, G]), np.concatenate([c, h])
    results = {}
    for form in ("ahead", "serial"):
        monkeypatch.delenv("OGSQP_WIDE_AHEAD", raising=False)
        if form == "serial":
            monkeypatch.setenv("OGSQP_WIDE_AHEAD", "0")
        core = _sqp_native.QpCore(n, meq, mg)
        out = []
        for rep in range(3):
            core.set_factor(Z * (1.0 + 0.25 * rep))
            core.set_active()
            d, mult, bm, status, iters = core.solve(A, g, cc, lb, ub)
            out.append((d.copy(), mult.copy(), status, iters))
        assert core.recoveries() == 0
        core.close()
        results[form] = out
    for (d0, m0, s0, i0), (d1, m1, s1, i1) in zip(results["serial"], results["ahead"]):
        assert s0 == s1 == 1 and i0 == i1
        assert np.array_equal(d0, d1) and np.array_equal(m0, m1)


@pytest.mark.gpu
@pytest.mark.parametrize("n,meq,mg", [(420, 120, 90), (900, 200, 120), (2400, 300, 60)])
def test_gpu_warm_start_removals_spread_over_the_grid_give_the_bits_of_one_workgroup(n, meq, mg, monkeypatch):
    """Round 5: a removal of the warm start recomputes the point and its multipliers on the remaining rows (two products
    with the q x q inverse).  One workgroup did both (52-90 us at C3, a millisecond at C5); now the grid of
    ``k_rows_decide`` does: blocks of 16 columns of the first product per workgroup, a counter-and-wait, the rows of the
    second dealt to all wavefronts, the last workgroup at the ticket decides.  Same sums in the same order - the BITS of
    ``OGSQP_WARM_SPREAD=0`` - on subproblems whose warm start has rows to remove (the cost vector is turned between
    repetitions, so that the previous active set is partly wrong)."""
    rng = np.random.default_rng(3 * n)
    Z, g, C, c, G, h, lb, ub = random_qp(rng, n, meq, mg)
    A, cc = np.vstack([C, G]), np.concatenate([c, h])
    turns = [g, g[::-1].copy(), -g, 0.5 * g + 0.5 * g[::-1]]
    results = {}
    for form in ("spread", "one"):
        monkeypatch.delenv("OGSQP_WARM_SPREAD", raising=False)
        if form == "one":
            monkeypatch.setenv("OGSQP_WARM_SPREAD", "0")
        core = _sqp_native.QpCore(n, meq, mg)
        out = []
        for rep, gv in enumerate(turns):
            core.set_factor(Z * (1.0 + 0.1 * rep))
            d, mult, bm, status, iters = core.solve(A, gv, cc, lb, ub)
            out.append((d.copy(), mult.copy(), bm.copy(), status, iters, sorted(int(v) for v in core.get_active())))
        assert core.recoveries() == 0
        core.close()
        results[form] = out
    monkeypatch.delenv("OGSQP_WARM_SPREAD", raising=False)
    for (d0, m0, b0, s0, i0, a0), (d1, m1, b1, s1, i1, a1) in zip(results["one"], results["spread"]):
        assert s0 == s1 == 1 and i0 == i1 and a0 == a1, (s0, s1, i0, i1)
        assert np.array_equal(d0, d1) and np.array_equal(m0, m1) and np.array_equal(b0, b1)


@pytest.mark.gpu
@pytest.mark.parametrize("n,meq,mg", [(420, 120, 90), (900, 200, 120)])
def test_gpu_short_row_forms_give_the_same_bits(n, meq, mg, monkeypatch):
    """Rows of up to 1024 null-space coordinates have two forms of the pass over the rows: rounds 3-4's register kernel
    (``k_rows_apply_r4``: the default up to 512 coordinates, where it measured faster) and round 5's (``k_rows_apply``: the
    vector in LDS, the row requested with the first round trip; the default from 513 to 1024).  ``OGSQP_ROWS=r4`` / ``lds``
    force one of them: step, multipliers, active set and change count must be the same bits (300 and 700 coordinates:
    both template sizes; cold and warm-started subproblems)."""
    rng = np.random.default_rng(n)
    Z, g, C, c, G, h, lb, ub = random_qp(rng, n, meq, mg)
    A, cc = np.vstack([C, G]), np.concatenate([c, h])
    results = {}
    for form in ("r4", "lds"):
        monkeypatch.setenv("OGSQP_ROWS", form)
        core = _sqp_native.QpCore(n, meq, mg)
        out = []
        for rep in range(3):
            core.set_factor(Z * (1.0 + 0.25 * rep))
            d, mult, bm, status, iters = core.solve(A, g * (1.0 + 0.1 * rep), cc, lb, ub)
            out.append((d.copy(), mult.copy(), bm.copy(), status, iters, sorted(int(v) for v in core.get_active())))
        core.close()
        results[form] = out
    monkeypatch.delenv("OGSQP_ROWS")
    assert results["r4"][0][4] > 10
    for (d0, m0, b0, s0, i0, a0), (d1, m1, b1, s1, i1, a1) in zip(results["r4"], results["lds"]):
        assert s0 == s1 == 1 and i0 == i1 and a0 == a1
        assert np.array_equal(d0, d1) and np.array_equal(m0, m1) and np.array_equal(b0, b1)


@pytest.mark.gpu
@pytest.mark.parametrize("n,meq,mg", [(4300, 2000, 100), (2500, 470, 40)])
def test_gpu_in_block_update_in_one_launch_gives_the_bits_of_the_three_launches(n, meq, mg, monkeypatch):
    """Round 5 (opt-in, ``OGSQP_WIDE_INBLOCK=1``; measured 1 % of a C5 subproblem, so the three launches stay the
    default): inside a 64-reflector block of the wide sweep the later panels' rows get each panel's reflectors from ONE
    launch (``k_wy_inblock``: the column slices' workgroups exchange their shares of the products through memory and a
    counter) instead of three (product over slices, finish, rank-16 update).  Same slices, same sums, same MFMA chains:
    the BITS must be equal - a race in the exchange would show as a difference (three subproblems per handle; the second
    shape has a partial last block and a partial last panel: 470 = 7 x 64 + 22 equalities)."""
    rng = np.random.default_rng(n + 1)
    Z, g, C, c, G, h, lb, ub = random_qp(rng, n, meq, mg)
    A, cc = np.vstack([C, G]), np.concatenate([c, h])
    results = {}
    for form in ("one", "three"):
        monkeypatch.setenv("OGSQP_WIDE_INBLOCK", "1" if form == "one" else "0")
        core = _sqp_native.QpCore(n, meq, mg)
        out = []
        for rep in range(3):
            core.set_factor(Z * (1.0 + 0.25 * rep))
            core.set_active()
            d, mult, bm, status, iters = core.solve(A, g, cc, lb, ub)
            out.append((d.copy(), mult.copy(), status, iters))
        assert core.recoveries() == 0
        core.close()
        results[form] = out
    monkeypatch.delenv("OGSQP_WIDE_INBLOCK", raising=False)
    for (d0, m0, s0, i0), (d1, m1, s1, i1) in zip(results["three"], results["one"]):
        assert s0 == s1 == 1 and i0 == i1
        assert np.array_equal(d0, d1) and np.array_equal(m0, m1)


@pytest.mark.gpu
@pytest.mark.parametrize("n,meq,mg", [(1500, 300, 200), (2400, 300, 60)])
def test_gpu_streamed_rows_give_the_bits_of_the_register_kernels(n, meq, mg, monkeypatch):
    """Round 5: rows of more than 1024 null-space coordinates are STREAMED by ``k_rows_apply_stream`` (run-time strips of
    64 coordinates, the second pass out of LDS or the caches) instead of held in registers by ``k_rows_apply<TAIL>``
    (one wavefront per SIMD at C5).  Same sums in the same order: step, multipliers, active set and change count must be
    the same BITS in all three forms - cold start, a warm-started second subproblem (phase -1 removals, the kind-3 / kind-4
    passes) and a third one."""
    rng = np.random.default_rng(n)
    Z, g, C, c, G, h, lb, ub = random_qp(rng, n, meq, mg)
    A, cc = np.vstack([C, G]), np.concatenate([c, h])
    results = {}
    for form in ("reg", "nostage", "stage"):
        monkeypatch.setenv("OGSQP_ROWS", form)
        core = _sqp_native.QpCore(n, meq, mg)
        out = []
        for rep in range(3):
            core.set_factor(Z * (1.0 + 0.25 * rep))
            d, mult, bm, status, iters = core.solve(A, g * (1.0 + 0.1 * rep), cc, lb, ub)
            out.append((d.copy(), mult.copy(), bm.copy(), status, iters, sorted(int(v) for v in core.get_active())))
        core.close()
        results[form] = out
    monkeypatch.delenv("OGSQP_ROWS")
    assert results["reg"][0][4] > 20                                  # the active-set loop did run
    for form in ("nostage", "stage"):
        for (d0, m0, b0, s0, i0, a0), (d1, m1, b1, s1, i1, a1) in zip(results["reg"], results[form]):
            assert s0 == s1 == 1 and i0 == i1 and a0 == a1, (form, s0, s1, i0, i1)
            assert np.array_equal(d0, d1) and np.array_equal(m0, m1) and np.array_equal(b0, b1), form


@pytest.mark.gpu
@pytest.mark.parametrize("n,meq,mg", [(420, 120, 90), (900, 200, 120), (1500, 700, 260)])
def test_gpu_resident_active_set_gives_the_bits_of_the_two_launch_form(n, meq, mg, monkeypatch):
    """Round 6: where the rows of W and of the inverse fit the chip's LDS the whole active-set loop is ONE launch
    (``k_rows_resident``: a wavefront per row, the shared state replicated in every workgroup, four self-validating
    messages per change) instead of ``k_rows_decide`` + ``k_rows_apply`` per change.  Same arithmetic in the same order:
    step, multipliers, active set and change count are the same BITS - cold start, a warm-started second subproblem
    (the warm start's removals run as two-launch pairs in front of the resident launch) and a third one."""
    rng = np.random.default_rng(n + 7)
    Z, g, C, c, G, h, lb, ub = random_qp(rng, n, meq, mg)
    A, cc = np.vstack([C, G]), np.concatenate([c, h])
    results, stats = {}, {}
    for form in ("0", "1"):
        monkeypatch.setenv("OGSQP_RESIDENT", form)
        core = _sqp_native.QpCore(n, meq, mg)
        out = []
        for rep in range(3):
            core.set_factor(Z * (1.0 + 0.25 * rep))
            d, mult, bm, status, iters = core.solve(A, g * (1.0 + 0.1 * rep), cc, lb, ub)
            out.append((d.copy(), mult.copy(), bm.copy(), status, iters, sorted(int(v) for v in core.get_active())))
        stats[form] = core.resident_stats()
        assert core.recoveries() == 0
        core.close()
        results[form] = out
    monkeypatch.delenv("OGSQP_RESIDENT")
    assert stats["0"] == (0, 0)
    launches, changes = stats["1"]
    assert launches >= 3 and changes > 20
    assert results["0"][0][4] > 20                                    # the active-set loop did run
    for (d0, m0, b0, s0, i0, a0), (d1, m1, b1, s1, i1, a1) in zip(results["0"], results["1"]):
        assert s0 == s1 == 1 and i0 == i1 and a0 == a1, (s0, s1, i0, i1)
        assert np.array_equal(d0, d1) and np.array_equal(m0, m1) and np.array_equal(b0, b1)


@pytest.mark.gpu
def test_device_resident_jacobian_equals_host_staged():
    """og_qp_solve_dev on the Jacobian the sweep kernel left in HBM == og_qp_solve on its host copy;
    og_jt_times gives the cost gradient and the gradient of the Lagrangian."""
    import torch
    from opengoddard_amd.engine import HipEngine
    from opengoddard_amd.sqp import DeviceJacobian
    prob, obj = problems.build("goddard")
    eng = HipEngine(prob, obj)
    lb, ub = np_path.bounds_arrays(prob)
    x = np.clip(prob.p, lb, ub)
    dj = DeviceJacobian(eng)
    F = dj.sweep(x, lb, ub)
    JT = dj.d_JT.cpu().numpy().reshape(eng.n, eng.m)
    F0, JT_host = eng.sweep_stacked(x, _native.fd_step(x, lb, ub))
    assert np.array_equal(F, F0) and np.array_equal(JT, JT_host)
    core = _sqp_native.QpCore(eng.n, eng.m_eq, eng.m_ineq)
    g = JT[:, 0].copy()
    unit = np.zeros(eng.m)
    unit[0] = 1.0
    assert np.array_equal(core.jt_times(dj.ptr, dj.ld, unit, dj.stream), g)
    r = np.random.default_rng(4).normal(size=eng.m - 1)
    v = core.jt_times(dj.ptr, dj.ld, np.concatenate([[1.0], -r]), dj.stream)
    assert np.max(np.abs(v - (g - JT[:, 1:] @ r))) <= 1e-10 * np.abs(JT).max() * np.abs(r).max()
    a = core.solve_dev(dj.ptr, dj.ld, g, F[1:], lb - x, ub - x, False, 100.0, dj.stream)
    core.reset()
    core.set_active()                                                # the same start: the empty active set
    b = core.solve(JT[:, 1:].T.copy(), g, F[1:], lb - x, ub - x)
    assert a[3] == b[3] == 1 and a[4] == b[4]
    for u, w in zip(a[:3], b[:3]):
        assert np.array_equal(u, w)                                  # same kernels, same order: same bits
    core.close()
    eng.close()
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_solve_with_hip_core_converges_like_scipy_core(capsys):
    """Problem.solve: same exit mode, iteration and evaluation counts and optimum with the QP
    subproblems on the GPU as with SciPy's Fortran core (C1; ftol 1e-6)."""
    results = {}
    for core in ("scipy", "hip"):
        prob, obj = problems.build("brachistochrone")
        prob.maxIterator = 1
        prob.solve(obj, maxiter=100, ftol=1e-6, sqp_core=core)
        results[core] = (prob.last_result, prob.p.copy())
    out = capsys.readouterr().out
    assert out.count("Optimization terminated successfully") >= 2
    ref, ours = results["scipy"][0], results["hip"][0]
    assert ref.status == ours.status == 0
    assert (ours.nit, ours.nfev, ours.njev) == (ref.nit, ref.nfev, ref.njev)
    assert abs(ours.fun - ref.fun) <= 1e-6
    assert np.max(np.abs(results["hip"][1] - results["scipy"][1])) <= 1e-3
    assert abs(ours.fun - 1.7724562) <= 2e-5                        # the golden optimum of example 01


@pytest.mark.gpu
def test_goddard_converges_with_both_cores(capsys):
    """C2 with the FD cost gradient taken from the device-resident Jacobian (og_jt_times): both cores
    stop with exit mode 0 at the same optimum (ftol 1e-10; the paths differ after a few iterations
    because the FD Jacobian amplifies rounding differences, the optimum does not)."""
    found = {}
    for core in ("scipy", "hip"):
        prob, obj = problems.build("goddard")
        prob.maxIterator = 1
        prob.solve(obj, maxiter=600, ftol=1e-10, sqp_core=core)
        found[core] = prob.last_result
    capsys.readouterr()
    assert found["scipy"].status == 0 and found["hip"].status == 0
    assert abs(found["hip"].fun - found["scipy"].fun) <= 5e-6
    assert abs(found["hip"].fun + 1.01283) <= 2e-5                   # final altitude 1.01283 (example 04)
    timing = found["hip"].timing
    assert timing["qp_solves"] >= found["hip"].nit - 1 and timing["qp"] > 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("update", UPDATES)
@pytest.mark.parametrize("name,tol", [("brachistochrone", 1e-5),      # v(0) = 0: cond(C) = 3e10 at the initial guess
                                      ("goddard", 1e-8), ("polar_tsto_shipped", 1e-8),
                                      ("low_thrust_shipped", 1e-8), ("table_ascent", 1e-8)])
def test_gpu_first_subproblem_of_every_small_configuration(name, tol, update, monkeypatch):
    """The first QP subproblem (B = I) of each small configuration, on the Jacobian the sweep kernel
    produces: same exit mode as the restatement; same step and multipliers where it is solvable, the
    relaxed problem (rho = 100) where the linearisation is inconsistent.  These Jacobians contain
    what random matrices do not: inequality rows that repeat equalities, zero rows, bounds that
    coincide with path constraints.  Both active-set updates."""
    from opengoddard_amd.engine import HipEngine
    monkeypatch.setenv("OGSQP_GI", update)
    prob, obj = problems.build(name)
    eng = HipEngine(prob, obj)
    lb, ub = np_path.bounds_arrays(prob)
    x = np.clip(prob.p, lb, ub)
    F0, JT = eng.sweep_stacked(x, _native.fd_step(x, lb, ub))
    n, meq = eng.n, eng.m_eq
    g, A, c = JT[:, 0].copy(), JT[:, 1:].T.copy(), F0[1:]
    ref = slsqp_np.qp_solve(np.eye(n), g, A[:meq], c[:meq], A[meq:], c[meq:], lb - x, ub - x)
    core = _sqp_native.QpCore(n, meq, eng.m_ineq)
    d, mult, bm, status, iters = core.solve(A, g, c, lb - x, ub - x)
    assert status == ref[3]
    if status != 1:
        assert status == 4
        Za = np.eye(n + 1)
        Za[n, n] = 1.0 / 100.0
        extra = np.concatenate([-c[:meq], np.maximum(-c[meq:], 0.0)])
        Aa = np.hstack([A, extra[:, None]])
        lo, hi = np.append(lb - x, 0.0), np.append(ub - x, 1.0)
        ref = slsqp_np.qp_solve(Za, np.append(g, 0.0), Aa[:meq], c[:meq], Aa[meq:], c[meq:], lo, hi)
        d, mult, bm, status, iters = core.solve(A, g, c, lo, hi, True, 100.0)
        assert status == ref[3] == 1
    scale = max(1.0, np.abs(ref[0]).max())
    assert np.max(np.abs(d - ref[0])) <= tol * scale
    mscale = max(1.0, np.abs(ref[1]).max(initial=0.0), np.abs(ref[2]).max(initial=0.0))
    assert np.max(np.abs(mult - np.concatenate([ref[1], ref[2]]))) <= 100 * tol * mscale
    core.close()
    eng.close()


@pytest.mark.gpu
def test_environment_variable_selects_the_hip_core(monkeypatch, capsys):
    """OG_SQP_CORE=hip: an unmodified script (no sqp_core argument) gets the GPU SQP core."""
    monkeypatch.setenv("OG_SQP_CORE", "hip")
    prob, obj = problems.build("brachistochrone")
    prob.solve(obj)                                  # reference defaults: maxiter 25, restarts until converged
    out = capsys.readouterr().out
    assert "Optimization terminated successfully" in out and "---- iteration : 1 ----" in out
    assert prob.sqp_timings and prob.last_result.status == 0
    assert abs(prob.last_result.fun - 1.7724562) <= 2e-5
    with pytest.raises(ValueError, match="sqp_core"):
        prob.solve(obj, sqp_core="fortran")


def test_unknown_sqp_core_is_rejected_before_any_work():
    prob, obj = problems.build("brachistochrone")
    with pytest.raises(ValueError, match="sqp_core"):
        prob.solve(obj, sqp_core="cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("update", UPDATES)
def test_sqp_core_is_bit_reproducible_from_run_to_run(update, monkeypatch):
    """All reductions of the SQP kernels run in a fixed order (the cooperative kernel's partial sums are
    gathered in workgroup order, ties in the elections go to the lower index): two runs of the same solve
    give the same bits."""
    from opengoddard_amd import sqp
    from opengoddard_amd.engine import HipEngine
    monkeypatch.setenv("OGSQP_GI", update)
    runs = []
    for _ in range(2):
        prob, obj = problems.build("polar_tsto_shipped")
        eng = HipEngine(prob, obj)
        lb, ub = np_path.bounds_arrays(prob)
        res = sqp.minimize_slsqp_hip(eng, prob.p.copy(), lb, ub, ftol=1e-6, maxiter=12)
        runs.append((res.x.copy(), res.fun, res.nfev, res.timing["qp_iterations"]))
        eng.close()
    assert np.array_equal(runs[0][0], runs[1][0])
    assert runs[0][1:] == runs[1][1:]


def test_referee_finds_the_exact_step_of_random_subproblems():
    """oracle/qp_referee.py on small random QPs: the restatement's step lies within rounding of the refined solution,
    a perturbed step at its perturbation, the longdouble residuals fall to ~1e-18, and the multipliers of the active
    inequalities come out non-negative (the active set the restatement reports is the optimal one)."""
    from oracle import qp_referee
    rng = np.random.default_rng(11)
    for trial in range(6):
        n, meq, mg = 30 + 5 * trial, 8 + trial, 25
        Z = np.eye(n) + 0.3 * rng.standard_normal((n, n))
        g, C, c = rng.standard_normal(n), rng.standard_normal((meq, n)), 0.1 * rng.standard_normal(meq)
        G, h = rng.standard_normal((mg, n)), rng.uniform(-0.2, 1.0, mg)
        lb, ub = -np.abs(rng.standard_normal(n)), np.abs(rng.standard_normal(n))
        d, lam, mu, mode, _, info = slsqp_np.qp_solve(Z, g, C, c, G, h, lb, ub)
        if mode != 1:
            continue
        ids = canonical_ids(info["active"], mg)
        dist, d_star, rinfo = qp_referee.distances(Z, g, np.vstack([C, G]), np.concatenate([c, h]), lb, ub, meq, ids,
                                                   {"restatement": d, "perturbed": d + 1e-8})
        assert dist["restatement"] <= 1e-12 and 0.5e-8 <= dist["perturbed"] <= 2e-8
        hist = rinfo["residual_history"]
        assert hist[1] <= 1e-10 * hist[0] and hist[3] <= 1e-16 * max(1.0, hist[0])
        assert rinfo["min_multiplier_of_inequalities"] is None or rinfo["min_multiplier_of_inequalities"] >= -1e-10


@pytest.mark.gpu
def test_gpu_recovers_when_an_inter_workgroup_wait_gives_up(monkeypatch):
    """VERDICT r3 next #6 / ADVICE r3: the look-ahead LQ sweep and the chained triangular solves wait for other
    workgroups of their own launch (bounded).  ``OGSQP_SPIN_LIMIT=1`` makes every such wait give up at once - what a
    device shared with other tenants can do to them - and the subproblem must then be solved again with the
    separate-launch forms instead of failing: same bits as a handle that runs those forms from the start
    (``OGSQP_LQ=16``, ``OGSQP_TRSV=block``), and the handle counts the recovery."""
    from opengoddard_amd.engine import HipEngine
    prob, obj = problems.build("polar_tsto")
    eng = HipEngine(prob, obj)
    lb, ub = np_path.bounds_arrays(prob)
    x = np.clip(prob.p, lb, ub)
    F0, JT = eng.sweep_stacked(x, _native.fd_step(x, lb, ub))
    n, meq = eng.n, eng.m_eq
    g, A, c = JT[:, 0].copy(), JT[:, 1:].T.copy(), F0[1:]
    lo, hi = np.append(lb - x, 0.0), np.append(ub - x, 1.0)          # the relaxed first subproblem (the plain one is mode 4)

    def solve(env):
        for key in ("OGSQP_LQ", "OGSQP_TRSV", "OGSQP_SPIN_LIMIT"):
            monkeypatch.delenv(key, raising=False)
        for key, value in env.items():
            monkeypatch.setenv(key, value)
        core = _sqp_native.QpCore(n, meq, eng.m_ineq)
        first = core.solve(A, g, c, lb - x, ub - x)                    # inconsistent: status 4 either way
        core.set_active()
        out = core.solve(A, g, c, lo, hi, True, 100.0)
        count = core.recoveries()
        core.close()
        return first[3], out, count

    s_safe, safe, r_safe = solve({"OGSQP_LQ": "16", "OGSQP_TRSV": "block"})
    s_lost, lost, r_lost = solve({"OGSQP_SPIN_LIMIT": "1"})
    s_def, default, r_def = solve({})
    assert r_safe == 0 and r_def == 0 and r_lost >= 1
    assert s_safe == s_lost == s_def == 4 and safe[3] == lost[3] == default[3] == 1
    for a, b in zip(safe[:3], lost[:3]):
        assert np.array_equal(a, b)                                    # the recovery IS the safe forms' run
    assert safe[4] == lost[4]
    scale = max(1.0, np.abs(safe[0]).max())
    assert np.max(np.abs(default[0] - safe[0])) <= 1e-6 * scale        # (look-ahead sums in another order: to rounding)
    eng.close()
