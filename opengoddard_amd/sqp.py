"""SLSQP major iteration with the QP subproblem and the quasi-Newton matrix on the GPU.

What it stands in for: ``scipy.optimize.minimize(method='SLSQP')`` as the reference calls it
(``optimize.py:723-749``; SciPy's driver ``scipy:_slsqp_py.py:214-512`` around Kraft's Fortran
``slsqpb``).  The algorithm is Kraft's, statement for statement where it decides anything - QP
subproblem, relaxed QP for an inconsistent linearisation (``rho`` = 100, x10 up to five times),
multiplier averaging for the L1 merit function, the inexact line search
(``alpha = max(h3 / (2 (h3 - h1)), 0.1)``, at most ten cuts), Powell-damped BFGS, up to five
resets on a non-descent direction, the relaxed convergence test after the last reset, and the exit
modes 0 / 4 / 6 / 8 / 9 with SciPy's messages - so iteration counts and results match SciPy's up
to the rounding of the QP solution (``tests/test_slsqp_core.py`` replays SciPy's own iterates).

What is different is where the data lives.  The FD Jacobian never leaves HBM: the sweep kernel
writes it transposed, the QP core (``include/ogsqp.h``) reads it in place, and only O(n) vectors
(step, multipliers, gradients of the Lagrangian) cross PCIe.  The O(n) logic below runs on the
host in NumPy, like SciPy's wrapper does.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

from . import _native, _sqp_native

EXIT_MODES = {-1: "Gradient evaluation required (g & a)",
              0: "Optimization terminated successfully",
              1: "Function evaluation required (f & c)",
              2: "More equality constraints than independent variables",
              3: "More than 3*n iterations in LSQ subproblem",
              4: "Inequality constraints incompatible",
              5: "Singular matrix E in LSQ subproblem",
              6: "Singular matrix C in LSQ subproblem",
              7: "Rank-deficient equality constraint subproblem HFTI",
              8: "Positive directional derivative for linesearch",
              9: "Iteration limit reached"}


_DEBUG = bool(os.environ.get("OGSQP_DEBUG"))


class SqpResult(dict):
    """Attribute access like ``scipy.optimize.OptimizeResult``."""
    __getattr__ = dict.get
    __setattr__ = dict.__setitem__


class DeviceJacobian:
    """The transposed FD Jacobian of one engine, resident in HBM.

    One device: torch owns ``x | h | F0 | J_T`` and the sweep writes into them (``og_fd_sweep_dev``).  An engine
    made with ``devices=[d0, d1, ...]`` shards the FD columns of every sweep over those GPUs
    (``og_multi_fd_sweep_enqueue``: one launch per device, one all-gather of the packed non-zeros) and the QP core
    reads device d0's replica of the whole matrix in place (``og_multi_replica_dev(g = 0)``), stream-ordered behind
    the exchange - the Jacobian takes no trip through the host either way.  The exact-Jacobian mode is a
    single-device kernel and uses the first form."""

    def __init__(self, engine):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("the HIP SQP core needs a GPU (no CPU fallback)")
        self.engine = engine
        self.torch = torch
        self.n, self.ld = engine.n, engine.m                 # ld = 1 + m_eq + m_ineq
        self._own = None                                     # (x, h, F0, JT, stream) of the one-device form
        self._multi = None
        multi = getattr(engine, "_multi", None)
        if multi is not None and multi.value:
            import ctypes as C
            jt, f0, stream = C.c_void_p(), C.c_void_p(), C.c_void_p()
            _native.check(engine._lib.og_multi_replica_dev(multi, 0, C.byref(jt), C.byref(f0), C.byref(stream)),
                          "og_multi_replica_dev")
            self._multi = (multi, jt.value, f0.value, stream.value or 0)
        else:
            self._single()
        self._last = "multi" if self._multi else "own"

    def _single(self):
        if self._own is None:
            torch, engine = self.torch, self.engine
            dev = torch.device("cuda", engine.device)
            d_x = torch.empty(self.n, dtype=torch.float64, device=dev)
            d_h = torch.empty(self.n, dtype=torch.float64, device=dev)
            d_F0 = torch.empty(self.ld, dtype=torch.float64, device=dev)
            d_JT = torch.empty(self.n * self.ld, dtype=torch.float64, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            # persistent-zero output: every sweep of the solve writes the non-zeros only (include/ogpsx.h)
            engine.register_jt_dev(d_JT.data_ptr(), 0, self.n, stream)
            self._own = (d_x, d_h, d_F0, d_JT, stream)
        return self._own

    def sweep(self, x, lb, ub):
        """One FD sweep at ``x`` (SciPy's step rule); returns F(x) on the host."""
        torch = self.torch
        h = _native.fd_step(x, lb, ub)
        self.last_step = h
        x = np.ascontiguousarray(x, dtype=np.float64)
        exact = getattr(self.engine, "jacobian_mode", "fd") == "exact"
        if self._multi and not exact:
            multi, _, d_f0, _ = self._multi
            lib = self.engine._lib
            _native.check(lib.og_multi_fd_sweep_enqueue(multi, _native.dptr(x), _native.dptr(h)),
                          "og_multi_fd_sweep_enqueue")
            F0 = np.empty(self.ld)
            # (synchronises device d0: its replica is complete when its stream - sweep, exchange, unpack - is)
            _native.check(lib.og_device_read(int(self.engine.device), d_f0, F0.ctypes.data, 8 * self.ld),
                          "og_device_read")
            self._last = "multi"
            return F0
        d_x, d_h, d_F0, d_JT, stream = self._single()
        self._last = "own"
        d_x.copy_(torch.from_numpy(x))
        if exact:
            self.engine.exact_dev(d_x.data_ptr(), 0, self.n, d_JT.data_ptr(), d_F0.data_ptr(), stream)
        else:
            d_h.copy_(torch.from_numpy(h))
            self.engine.sweep_dev(d_x.data_ptr(), d_h.data_ptr(), 0, self.n, d_JT.data_ptr(), d_F0.data_ptr(), stream)
        return d_F0.cpu().numpy()

    @property
    def ptr(self):
        """Device address of the matrix the LAST sweep wrote."""
        return self._multi[1] if self._last == "multi" else self._own[3].data_ptr()

    @property
    def stream(self):
        return self._multi[3] if self._last == "multi" else self._own[4]

    @property
    def d_JT(self):
        """The one-device form's matrix as a torch tensor (n * ld entries, row i = FD column i)."""
        return self._single()[3]

    @property
    def sharded_over(self):
        """Number of devices the last sweep was split over."""
        return len(self.engine.devices) if self._last == "multi" else 1


# capacity of the QP core (include/ogsqp.h, csrc/ogsqp.hip og_qp_create / rows_lds_bytes): the column-split panels of the
# wide LQ sweep take rows of n + 1 <= 16384 entries (round 5; 8192 before - the reference, ``optimize.py:759-781``, has no
# bound, SciPy's core just gets slower: hours per major iteration from 6 000 variables on); the active-set kernels keep
# three vectors of the null-space dimension in LDS
MAX_N1 = 16384
MAX_NULL_SPACE = 6736


def _cache(engine):
    """(DeviceJacobian, QpCore) of an engine, built once: the restarts of Problem.solve reuse buffers and work space."""
    cache = getattr(engine, "_sqp_cache", None)
    if cache is None:
        n, meq, mineq = engine.n, engine.m_eq, engine.m_ineq
        cache = (DeviceJacobian(engine), _sqp_native.QpCore(n, meq, mineq, device=engine.device))
        try:
            engine._sqp_cache = cache
        except AttributeError:
            pass
    return cache


def prepare(engine):
    """Can the HIP core run this engine's problem?  None when it can (device buffers and QP work space are then
    built and cached on the engine), otherwise the reason as text.  ``Problem.solve(sqp_core="auto")`` asks before it
    commits to the HIP core; ``sqp_core="hip"`` does not ask and fails loudly."""
    n, meq = int(engine.n), int(engine.m_eq)
    if n + 1 > MAX_N1:
        return "n + 1 = %d exceeds the QP core's %d-entry row segments" % (n + 1, MAX_N1)
    if n + 1 - meq > MAX_NULL_SPACE:
        return "null space of the equalities (%d) exceeds the active-set kernels' limit of %d" % (n + 1 - meq,
                                                                                                 MAX_NULL_SPACE)
    try:
        import torch
    except Exception as exc:                                   # torch is optional everywhere else in the package
        return "torch is not importable (%s)" % (exc,)
    if not torch.cuda.is_available():
        return "torch sees no GPU"
    try:
        _cache(engine)
    except (RuntimeError, ImportError, OSError) as exc:        # SqpNativeError is a RuntimeError
        return "building the QP core failed: %s" % (exc,)
    return None


def minimize_slsqp_hip(engine, x0, lb, ub, ftol=1e-6, maxiter=100, cost_derivative=None, disp=False,
                       callback=None, iprint=1):
    """Minimise with the engine's callbacks.  ``engine`` is a :class:`~.engine.HipEngine`;
    ``cost_derivative(x) -> (n,)`` replaces the FD cost gradient when given.  Returns an
    :class:`SqpResult` with SciPy's fields plus ``timing`` (seconds spent in callbacks / QP / BFGS)."""
    n, meq, mineq = engine.n, engine.m_eq, engine.m_ineq
    m = meq + mineq
    lb = np.asarray(lb, dtype=float)
    ub = np.asarray(ub, dtype=float)
    x = np.clip(np.asarray(x0, dtype=float), lb, ub)
    # device buffers and the QP work space belong to the engine: the restarts of Problem.solve reuse them
    t_setup = time.perf_counter()
    jacobian, core = _cache(engine)
    # ("setup": device buffers and the QP work space - two n x n factors, the mailbox ... - on the first call of an engine)
    timing = {"callbacks": 0.0, "qp": 0.0, "bfgs": 0.0, "qp_iterations": 0, "qp_solves": 0, "recoveries": 0,
              "setup": time.perf_counter() - t_setup}
    recoveries_before = core.recoveries()
    unit0 = np.zeros(m + 1)
    unit0[0] = 1.0

    last_p = [x.copy()]

    def evaluate(xv):
        t = time.perf_counter()
        F = engine.eval_stacked(xv)
        last_p[0] = xv.copy()
        timing["callbacks"] += time.perf_counter() - t
        return float(F[0]), F[1:]

    def linearise(xv):
        """Sweep at xv: JT stays on the device, the cost gradient comes back."""
        t = time.perf_counter()
        F = jacobian.sweep(xv, lb, ub)
        last_p[0] = xv.copy()
        last_p[0][-1] += jacobian.last_step[-1]          # quirk Q13: the FD loop ends on the last column
        if cost_derivative is None:
            grad = core.jt_times(jacobian.ptr, jacobian.ld, unit0, jacobian.stream)
        else:
            grad = np.asarray(cost_derivative(xv), dtype=float).reshape(n)
        timing["callbacks"] += time.perf_counter() - t
        return float(F[0]), F[1:], grad

    def lagrangian_gradient(grad, r):
        """v = g - A'r with A on the device."""
        coef = np.concatenate([[0.0], -r])
        return grad + core.jt_times(jacobian.ptr, jacobian.ld, coef, jacobian.stream)

    def violation(cv):
        return float(np.sum(np.abs(cv[:meq])) + np.sum(np.maximum(-cv[meq:], 0.0)))

    acc = abs(ftol)
    tol = 10.0 * acc
    f, c, g = linearise(x)
    nfev, njev = 1, 1
    mu = np.zeros(m)

    def merit_terms(cv):
        return float(mu[:meq] @ np.abs(cv[:meq]) + mu[meq:] @ np.maximum(-cv[meq:], 0.0))

    itermx = maxiter - 1
    it = 0
    status = None
    badlin = False
    f0 = f
    s = np.zeros(n)
    ireset = 0
    reset = True
    while status is None:
        if reset:
            ireset += 1
            if ireset > 5:
                ok = ((abs(f - f0) < tol or np.linalg.norm(s) < tol) and violation(c) < tol
                      and not badlin and f == f)
                status = 0 if ok else 8
                break
            core.reset()
            reset = False
        it += 1
        if it > itermx:
            status = 9
            break
        dl, du = lb - x, ub - x
        t = time.perf_counter()
        if _DEBUG:
            print("[sqp] it %d solving, finite g %s c %s x %s" % (it, bool(np.all(np.isfinite(g))),
                  bool(np.all(np.isfinite(c))), bool(np.all(np.isfinite(x)))), flush=True, file=sys.stderr)
        d, r, bmult, mode, qp_it = core.solve_dev(jacobian.ptr, jacobian.ld, g, c, dl, du, False, 100.0,
                                                  jacobian.stream)
        timing["qp_solves"] += 1
        timing["qp_iterations"] += qp_it
        h4 = 1.0
        badlin = False
        if mode == 6 and n == meq:
            mode = 4
        if mode == 4:
            badlin = True
            rho = 100.0
            for _ in range(6):
                da, r, bmult, mode, qp_it = core.solve_dev(jacobian.ptr, jacobian.ld, g, c, np.append(dl, 0.0),
                                                           np.append(du, 1.0), True, rho, jacobian.stream)
                timing["qp_solves"] += 1
                timing["qp_iterations"] += qp_it
                if mode != 4:
                    break
                rho *= 10.0
            if mode == 1:
                d = da[:n]
                h4 = 1.0 - da[n]
                bmult = bmult[:n]
        timing["qp"] += time.perf_counter() - t
        if _DEBUG:
            print("[sqp] it %d qp mode %d iterations %d badlin %s |d| %.3e finite g %s c %s" % (
                it, mode, qp_it, badlin, float(np.max(np.abs(d))) if mode == 1 else np.nan,
                bool(np.all(np.isfinite(g))), bool(np.all(np.isfinite(c)))), flush=True, file=sys.stderr)
        if mode != 1:
            status = mode
            break
        s = d.copy()
        v = lagrangian_gradient(g, r)
        f0 = f
        x_base = x.copy()
        gs = float(g @ s)
        h1 = abs(gs)
        h2 = violation(c)
        absr = np.abs(r)
        mu = np.maximum(absr, 0.5 * (mu + absr))
        h1 += float(absr @ np.abs(c))
        if h1 < acc and h2 < acc and not badlin and f == f:
            status = 0
            break
        h1 = merit_terms(c)
        t0 = f + h1
        h3 = gs - h1 * h4
        if h3 >= 0.0:
            reset = True
            continue
        Bd = bmult - v                           # B d from the stationarity of the QP
        alpha = 1.0
        fraction = 1.0
        line = 0
        while True:
            line += 1
            h3 = alpha * h3
            s = alpha * s
            fraction *= alpha
            x = np.clip(x_base + s, lb, ub)
            f, c = evaluate(x)
            nfev += 1
            h1 = f + merit_terms(c) - t0
            if h1 <= h3 / 10.0 or line > 10:
                break
            alpha = max(h3 / (2.0 * (h3 - h1)), 0.1)
        if callback is not None:
            callback(np.copy(x))
        if disp and iprint >= 2:
            print("%5i %5i % 16.6E % 16.6E" % (it, nfev, f, np.linalg.norm(g)))
        h3 = violation(c)
        if (abs(f - f0) < acc or np.linalg.norm(s) < acc) and h3 < acc and not badlin and f == f:
            status = 0
            break
        f, c, g_new = linearise(x)
        njev += 1
        eta = lagrangian_gradient(g_new, r) - v
        g = g_new
        t = time.perf_counter()
        if core.bfgs(s, eta, fraction * Bd):
            reset = True
        timing["bfgs"] += time.perf_counter() - t
    timing["recoveries"] = core.recoveries() - recoveries_before
    # (round 6) how much of the active-set work ran as the one resident launch (csrc/ogsqp_resident.h): totals of the handle
    timing["resident_launches"], timing["resident_changes"] = core.resident_stats()
    if getattr(engine, "_sqp_cache", None) is None:
        core.close()
    message = EXIT_MODES.get(int(status), "mode %d" % status)
    if disp:
        print(message + "    (Exit mode " + str(int(status)) + ")")
        print("            Current function value:", f)
        print("            Iterations:", it)
        print("            Function evaluations:", nfev)
        print("            Gradient evaluations:", njev)
    return SqpResult(x=x, fun=f, jac=g, nit=int(it), nfev=nfev, njev=njev, status=int(status),
                     message=message, success=(status == 0), timing=timing,
                     last_callback_p=last_p[0])
