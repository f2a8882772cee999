"""TEST INFRASTRUCTURE - an extended-precision referee for one QP subproblem of the SQP core.

Two solvers of the same strictly convex QP (SciPy's Fortran chain, ``oracle/slsqp_np.py``, ``csrc/ogsqp.hip``)
agree only as far as the conditioning of the subproblem lets them: the first subproblem of the BASELINE
configurations is a *vertex* solution, where a 1e-12 relative perturbation of the Jacobian moves the step by 2e-4
(DESIGN.md section 9).  A tolerance chosen from that argument says nothing about WHICH solver is nearer the exact
step.  This module computes the exact step of the subproblem the solvers were given - the double-precision data taken
as exact - on the active set they report, and so gives both a distance that can be compared.

On the active set the solution of

    min 1/2 d'B d + g'd      rows d + rhs = 0          (B^-1 = Z Z')

is, with d = Z y and T = rows Z, the solution of  y - T'lam = -Z'g,  T y = -rhs.  It is solved once in double
(Householder QR of T', LAPACK) and then refined: the residuals of BOTH equations are evaluated in ``np.longdouble``
(64-bit mantissa, products with ``rows`` and ``Z`` themselves, never with the rounded product T), the correction
comes from the double factorisation.  Each sweep shrinks the error by about cond(T) x 1e-16; the fixed point is
accurate to about cond(T) x 5e-20 - seven or more digits beyond what a double solver can deliver on these problems.
Only ``tests/`` and ``bench.py``'s parity check call this; it is the checker, never the thing measured.
(No reference counterpart: SciPy's ``slsqp`` has no such check; ``scipy/optimize/_slsqp_py.py:427-432`` is the call
whose result this referees.)
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import solve_triangular

LD = np.longdouble


def active_rows(A, c, lo, hi, m_eq, active, n):
    """The active set as equality rows: ``(rows, rhs)`` with ``rows d + rhs = 0``.  ``A``: m x n general rows
    (``a_j d + c_j``), the first ``m_eq`` always active; ``active``: ids in the numbering of ``og_qp_get_active`` -
    general inequality j (< m_ineq), then ``m_ineq + 2 i`` / ``m_ineq + 2 i + 1`` for the lower / upper bound of
    variable i."""
    m_ineq = A.shape[0] - m_eq
    rows, rhs = [A[:m_eq]], [c[:m_eq]]
    gen = sorted(j for j in active if j < m_ineq)
    if gen:
        rows.append(A[m_eq + np.array(gen)])
        rhs.append(c[m_eq + np.array(gen)])
    for j in sorted(j for j in active if j >= m_ineq):
        i, upper = (j - m_ineq) >> 1, (j - m_ineq) & 1
        e = np.zeros((1, n))                   # d_i - lo_i >= 0  /  hi_i - d_i >= 0: multipliers >= 0 like a general row's
        e[0, i] = -1.0 if upper else 1.0
        rows.append(e)
        rhs.append(np.array([hi[i] if upper else -lo[i]]))
    return np.vstack(rows), np.concatenate(rhs)


def refine(Z, g, rows, rhs, sweeps=6):
    """Exact solution (to about cond x 5e-20) of the equality-constrained QP above.
    -> ``(d, lam, history)``: step, multipliers (``B d + g = rows' lam``) as longdouble arrays; ``history["residuals"]``
    the largest residual entry before each sweep (it falls by orders of magnitude per sweep, then stalls at ~1e-19 x
    the size of the terms that cancel in it - large multipliers raise that floor), ``history["step_moved"]`` how far
    each sweep moved the step: the last entries bound the error of the returned step."""
    Z = np.ascontiguousarray(Z, dtype=np.float64)
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    n, ma = Z.shape[0], rows.shape[0]
    if ma > n:
        raise ValueError("more active rows (%d) than variables (%d): the reported active set is not independent" % (ma, n))
    T = rows @ Z
    Q, R = np.linalg.qr(T.T)                               # T' = Q R,  n x ma, ma x ma
    Zl, Al = Z.astype(LD), rows.astype(LD)
    gl, cl = np.asarray(g, dtype=LD), np.asarray(rhs, dtype=LD)
    ztg = Zl.T @ gl
    y, lam = np.zeros(n, dtype=LD), np.zeros(ma, dtype=LD)
    history, moved, d_prev = [], [], None
    for _ in range(sweeps + 1):
        d = Zl @ y
        if d_prev is not None:
            moved.append(float(np.abs(d - d_prev).max()))
        d_prev = d
        r1 = y + ztg - Zl.T @ (Al.T @ lam)                  # y - T'lam + Z'g
        r2 = Al @ d + cl                                    # T y + rhs
        history.append(float(max(np.abs(r1).max(initial=0.0), np.abs(r2).max(initial=0.0))))
        r1d, r2d = r1.astype(np.float64), r2.astype(np.float64)
        # dy - T'dlam = -r1,  T dy = -r2  with  dy = Q a + w,  w orthogonal to range(Q)
        a = -solve_triangular(R, r2d, trans="T")
        qr1 = Q.T @ r1d
        dlam = solve_triangular(R, a + qr1)
        dy = Q @ a - (r1d - Q @ qr1)
        y = y + dy.astype(LD)
        lam = lam + dlam.astype(LD)
    d = Zl @ y
    moved.append(float(np.abs(d - d_prev).max()))
    return d, lam, {"residuals": history, "step_moved": moved}


def distances(Z, g, A, c, lo, hi, m_eq, active, steps, sweeps=6):
    """Distance of each candidate step in ``steps`` (dict name -> d) to the refined solution on ``active``,
    relative to max(1, |d*|_inf).  -> ``(dict name -> distance, d*, info)``."""
    n = Z.shape[0]
    rows, rhs = active_rows(A, c, lo, hi, m_eq, active, n)
    d_star, lam, history = refine(Z, g, rows, rhs, sweeps)
    scale = max(1.0, float(np.abs(d_star).max()))
    out = {k: float(np.abs(np.asarray(v, dtype=LD)[:n] - d_star).max() / scale) for k, v in steps.items()}
    info = {"active_rows": int(rows.shape[0]), "residual_history": history["residuals"],
            "step_moved": [v / scale for v in history["step_moved"]],
            "min_multiplier_of_inequalities": float(lam[m_eq:].min()) if rows.shape[0] > m_eq else None}
    return out, d_star, info
