"""TEST INFRASTRUCTURE - CPU (NumPy) restatement of the SQP driver that sits above the callbacks.

SURVEY.md section 8(f) rank 1: the reference reaches SciPy's Fortran ``slsqp`` through
``scipy.optimize.minimize(method='SLSQP')`` (``optimize.py:723-749`` -> ``scipy:_slsqp_py.py:427-432``).
SciPy 1.15.3 is a third-party dependency that is not vendored under ``/root/reference`` and ships
no Fortran source in its wheel, so this file restates Kraft's published algorithm (D. Kraft, "A
software package for sequential quadratic programming", DFVLR-FB 88-28, 1988) and is pinned by
running SciPy itself on the same inputs (``tests/test_slsqp_core.py``):

* ``slsqp``      - the major iteration: QP subproblem, the augmented QP for inconsistent
                   linearisations, L1 merit function with multiplier averaging, Armijo-type
                   inexact line search (step ``max(h3/(2(h3-h1)), 0.1)``, at most 10 cuts), Powell-
                   damped BFGS, the reset / relaxed-tolerance exits (modes 0, 4, 6, 8, 9).
* ``qp_solve``   - the strictly convex QP ``min 1/2 d'Bd + g'd  s.t.  C d + c = 0,  G d + h >= 0,
                   lb <= d <= ub``.  Its solution is unique, so any exact method reproduces the
                   step of SciPy's LSQ/LSEI/LSI/LDP/NNLS chain up to rounding.  The method here is
                   the one the HIP core uses: ``B^-1 = Z Z'`` is carried as the factor ``Z``; the
                   equalities are eliminated with one orthogonal LQ sweep of ``[C Z; Z]``; the
                   inequalities become a least-distance problem ``min |y|^2, W y + b >= 0`` in the
                   null space, solved by the dual active-set method of Goldfarb & Idnani (Math.
                   Programming 27 (1983) 1-33).
* ``bfgs_factor_update`` - product-form BFGS on the inverse factor: ``Z+ = Z - s (v'Z)/alpha``.

Nothing outside ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s CPU-baseline leg may
import this module.
"""
from __future__ import annotations

import numpy as np

EXIT_MODES = {0: "Optimization terminated successfully",
              2: "More equality constraints than independent variables",
              4: "Inequality constraints incompatible",
              6: "Singular matrix C in LSQ subproblem",
              8: "Positive directional derivative for linesearch",
              9: "Iteration limit reached"}

DEPENDENT = 1e-10        # |component outside the active normals| / |normal| below this -> dependent
FEASIBLE = 1e-12         # normalised violation below this counts as satisfied
SINGULAR_C = np.finfo(float).eps   # lsei: ABS(C(I,I)) < EPMACH -> mode 6 (an absolute test, as there)
REDUNDANT = 1e-13        # |L_kk| <= this * max|L_jj|: equality k is a combination of the ones before it ...
CONSISTENT = 1e-9        # ... redundant if its residual is below this * (1 + max|c|), else mode 6


# ----------------------------------------------------------------------------------------------
# least-distance problem, Goldfarb-Idnani
# ----------------------------------------------------------------------------------------------
def ldp_gi(W, b, full_norms=None, reference=None, max_iter=None, warm=None, active_out=None):
    """``min 1/2 |y|^2  s.t.  W y + b >= 0``.  Returns ``(y, u, status, iterations)`` with
    multipliers ``u >= 0`` (``y = W' u``); status 1 solved, 4 incompatible, 3 iteration limit.

    ``full_norms[j]`` is the length of constraint ``j``'s normal before it was projected onto the
    null space of the equalities: a row whose projection is shorter than ``DEPENDENT`` times that
    is a combination of the equalities (e.g. an inequality that repeats an equality) and cannot
    be influenced by ``y``; it is left out.

    ``warm``: rows to start from instead of the empty active set (the rows active at the solution of the previous
    subproblem of an SQP run).  The dual method may start from any S-pair - a point that minimises ``|y|^2`` on
    the rows of a linearly independent set taken as equalities, with non-negative multipliers: the warm rows are
    factorised in the given order (rows that depend on the ones before them are left out), the minimiser on
    them and its multipliers are computed, and while a multiplier is negative the row with the most negative
    one is taken out (one iteration each, like a partial step).  What is left is an S-pair; the solution is the
    same, reached in fewer changes.  ``active_out`` (a list) receives the rows active at the solution."""
    mt, nr = W.shape
    y = np.zeros(nr)
    u = np.zeros(mt)
    if mt == 0:
        return y, u, 1, 0
    norms = np.sqrt(np.einsum("ij,ij->i", W, W))
    if full_norms is None:
        full_norms = norms
    usable = norms > DEPENDENT * full_norms
    scale = np.where(usable, norms, 1.0)
    own = FEASIBLE * np.abs(b) / scale                      # rounding level of each row's value
    s = b.astype(float).copy()                              # constraint values W y + b
    Q = np.eye(nr)                                          # orthogonal; first q columns span the active normals
    R = np.zeros((0, 0))
    active = []                                             # constraint index per column of R
    is_active = np.zeros(mt, dtype=bool)
    iters = 0
    limit = max_iter if max_iter is not None else 10 * (mt + nr) + 100
    if warm:
        for p in warm:
            p = int(p)
            if not usable[p] or is_active[p] or len(active) >= nr:
                continue
            d = Q.T @ W[p]
            q = len(active)
            if float(d[q:] @ d[q:]) <= (DEPENDENT * norms[p]) ** 2:
                continue
            Q, R = _append_column(Q, R, d[:q], d[q:])
            active.append(p)
            is_active[p] = True
        while active:
            q = len(active)
            y1 = _forward_substitute(R, -b[active])         # R' y1 = -b_A: the warm rows hold as equalities
            ua = _back_substitute(R, y1)                    # y = N u
            k = int(np.argmin(ua))
            if not ua[k] < 0.0:
                y = Q[:, :q] @ y1
                u[active] = ua
                s = W @ y + b
                s[active] = 0.0
                break
            iters += 1
            is_active[active[k]] = False
            Q, R = _delete_column(Q, R, k)
            del active[k]
    while True:
        viol = np.where(usable & ~is_active, s / scale + own + FEASIBLE * np.sqrt(y @ y), np.inf)
        p = int(np.argmin(viol))
        if not viol[p] < 0.0:
            if active_out is not None:
                active_out[:] = active
            return y, u, 1, iters
        normal = W[p]
        up = 0.0
        while True:
            iters += 1
            if iters > limit:
                return y, u, 3, iters
            q = len(active)
            d = Q.T @ normal
            d1, d2 = d[:q], d[q:]
            r = _back_substitute(R, d1)
            zz = float(d2 @ d2)                             # = z'z = z'normal
            dependent = zz <= (DEPENDENT * norms[p]) ** 2
            t1, k = np.inf, -1
            for j in range(q):
                if r[j] > 0.0:
                    cand = u[active[j]] / r[j]
                    if cand < t1:
                        t1, k = cand, j
            t2 = np.inf if dependent else -s[p] / zz
            t = min(t1, t2)
            if not np.isfinite(t):
                return y, u, 4, iters
            for j in range(q):
                u[active[j]] -= t * r[j]
            up += t
            if not dependent:
                z = Q[:, q:] @ d2
                y += t * z
                s += t * (W @ z)
            if np.isfinite(t2) and t2 <= t1:                # full step: p becomes active
                Q, R = _append_column(Q, R, d1, d2)
                active.append(p)
                is_active[p] = True
                u[p] = up
                s[p] = 0.0
                break
            u[active[k]] = 0.0                              # partial step: constraint k leaves
            is_active[active[k]] = False
            Q, R = _delete_column(Q, R, k)
            del active[k]


def _forward_substitute(R, rhs):
    """Solve ``R' x = rhs`` (``R`` upper triangular)."""
    q = R.shape[0]
    out = np.zeros(q)
    for i in range(q):
        out[i] = (rhs[i] - R[:i, i] @ out[:i]) / R[i, i]
    return out


def _back_substitute(R, rhs):
    q = R.shape[0]
    out = np.zeros(q)
    for i in range(q - 1, -1, -1):
        out[i] = (rhs[i] - R[i, i + 1:] @ out[i + 1:]) / R[i, i]
    return out


def _append_column(Q, R, d1, d2):
    """Householder H with H d2 = -sign(d2[0]) |d2| e1, applied to the trailing columns of Q."""
    q = R.shape[0]
    delta = float(np.sqrt(d2 @ d2))
    alpha = -delta if d2[0] >= 0 else delta
    v = d2.copy()
    v[0] -= alpha
    vv = float(v @ v)
    if vv > 0.0:
        tail = Q[:, q:]
        Q[:, q:] = tail - np.outer(tail @ v, (2.0 / vv) * v)
    R2 = np.zeros((q + 1, q + 1))
    R2[:q, :q] = R
    R2[:q, q] = d1
    R2[q, q] = alpha
    return Q, R2


def _delete_column(Q, R, k):
    q = R.shape[0]
    R = np.delete(R, k, axis=1)                             # q x (q-1), upper Hessenberg from column k
    for j in range(k, q - 1):
        a, bb = R[j, j], R[j + 1, j]
        rho = np.hypot(a, bb)
        if rho == 0.0:
            continue
        c, sn = a / rho, bb / rho
        upper, lower = R[j, j:].copy(), R[j + 1, j:].copy()
        R[j, j:] = c * upper + sn * lower
        R[j + 1, j:] = -sn * upper + c * lower
        qa, qb = Q[:, j].copy(), Q[:, j + 1].copy()
        Q[:, j] = c * qa + sn * qb
        Q[:, j + 1] = -sn * qa + c * qb
    return Q, R[:q - 1, :]


# ----------------------------------------------------------------------------------------------
# QP subproblem
# ----------------------------------------------------------------------------------------------
def _lq_lapack(C, Z):
    """``[C Z; Z] Q = [L 0; J]`` with LAPACK's Householder QR of ``(C Z)'`` - the same factorisation as the
    loop in :func:`qp_solve` up to the signs of the columns (the QP solution does not depend on them), two orders
    of magnitude faster at n > 1000.  Returns ``None`` when a pivot is small enough for the redundancy rule of the
    loop to matter (the caller then runs the loop)."""
    import scipy.linalg as sl
    meq = C.shape[0]
    Q, R = sl.qr((C @ Z).T, mode="full")
    diag = np.abs(np.diag(R)[:meq])
    if meq and not diag.min() > 1e3 * REDUNDANT * diag.max():
        return None
    T = np.empty((meq + Z.shape[0], Z.shape[1]))
    T[:meq, :meq] = R[:meq, :meq].T
    T[:meq, meq:] = 0.0
    T[meq:] = Z @ Q
    return T


def qp_solve(Z, g, C, c, G, h, lb, ub, lq="loop", warm=None):
    """``min 1/2 d'Bd + g'd`` with ``B^-1 = Z Z'``; ``C d + c = 0``; ``G d + h >= 0``;
    ``lb <= d <= ub`` (non-finite entries: no bound).

    Returns ``(d, lam, mu_g, mode, Znew, info)``: multipliers of the equalities (free sign) and of
    the general inequalities (>= 0) in the convention ``grad L = B d + g - C'lam - G'mu - ...``;
    ``mode`` 1 solved / 4 incompatible / 6 singular C; ``Znew = Z Q`` is an equally valid inverse
    factor whose first ``meq`` columns are B-conjugate to the null space of ``C``.

    ``lq="lapack"``: the LQ sweep through LAPACK (:func:`_lq_lapack`) - for the BASELINE sizes, where the
    reflector-by-reflector loop below takes minutes.  ``warm``: constraint ids ``("g", j)`` / ``("l", i)`` /
    ``("u", i)`` the active-set method starts from (see :func:`ldp_gi`); ``info["active"]`` returns the ids
    active at the solution."""
    n = Z.shape[0]
    meq = C.shape[0]
    info = {"ldp_iterations": 0}
    if meq > n:
        return np.zeros(n), np.zeros(meq), np.zeros(G.shape[0]), 2, Z, info
    # LQ of C Z with the same orthogonal transformations applied to Z
    T = _lq_lapack(C, Z) if lq == "lapack" else None
    fast = T is not None
    if not fast:
        T = np.vstack([C @ Z, Z])
    dmax = 0.0
    for k in range(0 if not fast else meq, meq):
        row = T[k, k:]
        sigma = float(np.sqrt(row @ row))
        # what is left of a row that depends on the earlier ones is rounding noise: no reflector from it
        live = sigma > REDUNDANT * dmax and sigma > 0.0
        dmax = max(dmax, sigma)
        if not live:
            T[k, k] = 0.0
            continue
        alpha = -sigma if row[0] >= 0 else sigma
        v = row.copy()
        v[0] -= alpha
        beta = 2.0 / float(v @ v)
        block = T[:, k:]
        T[:, k:] = block - np.outer(block @ v, beta * v)
        T[k, k] = alpha
        T[k, k + 1:] = 0.0
    L = T[:meq, :meq]                                       # = R', lower triangular
    J = T[meq:]
    diag = np.abs(np.diag(L))
    # a vanishing pivot: that equality is a combination of the earlier ones.  Redundant (dropped: zero
    # component, zero multiplier) if its residual vanishes too, "singular matrix C" (mode 6) if not
    tiny = max(REDUNDANT * (diag.max() if meq else 0.0), SINGULAR_C)
    gone = ~(diag > tiny)
    tol = CONSISTENT * (1.0 + (np.abs(c).max() if meq else 0.0))
    J1, Y = J[:, :meq], J[:, meq:]
    w1 = np.zeros(meq)
    for i in range(meq):                                    # L w1 = -c
        num = -c[i] - L[i, :i] @ w1[:i]
        if gone[i]:
            if abs(num) > tol:
                return np.zeros(n), np.zeros(meq), np.zeros(G.shape[0]), 6, Z, info
            continue
        w1[i] = num / L[i, i]
    d_eq = J1 @ w1 - Y @ (Y.T @ g)
    has_lb, has_ub = np.isfinite(lb), np.isfinite(ub)
    GJ = G @ J
    W = np.vstack([GJ[:, meq:], Y[has_lb], -Y[has_ub]])
    b = np.concatenate([G @ d_eq + h, (d_eq - lb)[has_lb], (ub - d_eq)[has_ub]])
    full = np.concatenate([np.sqrt(np.einsum("ij,ij->i", GJ, GJ)),
                           np.sqrt(np.einsum("ij,ij->i", J, J))[has_lb],
                           np.sqrt(np.einsum("ij,ij->i", J, J))[has_ub]])
    reference = np.concatenate([np.abs(h), np.abs(lb[has_lb]), np.abs(ub[has_ub])])
    mg = G.shape[0]
    ilb, iub = np.nonzero(has_lb)[0], np.nonzero(has_ub)[0]
    rows = None
    if warm:
        where = {("g", j): j for j in range(mg)}
        where.update({("l", int(i)): mg + k for k, i in enumerate(ilb)})
        where.update({("u", int(i)): mg + ilb.size + k for k, i in enumerate(iub)})
        rows = [where[w] for w in warm if w in where]
    final = []
    y, u, status, iters = ldp_gi(W, b, full, reference, warm=rows, active_out=final)
    info["ldp_iterations"] = iters
    info["active"] = [("g", r) if r < mg else ("l", int(ilb[r - mg])) if r < mg + ilb.size
                      else ("u", int(iub[r - mg - ilb.size])) for r in final]
    if status != 1:
        return np.zeros(n), np.zeros(meq), np.zeros(G.shape[0]), 4 if status == 4 else 3, Z, info
    d = d_eq + Y @ y
    mu_g = u[:mg]
    ub_mult = np.zeros(n)
    ub_mult[has_lb] += u[mg:mg + int(has_lb.sum())]
    ub_mult[has_ub] -= u[mg + int(has_lb.sum()):]
    rhs = w1 + J1.T @ (g - G.T @ mu_g - ub_mult)
    lam = np.zeros(meq)
    for i in range(meq - 1, -1, -1):                        # L' lam = rhs
        if not gone[i]:
            lam[i] = (rhs[i] - L[i + 1:, i] @ lam[i + 1:]) / L[i, i]
    d = np.minimum(np.maximum(d, np.where(has_lb, lb, -np.inf)), np.where(has_ub, ub, np.inf))
    info["bound_multipliers"] = ub_mult
    return d, lam, mu_g, 1, J, info


def bfgs_factor_update(Z, s, eta, Bs):
    """Powell-damped BFGS in product form on the inverse factor (``B^-1 = Z Z'``).
    Returns the new factor, or ``None`` when the update is undefined (caller resets)."""
    h1 = float(s @ eta)
    h2 = float(s @ Bs)
    h3 = 0.2 * h2
    r = eta
    if h1 < h3:
        theta = (h2 - h3) / (h2 - h1)
        h1 = h3
        r = theta * eta + (1.0 - theta) * Bs
    if not (h1 > 0.0 and h2 > 0.0):
        return None
    alpha = np.sqrt(h1 / h2)
    v = (r - alpha * Bs) / (alpha * h2)
    return Z - np.outer(s, (v @ Z) / alpha)


# ----------------------------------------------------------------------------------------------
# major iteration
# ----------------------------------------------------------------------------------------------
def slsqp(fun, jac, x0, lb, ub, meq, ftol=1e-6, maxiter=100, callback=None, trace=None,
          teacher=None, qp=None):
    """``fun(x) -> (f, c)`` with ``c`` = equalities then inequalities (``>= 0``); ``jac(x) ->
    (g, A)`` with ``A`` of shape ``(m, n)``.  Mirrors ``minimize(method='SLSQP')`` semantics:
    ``x0`` clipped to the bounds, iterates clipped before evaluation, ``nit``/``status``/``message``.

    ``teacher`` (tests only): the accepted iterates of another SLSQP run on the same callbacks.
    After each line search the iterate is compared with the teacher's (recorded in ``trace`` as
    ``mismatch``) and then replaced by it, so that the comparison of iteration k+1 starts from
    identical inputs - the FD Jacobian amplifies a 1e-13 difference in x to 1e-5 in A, which
    makes free-running trajectories diverge although every step agrees to rounding.
    ``qp``: alternative QP solver with the signature of :func:`qp_solve` (the HIP core's)."""
    solve_qp = qp or qp_solve
    lb = np.asarray(lb, dtype=float)
    ub = np.asarray(ub, dtype=float)
    x = np.clip(np.asarray(x0, dtype=float), lb, ub)
    n = x.size
    acc = abs(ftol)
    tol = 10.0 * acc
    f, c = fun(x)
    g, A = jac(x)
    m = c.size
    mu = np.zeros(m)
    itermx = maxiter - 1
    it = 0
    nfev, njev = 1, 1
    status = None
    x0_ = x.copy()
    f0 = f
    s = np.zeros(n)
    ireset = 0
    badlin = False
    Z = np.eye(n)

    def violation(cv):
        return float(np.sum(np.abs(cv[:meq])) + np.sum(np.maximum(-cv[meq:], 0.0)))

    def merit_terms(cv):
        return float(mu[:meq] @ np.abs(cv[:meq]) + mu[meq:] @ np.maximum(-cv[meq:], 0.0))

    reset = True
    while status is None:
        if reset:
            ireset += 1
            if ireset > 5:
                ok = ((abs(f - f0) < tol or np.linalg.norm(s) < tol) and violation(c) < tol
                      and not badlin and f == f)
                status = 0 if ok else 8
                break
            Z = np.eye(n)
            reset = False
        it += 1
        if it > itermx:
            status = 9
            break
        dl, du = lb - x, ub - x
        C, G = A[:meq], A[meq:]
        d, lam, mug, mode, Zq, info = solve_qp(Z, g, C, c[:meq], G, c[meq:], dl, du)
        h4 = 1.0
        badlin = False
        if mode == 6 and n == meq:
            mode = 4
        if mode == 4:
            badlin = True
            extra = np.concatenate([-c[:meq], np.maximum(-c[meq:], 0.0)])
            rho = 100.0
            incons = 0
            while True:
                Za = np.zeros((n + 1, n + 1))
                Za[:n, :n] = Z
                Za[n, n] = 1.0 / rho       # Kraft's LSQ puts l(n3) itself, not its square root, on E's diagonal
                Aa = np.hstack([A, extra[:, None]])
                da, lam, mug, mode, Zq, info = solve_qp(
                    Za, np.append(g, 0.0), Aa[:meq], c[:meq], Aa[meq:], c[meq:],
                    np.append(dl, 0.0), np.append(du, 1.0))
                if mode == 4:
                    rho *= 10.0
                    incons += 1
                    if incons > 5:
                        break
                    continue
                break
            if mode != 1:
                status = mode
                break
            d = da[:n]
            h4 = 1.0 - da[n]
            Zq = None                                       # augmented factor is not carried over
        elif mode != 1:
            status = mode
            break
        r = np.concatenate([lam, mug])
        s = d.copy()
        v = g - A.T @ r
        f0 = f
        x0_ = x.copy()
        gs = float(g @ s)
        h1 = abs(gs)
        h2 = violation(c)
        absr = np.abs(r)
        mu = np.maximum(absr, 0.5 * (mu + absr))
        h1 += float(absr @ np.abs(c))
        if trace is not None:
            trace.append({"x": x.copy(), "d": d.copy(), "r": r.copy(), "f": f,
                          "ldp_iterations": info.get("ldp_iterations", 0)})
        if h1 < acc and h2 < acc and not badlin and f == f:
            status = 0
            break
        h1 = merit_terms(c)
        t0 = f + h1
        h3 = gs - h1 * h4
        if h3 >= 0.0:
            reset = True
            continue
        if Zq is not None:
            Z = Zq                                          # same B, columns rotated by the QP
        Bs_unit = -v                                        # B d = -(g - A'r) up to the bound multipliers
        if "bound_multipliers" in info:
            Bs_unit = Bs_unit + info["bound_multipliers"][:n]
        line = 0
        alpha = 1.0
        step_fraction = 1.0
        while True:
            line += 1
            h3 = alpha * h3
            s = alpha * s
            step_fraction *= alpha
            x = np.clip(x0_ + s, lb, ub)
            f, c = fun(x)
            nfev += 1
            t = f + merit_terms(c)
            h1 = t - t0
            if h1 <= h3 / 10.0 or line > 10:
                break
            alpha = max(h3 / (2.0 * (h3 - h1)), 0.1)
        if teacher is not None and len(teacher) > len(trace):
            forced = np.asarray(teacher[len(trace)], dtype=float)
            trace[-1]["mismatch"] = float(np.max(np.abs(forced - x)))
            trace[-1]["step"] = float(np.max(np.abs(forced - x0_)))
            if not np.array_equal(forced, x):
                x = forced.copy()
                s = x - x0_
                f, c = fun(x)
        h3 = violation(c)
        if ((abs(f - f0) < acc or np.linalg.norm(s) < acc) and h3 < acc and not badlin and f == f):
            status = 0
            if callback is not None:
                callback(x.copy())
            break
        g_new, A_new = jac(x)
        njev += 1
        if callback is not None:
            callback(x.copy())
        eta = (g_new - A_new.T @ r) - v
        g, A = g_new, A_new
        Bs = step_fraction * Bs_unit
        Znew = bfgs_factor_update(Z, s, eta, Bs)
        if Znew is None:
            reset = True
            continue
        Z = Znew
    return {"x": x, "fun": f, "nit": it, "nfev": nfev, "njev": njev, "status": int(status),
            "message": EXIT_MODES.get(int(status), "mode %d" % status), "success": status == 0}
