"""TEST INFRASTRUCTURE - independent check of a returned optimum: the KKT residuals of the NLP the reference states.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s checker legs may import this module; nothing of the
product path does.

The problem ``Problem.solve`` hands to SLSQP (reference ``OpenGoddard/optimize.py:723-749``)::

    minimise   f(p)                       cost_add        optimize.py:700-709
    subject to c_eq(p)   = 0              equality_add    optimize.py:670-698
               c_ineq(p) >= 0             inequality      optimize.py:727
               lb <= p <= ub              self.bounds     optimize.py:740-743

SciPy's Fortran core cannot finish C3 / C5 in any affordable time, so "did the HIP SQP core stop at an optimum of
THAT problem" is answered here without any solver: every function value comes from the NumPy restatement of the
reference path (``oracle/np_path.py``: the Problem's own Python callbacks, the reference's operation order), the
Jacobian from central differences of those values (step 6e-6 relative: truncation and rounding both ~1e-10 relative,
two orders below the forward differences the solver itself saw), the multipliers from least-squares fits of the
cost gradient on the rows that are active at the point (rows priced negative are released and the fit repeated).  Reported, each on its own:

``feasibility``      max( |c_eq|, max(-c_ineq, 0), bound violation )               (absolute, the rows are non-dimensional)
``stationarity``     | g - A_eq' l - A_act' mu - z_lo + z_up |_inf  /  max(1, |g|_inf)
``dual``             most negative multiplier of an active inequality or bound, / max(1, largest multiplier)
``complementarity``  max_i |mu_i c_i|  over ALL inequalities and bounds (inactive rows carry mu = 0, so this prices how
                     far from zero the rows declared active are)
``kkt``              the largest of the four
"""
from __future__ import annotations

import numpy as np

from . import np_path

REL_STEP = 6.0e-6            # central differences: (eps)^(1/3)


def central_jacobian(prob, obj, x, columns=None):
    """``(F0, JT)``: F = [cost | c_eq | c_ineq] at x by the NumPy restatement and JT[j] = dF/dx_j by central
    differences of it (2 evaluations per column, serial like the reference's own FD loop)."""
    x = np.asarray(x, dtype=float)
    F0 = np_path.stacked_values(prob, obj, x)
    columns = range(x.size) if columns is None else columns
    JT = np.empty((len(columns), F0.size))
    xp = x.copy()
    for r, j in enumerate(columns):
        h = REL_STEP * max(1.0, abs(x[j]))
        xp[j] = x[j] + h
        hi = (xp[j] - x[j])
        Fp = np_path.stacked_values(prob, obj, xp)
        xp[j] = x[j] - h
        lo = (x[j] - xp[j])
        Fm = np_path.stacked_values(prob, obj, xp)
        xp[j] = x[j]
        JT[r] = (Fp - Fm) / (hi + lo)
    return F0, JT


def residuals(prob, obj, x, m_eq, active_tol=1e-6, bound_tol=1e-9, jacobian=None, max_rounds=200):
    """KKT residuals of the reference's NLP at ``x`` (see the module text).  ``jacobian=(F0, JT)`` lets a caller that
    already has the oracle's Jacobian hand it in; ``max_rounds`` bounds the number of fits (each a dense least-squares
    problem of n x active rows: a quarter of a minute at C5) - rows still priced negative then show in ``dual``.
    Returns a dict of plain floats / ints."""
    x = np.asarray(x, dtype=float)
    n = x.size
    lb, ub = np_path.bounds_arrays(prob)
    F0, JT = central_jacobian(prob, obj, x) if jacobian is None else jacobian
    g = JT[:, 0]
    ceq, cin = F0[1:1 + m_eq], F0[1 + m_eq:]
    Aeq, Ain = JT[:, 1:1 + m_eq], JT[:, 1 + m_eq:]            # (n, rows): columns are constraint gradients
    feas = max(float(np.max(np.abs(ceq), initial=0.0)), float(np.max(-cin, initial=0.0)),
               float(np.max(lb - x, initial=0.0)), float(np.max(x - ub, initial=0.0)), 0.0)
    act = np.flatnonzero(cin <= active_tol)
    with np.errstate(invalid="ignore"):
        on_lo = np.isfinite(lb) & (x - lb <= bound_tol * np.maximum(1.0, np.abs(np.where(np.isfinite(lb), lb, 0.0))))
        on_up = np.isfinite(ub) & (ub - x <= bound_tol * np.maximum(1.0, np.abs(np.where(np.isfinite(ub), ub, 0.0))))
    at_lo = np.flatnonzero(on_lo)
    at_up = np.flatnonzero(on_up & ~on_lo)
    # The certificate: ANY multipliers with mu >= 0, z >= 0 that make the four residuals small prove an approximate KKT
    # point, however they were found.  They are found by least squares over the free variables on the rows that are
    # active at x (variables on a bound are left out of the fit: their z_j is whatever closes row j); rows and bounds
    # that come out with a negative multiplier - weakly active ones, which a least-squares fit on a degenerate set
    # prices arbitrarily - are released (multiplier 0) and the fit repeated, most negative half first.
    act = list(act)
    at_lo, at_up = list(at_lo), list(at_up)
    floor = None                                              # stationarity no choice of multipliers on these rows beats
    rounds = max(1, int(max_rounds))
    for round_no in range(rounds):
        fixed = np.array(at_lo + at_up, dtype=int)
        free = np.setdiff1d(np.arange(n), fixed)
        M = np.hstack([Aeq, Ain[:, act]])
        lam, *_ = np.linalg.lstsq(M[free], g[free], rcond=None)
        r = g - M @ lam
        if floor is None:
            # the first fit is on EVERY row at zero, signs free: its residual is a lower bound (in the 2-norm) for any
            # certificate - a large value here means the point is not stationary, whatever the pruning below does
            floor = (float(np.max(np.abs(r[free]), initial=0.0)), float(np.linalg.norm(r[free])))
        z_lo, z_up = r[at_lo], -r[at_up]                      # g - A'lam = z_lo - z_up, both >= 0 at an optimum
        signed = np.concatenate([lam[m_eq:], z_lo, z_up])
        worst = float(np.min(signed, initial=0.0))
        scale = max(1.0, float(np.max(np.abs(lam), initial=0.0)), float(np.max(np.abs(signed), initial=0.0)))
        if worst >= -1e-9 * scale or round_no == rounds - 1:   # (the last fit stands as it is: what is still negative shows in `dual`)
            break
        drop = signed <= 0.5 * worst
        ka, kl = len(act), len(at_lo)
        act = [v for v, d in zip(act, drop[:ka]) if not d]
        at_lo = [v for v, d in zip(at_lo, drop[ka:ka + kl]) if not d]
        at_up = [v for v, d in zip(at_up, drop[ka + kl:]) if not d]
    act = np.array(act, dtype=int)
    at_lo, at_up = np.array(at_lo, dtype=int), np.array(at_up, dtype=int)
    r_free = r[free]
    gscale = max(1.0, float(np.max(np.abs(g), initial=0.0)))
    stationarity = float(np.max(np.abs(r_free), initial=0.0)) / gscale
    mu = lam[m_eq:]
    signed = np.concatenate([mu, z_lo, z_up])
    mscale = max(1.0, float(np.max(np.abs(lam), initial=0.0)), float(np.max(np.abs(signed), initial=0.0)))
    dual = max(0.0, -float(np.min(signed, initial=0.0))) / mscale
    comp = max(float(np.max(np.abs(mu * cin[act]), initial=0.0)),
               float(np.max(np.abs(z_lo * (x - lb)[at_lo]), initial=0.0)),
               float(np.max(np.abs(z_up * (ub - x)[at_up]), initial=0.0)))
    return {"kkt": max(feas, stationarity, dual, comp), "feasibility": feas, "stationarity": stationarity,
            "dual": dual, "complementarity": comp, "cost": float(F0[0]),
            "active_inequalities": int(act.size), "variables_on_bounds": int(fixed.size),
            "inequalities_at_zero": int(np.count_nonzero(cin <= active_tol)),
            "free_variables": int(free.size), "equalities": int(m_eq),
            "largest_multiplier": float(np.max(np.abs(lam), initial=0.0)),
            "stationarity_2norm": float(np.linalg.norm(r_free)) / gscale,
            "stationarity_floor_signs_free": floor[0] / gscale, "stationarity_floor_2norm": floor[1] / gscale,
            "gradient_scale": gscale,
            "jacobian": "central differences (rel. step %.0e) of oracle/np_path.stacked_values" % REL_STEP}
