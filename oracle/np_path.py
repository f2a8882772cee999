"""TEST INFRASTRUCTURE - CPU oracle of the hot path, NumPy restatement.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; it is never part of the shipped evaluation path (the product fails loudly
without its HIP extension instead of coming here).

What it restates, function by function:

* :func:`equality_add`  <- closure ``equality_add`` in reference ``OpenGoddard/optimize.py:670-698``
* :func:`cost_add`      <- closure ``cost_add``     in reference ``OpenGoddard/optimize.py:700-709``
* :func:`fd_step`       <- ``approx_derivative`` step selection, SciPy 1.15.3
  ``scipy/optimize/_numdiff.py:500-515`` + ``_adjust_scheme_to_bounds`` ``:44-70`` ('1-sided')
* :func:`dense_difference` <- ``_dense_difference`` ``scipy/optimize/_numdiff.py:584-625``
  ('2-point' branch): one Python callback evaluation per decision-vector column.

The FD Jacobian arithmetic lives in SciPy (pinned here: 1.15.3, not vendored under the
reference), so parity is anchored on the reference's call site (``optimize.py:723-749`` passes
no ``jac``) and pinned by golden vectors captured from the reference + SciPy in
``tests/golden/`` (``tools/make_golden.py``).  With NumPy doing the same operations in the
same order as the reference, this restatement reproduces the goldens bit for bit on the same
NumPy build (``tests/test_oracle_and_codegen.py::test_numpy_oracle_reproduces_reference_bitwise``).

It is also the "port" CPU baseline: the serial column loop is exactly how the reference
spends its time.
"""
from __future__ import annotations

import numpy as np

ABS_STEP = 1.4901161193847656e-08      # scipy/optimize/_slsqp_py.py:33  (_epsilon = sqrt(eps))


def equality_add(prob, obj):
    """User equalities, then per phase ``D x~ - (tf~ - t0~)/2 f``, then knot rows."""
    result = prob.equality(prob, obj)
    for i in range(prob.number_of_section):
        derivative = np.zeros(0)
        for j in range(prob.number_of_states[i]):
            state_temp = prob.states(j, i) / prob.unit_states[i][j]
            derivative = np.hstack((derivative, prob.D[i].dot(state_temp)))
        tix = prob.time_start(i) / prob.unit_time
        tfx = prob.time_final(i) / prob.unit_time
        dx = prob.dynamics[i](prob, obj, i)
        result = np.hstack((result, derivative - (tfx - tix) / 2.0 * dx))
    for knot in range(prob.number_of_section - 1):
        if prob.number_of_states[knot] != prob.number_of_states[knot + 1]:
            continue
        for state in range(prob.number_of_states[knot]):
            param_prev = prob.states(state, knot) / prob.unit_states[knot][state]
            param_post = prob.states(state, knot + 1) / prob.unit_states[knot][state]
            if prob.knot_states_smooth[knot]:
                result = np.hstack((result, param_prev[-1] - param_post[0]))
    return result


def cost_add(prob, obj):
    """Mayer cost plus raw-weight LGL quadrature of the running cost (builtin ``sum``)."""
    not_integrated = prob.cost(prob, obj)
    if prob.running_cost is None:
        return not_integrated
    integrand = prob.running_cost(prob, obj)
    weight = np.concatenate([w for w in prob.w])
    return not_integrated + sum(integrand * weight)


def callbacks(prob, obj):
    """The three SciPy-facing functions ``f(p)`` (reference ``wrap_for_solver``,
    ``optimize.py:711-715``: assign ``prob.p`` then evaluate)."""
    def wrap(fn):
        def call(p):
            prob.p = p
            return fn()
        return call
    return (wrap(lambda: cost_add(prob, obj)),
            wrap(lambda: equality_add(prob, obj)),
            wrap(lambda: prob.inequality(prob, obj)))


def stacked_values(prob, obj, p):
    """F(p) = [cost | c_eq | c_ineq] as one float64 vector."""
    saved = prob.p
    try:
        cost, ceq, cineq = (np.atleast_1d(np.asarray(f(p), dtype=float))
                            for f in callbacks(prob, obj))
    finally:
        prob.p = saved
    return np.concatenate([cost, ceq, cineq])


def bounds_arrays(prob):
    lb = np.array([-np.inf if b[0] is None else b[0] for b in prob.bounds], dtype=float)
    ub = np.array([np.inf if b[1] is None else b[1] for b in prob.bounds], dtype=float)
    return lb, ub


def fd_step(x0, lb, ub):
    """Signed forward-difference steps SciPy would use at ``x0`` (absolute step sqrt(eps),
    zero-step fallback, sign flip / shrink at bounds)."""
    x0 = np.asarray(x0, dtype=float)
    sign_x0 = (x0 >= 0).astype(float) * 2 - 1
    h = np.full_like(x0, ABS_STEP)
    dx = (x0 + h) - x0
    h = np.where(dx == 0, ABS_STEP * sign_x0 * np.maximum(1.0, np.abs(x0)), h)
    if np.all((lb == -np.inf) & (ub == np.inf)):
        return h
    adjusted = h.copy()
    lower_dist = x0 - lb
    upper_dist = ub - x0
    x = x0 + h
    violated = (x < lb) | (x > ub)
    fitting = np.abs(h) <= np.maximum(lower_dist, upper_dist)
    adjusted[violated & fitting] *= -1
    forward = (upper_dist >= lower_dist) & ~fitting
    adjusted[forward] = upper_dist[forward]
    backward = (upper_dist < lower_dist) & ~fitting
    adjusted[backward] = -lower_dist[backward]
    return adjusted


def dense_difference(fun, x0, f0, h, columns=None):
    """``J_transposed`` rows for the requested columns: one ``fun`` call per column."""
    x0 = np.asarray(x0, dtype=float)
    columns = range(x0.size) if columns is None else columns
    f0 = np.atleast_1d(f0)
    out = np.empty((len(columns), f0.size))
    x1 = x0.copy()
    for r, i in enumerate(columns):
        x1[i] += h[i]
        dx = x1[i] - x0[i]
        df = np.atleast_1d(fun(x1)) - f0
        out[r] = df / dx
        x1[i] = x0[i]
    return out


def sweep(prob, obj, x0, columns=None):
    """Full oracle sweep at ``x0``: returns ``(F0, h, JT)`` with ``JT`` = transposed Jacobian
    rows of the stacked function for ``columns`` (default all)."""
    lb, ub = bounds_arrays(prob)
    h = fd_step(x0, lb, ub)
    saved = prob.p
    try:
        f0 = stacked_values(prob, obj, x0)
        JT = dense_difference(lambda p: stacked_values(prob, obj, p), x0, f0, h, columns)
    finally:
        prob.p = saved
    return f0, h, JT


class NumpyEngine:
    """Same interface as :class:`opengoddard_amd.engine.HipEngine`, evaluated by the oracle.
    Lets the test-suite run ``Problem.solve`` end to end on a machine without a GPU."""

    def __init__(self, prob, obj):
        self.prob, self.obj = prob, obj
        self.m_eq = None
        self.evals = 0

    def _split(self, F):
        if self.m_eq is None:
            saved = self.prob.p
            self.m_eq = int(np.atleast_1d(callbacks(self.prob, self.obj)[1](saved)).size)
            self.prob.p = saved
        return F[0], F[1:1 + self.m_eq], F[1 + self.m_eq:]

    def values(self, p):
        self.evals += 1
        p = np.array(p, dtype=float, copy=True)
        F = stacked_values(self.prob, self.obj, p)
        return self._split(F)

    def jacobians(self, p, lb, ub):
        p = np.array(p, dtype=float, copy=True)
        h = fd_step(p, lb, ub)
        f0 = stacked_values(self.prob, self.obj, p)
        saved = self.prob.p
        JT = dense_difference(lambda q: stacked_values(self.prob, self.obj, q), p, f0, h)
        self.prob.p = saved
        self._split(f0)
        J = JT.T
        return (J[0], J[1:1 + self.m_eq], J[1 + self.m_eq:]), h
