"""TEST INFRASTRUCTURE - batch-last vectorised NumPy evaluation of the reference path (a CPU baseline).

Only ``tests/`` and ``bench.py``'s ``cpu_baseline`` leg use this; it is never part of the product path.

SURVEY.md section 7.4 item 1 / section 8(d): the strongest honest CPU number for the reference's
algorithm is not its serial column loop (``scipy/optimize/_numdiff.py:584-625``: one Python callback
evaluation per decision variable) but ONE evaluation of the *unmodified* user callbacks on a
batch-last array ``P`` of shape ``(n, B)`` - decision index first, FD column last - whose column b is
``x0 + h_b e_b`` (column 0 = ``x0``).  ``x[0]``, ``x[-1]``, slices, mask assignment and every ufunc
broadcast unchanged; only the row-stacking helpers have to concatenate on axis 0:

* :class:`Condition`  <- ``OpenGoddard/optimize.py:978-1072``  (``hstack`` of rows -> ``concatenate(axis=0)``)
* :class:`Dynamics`   <- ``OpenGoddard/optimize.py:1075-1127`` (the same)
* :func:`equality_add`, :func:`cost_add` <- ``OpenGoddard/optimize.py:670-709`` with ``D.dot(X)`` as one
  GEMM ``(N, N) x (N, B)`` per state instead of B matrix-vector products.

Elementwise rows come out bit-identical to the column loop; collocation rows differ in the last ulp
(GEMM vs GEMV summation order), i.e. the Jacobian agrees within the forward-difference noise floor
(``tests/test_oracle_and_codegen.py::test_batch_last_baseline_matches_the_column_loop``).
"""
from __future__ import annotations

import types

import numpy as np

from opengoddard_amd import optimize as _api
from . import np_path


def _rows(value, batch):
    """A row block ``(rows, B)`` from whatever a callback produced: a per-column scalar ``(B,)``, a
    block ``(rows, B)``, a constant scalar or a constant vector ``(rows,)``."""
    a = np.asarray(value, dtype=float)
    if a.ndim == 0:
        return np.broadcast_to(a, (1, batch))
    if a.ndim == 1:
        if a.shape[0] == batch:
            return a[None, :]
        return np.broadcast_to(a[:, None], (a.shape[0], batch))
    return a


class _Batch:
    size = 1


def _cat(items):
    items = [it for it in items if not (isinstance(it, np.ndarray) and it.size == 0)]
    if not items:
        return np.zeros((0, _Batch.size))
    return np.concatenate([_rows(it, _Batch.size) for it in items], axis=0)


class Condition(_api.Condition):
    def __init__(self, length=0):
        self._condition = np.zeros((length, _Batch.size))

    def add(self, arg, unit=1.0):
        self._condition = _cat([self._condition, np.asarray(arg) / unit])


class Dynamics(_api.Dynamics):
    def __call__(self):
        units = self.unit_states[self.section]
        return _cat([np.asarray(self._rhs[i]) * (self.unit_time / units[i]) for i in range(self.number_of_state)])


class Problem(_api.Problem):
    def states_all_section(self, state):
        return _cat([self.states(state, i) for i in range(self.number_of_section)])

    def controls_all_section(self, control):
        return _cat([self.controls(control, i) for i in range(self.number_of_section)])


api = types.SimpleNamespace(Problem=Problem, Condition=Condition, Dynamics=Dynamics, Guess=_api.Guess)


def equality_add(prob, obj):
    """``optimize.py:670-698`` on a batch-last ``prob.p``."""
    result = _rows(prob.equality(prob, obj), _Batch.size)
    for i in range(prob.number_of_section):
        parts = []
        for j in range(prob.number_of_states[i]):
            state_temp = prob.states(j, i) / prob.unit_states[i][j]
            parts.append(prob.D[i].dot(state_temp))                  # (N, N) x (N, B)
        derivative = np.concatenate(parts, axis=0)
        tix = np.asarray(prob.time_start(i)) / prob.unit_time
        tfx = prob.time_final(i) / prob.unit_time
        dx = _rows(prob.dynamics[i](prob, obj, i), _Batch.size)
        result = np.concatenate([result, derivative - (tfx - tix) / 2.0 * dx], axis=0)
    knots = []
    for knot in range(prob.number_of_section - 1):
        if prob.number_of_states[knot] != prob.number_of_states[knot + 1]:
            continue
        for state in range(prob.number_of_states[knot]):
            param_prev = prob.states(state, knot) / prob.unit_states[knot][state]
            param_post = prob.states(state, knot + 1) / prob.unit_states[knot][state]
            if prob.knot_states_smooth[knot]:
                knots.append((param_prev[-1] - param_post[0])[None, :])
    return np.concatenate([result] + knots, axis=0) if knots else result


def cost_add(prob, obj):
    """``optimize.py:700-709``: Mayer cost plus raw-weight quadrature, summed node by node like ``sum``."""
    total = _rows(prob.cost(prob, obj), _Batch.size)[0]
    if prob.running_cost is None:
        return total
    integrand = _rows(prob.running_cost(prob, obj), _Batch.size)
    weight = np.concatenate([w for w in prob.w])
    acc = 0
    for k in range(weight.shape[0]):                               # Python's left-to-right sum, per column
        acc = acc + integrand[k] * weight[k]
    return total + acc


def stacked_values(prob, obj, P):
    """F for every column of ``P`` (n, B) -> (m, B)."""
    saved, _Batch.size = prob.p, P.shape[1]
    try:
        prob.p = P
        cost = cost_add(prob, obj)[None, :]
        ceq = equality_add(prob, obj)
        cineq = _rows(prob.inequality(prob, obj), P.shape[1])
    finally:
        prob.p = saved
    return np.concatenate([cost, ceq, cineq], axis=0)


def sweep(prob, obj, x0, columns=None):
    """``(F0, h, JT)`` like :func:`np_path.sweep`, from one batch-last evaluation of all columns."""
    lb, ub = np_path.bounds_arrays(prob)
    h = np_path.fd_step(x0, lb, ub)
    cols = np.arange(x0.size) if columns is None else np.asarray(columns)
    P = np.repeat(np.asarray(x0, dtype=float)[:, None], cols.size + 1, axis=1)
    P[cols, np.arange(1, cols.size + 1)] += h[cols]
    dx = P[cols, np.arange(1, cols.size + 1)] - x0[cols]
    F = stacked_values(prob, obj, P)
    JT = ((F[:, 1:] - F[:, :1]) / dx[None, :]).T
    return F[:, 0].copy(), h, np.ascontiguousarray(JT)


def build(name, **options):
    """The registered configuration ``name`` on the batch-last API facade."""
    from opengoddard_amd import problems
    return problems.build(name, api=api, **options)
