"""Edge cases of the API that the BASELINE configurations do not exercise, as small inline
problems: running cost + quirk Q10 (Bryson-Denham from the reference's smoke script,
``OpenGoddard/optimize.py:1373-1451``, known answer 7.9985), an empty inequality set (Q14),
phases with different state counts (knot rows skipped, ``optimize.py:690-691``), smooth knots
(Q9), a constant cost, constant-array operands and ``np.where``.  CPU: tracer/lowering/twin
against the NumPy oracle; GPU: the HIP sweep against the twin, bit for bit."""
import numpy as np
import pytest

from conftest import has_gpu
from opengoddard_amd import _native, codegen
from opengoddard_amd import optimize as og
from opengoddard_amd.optimize import Condition, Dynamics, Guess, Problem
from oracle import np_path, program_eval, twin


class Params:
    max_x = 1.0 / 9.0
    k = 0.7


def bryson_denham(nodes=30):
    def dynamics(prob, obj, section):
        v = prob.states(1, section)
        u = prob.controls(0, section)
        dx = Dynamics(prob, section)
        dx[0] = v
        dx[1] = u
        return dx()

    def equality(prob, obj):
        x = prob.states_all_section(0)
        v = prob.states_all_section(1)
        tf = prob.time_final(-1)
        rows = Condition()
        rows.add(x[0] - 0.0)
        rows.add(v[0] - 1.0)
        rows.add(x[-1] - 0.0)
        rows.add(v[-1] + 1.0)
        rows.add(tf - 1.0)
        return rows()

    def inequality(prob, obj):
        x = prob.states_all_section(0)
        rows = Condition()
        rows.add(x - 0.0)
        rows.add(obj.max_x - x)
        return rows()

    prob = Problem([0, 1.0], [nodes], [2], [1], 10)
    prob.set_states_all_section(0, Guess.constant(prob.time_all_section, 0.1))
    prob.dynamics = [dynamics]
    prob.knot_states_smooth = []
    prob.cost = lambda prob, obj: 0.0
    prob.running_cost = lambda prob, obj: 0.5 * prob.controls_all_section(0) ** 2
    prob.equality = equality
    prob.inequality = inequality
    return prob, Params()


def ragged_two_phase():
    """Phase 0: 2 states / 1 control / 7 nodes; phase 1: 3 states / 2 controls / 5 nodes.
    Different state counts => no built-in knot rows; empty inequality; constant cost."""
    def dynamics(prob, obj, section):
        x = prob.states(0, section)
        v = prob.states(1, section)
        u = prob.controls(0, section)
        dx = Dynamics(prob, section)
        dx[0] = v * np.cos(x)
        dx[1] = np.where(u > 0.2, u, 0.2 * u) - obj.k * v
        if section == 1:
            dx[2] = prob.controls(1, section) * prob.tau[1]       # constant-array operand
        return dx()

    def equality(prob, obj):
        rows = Condition()
        rows.equal(prob.states(0, 0)[0], 0.0)
        rows.equal(prob.states(0, 1)[0], prob.states(0, 0)[-1])
        rows.equal(prob.states(1, 1)[0], prob.states(1, 0)[-1], unit=2.0)
        rows.equal(np.maximum(prob.states(2, 1)[1:3], 0.5), 0.5)
        return rows()

    prob = Problem([0.0, 1.0, 2.5], [7, 5], [2, 3], [1, 2], 3)
    prob.set_unit_states(1, 0, 2.0)
    prob.set_unit_controls(0, 1, 0.5)
    rng = np.random.default_rng(3)
    prob.p[:-2] = rng.uniform(0.1, 1.0, prob.number_of_variables - 2)
    prob.dynamics = [dynamics, dynamics]
    prob.cost = lambda prob, obj: 1.25
    prob.equality = equality
    prob.inequality = lambda prob, obj: Condition()()
    return prob, Params()


def smooth_knots():
    """Two phases with equal state counts and built-in knot continuity rows (quirk Q9: both
    sides divided by the first phase's unit)."""
    def dynamics(prob, obj, section):
        dx = Dynamics(prob, section)
        dx[0] = prob.states(1, section)
        dx[1] = prob.controls(0, section) - np.sin(prob.states(0, section))
        return dx()

    def equality(prob, obj):
        rows = Condition()
        rows.equal(prob.states(0, 0)[0], 0.1)
        return rows()

    def inequality(prob, obj):
        rows = Condition()
        rows.upper_bound(prob.controls_all_section(0), 2.0)
        rows.lower_bound(prob.time_final(0), 0.2)
        return rows()

    prob = Problem([0.0, 1.0, 2.0], [6, 9], [2, 2], [1, 1], 3)
    prob.set_unit_states(0, 0, 3.0)
    prob.set_unit_states(0, 1, 5.0)
    rng = np.random.default_rng(5)
    prob.p[:-2] = rng.uniform(-1.0, 1.0, prob.number_of_variables - 2)
    prob.dynamics = [dynamics, dynamics]
    prob.knot_states_smooth = [True]
    prob.cost = lambda prob, obj: -prob.states(0, 1)[-1]
    prob.equality = equality
    prob.inequality = inequality
    return prob, Params()


def running_cost_shapes():
    """Sequential sums of every shape the cached-term path distinguishes (codegen ``sum_term_q``, kernels
    ``XColT``): a two-phase running cost whose integrand reads two variables at the same node (one perturbed
    term per column), the phase's final time (every term is perturbed: evaluated in place), and a constant
    vector; plus a user sum inside a constraint row."""
    def dynamics(prob, obj, section):
        dx = Dynamics(prob, section)
        dx[0] = prob.states(1, section)
        dx[1] = prob.controls(0, section) - 0.3 * prob.states(0, section)
        return dx()

    def equality(prob, obj):
        rows = Condition()
        rows.equal(prob.states(0, 0)[0], 0.1)
        rows.equal(prob.states(1, 0)[0], 0.0)
        return rows()

    def inequality(prob, obj):
        rows = Condition()
        rows.upper_bound(prob.controls_all_section(0), 2.0)
        rows.lower_bound(prob.time_final(0), 0.2)
        rows.lower_bound(prob.time_final(1), 1.2)
        return rows()

    def running(prob, obj):
        u = prob.controls_all_section(0)
        v = prob.states_all_section(1)
        tf = prob.time_final(-1)
        return (0.5 * u ** 2 + 0.1 * u * v) * tf + obj.k * np.cos(v)

    prob = Problem([0.0, 1.0, 2.0], [20, 37], [2, 2], [1, 1], 3)
    rng = np.random.default_rng(8)
    prob.p[:-2] = rng.uniform(-1.0, 1.0, prob.number_of_variables - 2)
    prob.dynamics = [dynamics, dynamics]
    prob.knot_states_smooth = [True]
    prob.cost = lambda prob, obj: prob.time_final(-1)
    prob.running_cost = running
    prob.equality = equality
    prob.inequality = inequality
    return prob, Params()


def wide_functions():
    """Round 3's widened function set inside dynamics and constraints: a tanh throttle, cubic drag through cbrt /
    hypot, log10 / log2 / log1p / expm1 / sinh / cosh, ``x ** u`` with a traced exponent, ``2 ** x``."""
    def dynamics(prob, obj, section):
        h = prob.states(0, section)
        v = prob.states(1, section)
        m = prob.states(2, section)
        u = prob.controls(0, section)
        w = prob.controls(1, section)
        throttle = 0.5 * (1.0 + np.tanh(4.0 * (u - 0.5)))
        speed = np.hypot(v, w)
        drag = 0.02 * speed ** 2 * np.cbrt(speed) * np.exp2(-h)
        dx = Dynamics(prob, section)
        dx[0] = v
        dx[1] = (2.0 * throttle - drag * v / (speed + 1e-3)) / m - 1.0 / (1.0 + h) ** 2
        dx[2] = -throttle * (1.0 + 0.1 * np.sinh(w) / np.cosh(w)) - 0.01 * np.expm1(-m)
        return dx()

    def equality(prob, obj):
        rows = Condition()
        rows.equal(prob.states(0, 0)[0], 0.05)
        rows.equal(prob.states(2, 0)[0], 1.0)
        rows.equal(np.log10(prob.states(2, 0)[-1] + 1.0) + np.log2(prob.states(2, 0)[-1] + 1.0), 0.9)
        return rows()

    def inequality(prob, obj):
        m = prob.states(2, 0)
        u = prob.controls(0, 0)
        rows = Condition()
        rows.lower_bound(m ** (1.0 + 0.5 * u), 0.05)              # traced exponent
        rows.upper_bound(np.log1p(prob.controls(1, 0) ** 2), 3.0)
        return rows()

    prob = Problem([0.0, 1.5], [14], [3], [2], 3)
    rng = np.random.default_rng(21)
    prob.p[:-1] = rng.uniform(0.2, 1.2, prob.number_of_variables - 1)
    prob.dynamics = [dynamics]
    prob.cost = lambda prob, obj: -prob.states(0, 0)[-1]
    prob.equality = equality
    prob.inequality = inequality
    return prob, Params()


def wide_reductions():
    """Round 3's reductions and index forms: ``np.sum`` / ``.sum()`` / ``.mean()`` (NumPy's pairwise order, also
    above 128 elements), ``np.min`` / ``np.max``, ``np.cumsum``, ``np.roll``, ``x[::-1]``, strided and integer-array
    indexing, ``np.interp``, ``np.diff``, ``np.clip``."""
    table_x = np.array([-2.0, -0.5, 0.0, 0.7, 1.5, 3.0])
    table_y = np.array([0.3, 0.1, 0.0, 0.4, 0.2, 0.9])

    def dynamics(prob, obj, section):
        x = prob.states(0, section)
        v = prob.states(1, section)
        u = prob.controls(0, section)
        dx = Dynamics(prob, section)
        dx[0] = v + 0.01 * np.interp(x, table_x, table_y)
        dx[1] = np.clip(u, -1.0, 1.0) - 0.2 * v - 0.001 * x.mean()      # a term that couples every node
        return dx()

    def equality(prob, obj):
        x = prob.states_all_section(0)
        u = prob.controls_all_section(0)
        rows = Condition()
        rows.equal(x[0], 0.1)
        rows.equal(np.sum(u * u) + u.sum() - np.sum(x[3:9]), 2.0)
        rows.equal(u.mean(), 0.05)
        rows.equal(x[::-1][0:3], x[[-1, -2, -3]])                        # identically zero rows
        return rows()

    def inequality(prob, obj):
        x = prob.states_all_section(0)
        v = prob.states_all_section(1)
        u = prob.controls_all_section(0)
        rows = Condition()
        rows.lower_bound(np.min(x), -3.0)
        rows.upper_bound(np.max(v) + u.max(), 9.0)
        rows.upper_bound(np.cumsum(np.abs(u[0:12])) * 0.1, 5.0)
        rows.lower_bound(np.roll(v, 5) - np.roll(v, -3), -4.0)
        rows.upper_bound(np.diff(x)[::7], 2.5)
        rows.lower_bound(x[::-1] + x[np.arange(x.size)], -6.0)
        return rows()

    prob = Problem([0.0, 1.0, 2.0], [90, 45], [2, 2], [1, 1], 3)     # 135 nodes: pairwise sums above one 128-block
    rng = np.random.default_rng(22)
    prob.p[:-2] = rng.uniform(-1.0, 1.0, prob.number_of_variables - 2)
    prob.dynamics = [dynamics, dynamics]
    prob.knot_states_smooth = [True]
    prob.cost = lambda prob, obj: prob.time_final(-1) + 0.01 * np.sum(prob.controls_all_section(0) ** 2)
    prob.equality = equality
    prob.inequality = inequality
    return prob, Params()


def preallocated_outputs():
    """Round 4: callbacks that BUILD their result the way users write them - ``out = np.zeros(n)`` filled by item and
    by slice, ``np.empty`` + in-place slice arithmetic, ``np.zeros_like``, ``np.array([...])`` of traced scalars,
    ``np.vstack`` / 2-D buffers with ``axis=`` reductions, ``np.prod``, ``%``, ``np.trapz``, ``np.heaviside``."""
    trapz = getattr(np, "trapezoid", None) or np.trapz

    def dynamics(prob, obj, section):
        n = prob.nodes[section]
        x = prob.states(0, section)
        v = prob.states(1, section)
        u = prob.controls(0, section)
        acc = np.zeros(n)                                # allocated in the callback ...
        acc[:] = u - 0.3 * v * np.heaviside(v, 0.5)      # ... filled through a slice
        acc *= 1.0 + 0.1 * (x % 0.37)                    # in-place arithmetic; Python's % on traced values
        drift = np.zeros_like(x)
        drift += 0.01 * np.mod(x, -0.6)
        dx = Dynamics(prob, section)
        dx[0] = v + drift
        dx[1] = acc
        return dx()

    def equality(prob, obj):
        x = prob.states_all_section(0)
        v = prob.states_all_section(1)
        t = prob.time_update()
        ends = np.array([x[0] - 0.1, v[0], x[-1] * v[-1]])            # np.array of traced scalars
        both = np.vstack([x[0:6], v[0:6], np.linspace(0.0, 1.0, 6)])    # 2-D of traced rows
        rows = Condition()
        rows.equal(ends, np.array([0.0, 0.2, 0.05]))
        rows.equal(both.sum(axis=0)[1:4], 0.3)
        rows.equal(trapz(v * v, t), 0.4)
        rows.equal(np.prod(1.0 + 0.1 * x[0:5]), 1.2)
        return rows()

    def inequality(prob, obj):
        u = prob.controls_all_section(0)
        x = prob.states_all_section(0)
        out = np.empty(8)
        for i in range(4):
            out[i] = 2.0 - u[i] * u[i + 1]
        out[4:] = 3.0 - np.abs(x[4:8])
        out[1:3] *= 2.0                                  # in-place arithmetic on a slice of the buffer
        out[::4] = out[::4] + 0.5                        # strided targets
        table = np.zeros((2, 5))
        table[0] = u[0:5]
        table[1, :] = x[0:5] ** 2
        table[:, 4] = np.array([u[5], 0.25])
        rows = Condition()
        rows.lower_bound(out, 0.0)
        rows.upper_bound(table.max(axis=0) + table.mean(axis=1)[0], 6.0)
        rows.lower_bound(np.stack([u[0:3], x[0:3]], axis=1).ravel(), -4.0)
        # a buffer ASSEMBLED from constant pieces and then filled (np.concatenate / np.hstack of np.zeros /
        # np.ones must stay traced buffers, not become NumPy's own ndarray)
        pad = np.concatenate((np.zeros(2), np.ones(3)))
        pad[1:4] = u[6:9] * x[6:9]
        wide = np.hstack((np.zeros(1), np.full(2, 0.5)))
        wide[0] = x[9] - u[9]
        wide[2] += u[10]
        rows.lower_bound(np.append(pad, wide), -5.0)
        return rows()

    prob = Problem([0.0, 1.2], [18], [2], [1], 3)
    rng = np.random.default_rng(23)
    prob.p[:-1] = rng.uniform(-1.0, 1.0, prob.number_of_variables - 1)
    prob.dynamics = [dynamics]
    prob.cost = lambda prob, obj: np.sum(np.array([prob.controls(0, 0)[0:9], prob.controls(0, 0)[9:18]]) ** 2)
    prob.equality = equality
    prob.inequality = inequality
    return prob, Params()


def constant_scratch_buffers():
    """Buffers from ``np.zeros`` / ``np.ones`` that only ever hold plain numbers - a lookup table, a flag vector, an
    integer index list - next to traced ones: ``float(buf[i])``, ``if buf[0] > 0``, ``buf.astype(int)``,
    a buffer handed to a NumPy routine and to ``np.array(..., dtype=int)`` all have concrete answers."""
    def dynamics(prob, obj, section):
        x = prob.states(0, section)
        u = prob.controls(0, section)
        gains = np.zeros(3)
        gains[0], gains[1], gains[2] = 0.5, 1.5, 2.5
        k = float(gains[1])                               # a concrete number out of a constant buffer
        dx = Dynamics(prob, section)
        dx[0] = k * u - x if gains[0] > 0 else u          # control flow on a constant
        return dx()

    def equality(prob, obj):
        x = prob.states_all_section(0)
        pick = np.ones(2)
        pick[1] = 7.0
        idx = pick.astype(int)                            # -> array([1, 7]) again a plain array
        rows = Condition()
        rows.equal(x[0], 0.25)
        rows.equal(x[int(pick[1])] - x[idx[0]], 0.0)
        return rows()

    def inequality(prob, obj):
        u = prob.controls_all_section(0)
        table = np.zeros(4)
        table[:] = [0.0, 1.0, 4.0, 9.0]
        scale = np.interp(1.5, np.arange(4.0), table)     # a NumPy routine on a constant buffer: 2.5
        steps = np.array(np.ones(3) * 2.0, dtype=int)     # dtype=int of a constant buffer
        out = np.zeros(5)                                 # ... and a buffer that DOES receive traced values
        out[0] = float(steps.sum())
        out[1:] = scale - u[0:4] ** 2
        rows = Condition()
        rows.lower_bound(out, -1.0)
        return rows()

    prob = Problem([0.0, 1.0], [9], [1], [1], 3)
    rng = np.random.default_rng(5)
    prob.p[:-1] = rng.uniform(-1.0, 1.0, prob.number_of_variables - 1)
    prob.dynamics = [dynamics]
    prob.cost = lambda prob, obj: prob.time_final(-1)
    prob.equality = equality
    prob.inequality = inequality
    return prob, Params()


CASES = {"bryson_denham": bryson_denham, "constant_scratch_buffers": constant_scratch_buffers, "ragged_two_phase": ragged_two_phase,
         "smooth_knots": smooth_knots, "running_cost_shapes": running_cost_shapes,
         "wide_functions": wide_functions, "wide_reductions": wide_reductions,
         "preallocated_outputs": preallocated_outputs}


@pytest.mark.parametrize("name", sorted(CASES))
def test_trace_and_twin_match_numpy_oracle(name):
    prob, obj = CASES[name]()
    P = codegen.trace_problem(prob, obj)
    lb, ub = np_path.bounds_arrays(prob)
    x = np.clip(prob.p, lb, ub)
    F = np_path.stacked_values(prob, obj, x)
    assert P.m == F.size
    assert np.array_equal(program_eval.evaluate(P, prob, x), F)
    if name == "ragged_two_phase":
        assert P.m_ineq == 0 and P.m_eq == 5 + 2 * 7 + 3 * 5          # no knot rows
    if name == "smooth_knots":
        assert P.m_eq == 1 + 2 * 6 + 2 * 9 + 2                         # + 2 knot rows
    tw = twin.Twin(prob, obj, program=P)
    Ft = tw.values(x)
    assert np.all(np.abs(Ft - F) <= 1e-11 * np.maximum(1.0, np.abs(F)) + 1e-10)
    h = _native.fd_step(x, lb, ub)
    F0, JT = tw.sweep(x, h)
    _, _, JTo = np_path.sweep(prob, obj, x)
    scale = np.maximum(1.0, np.abs(F)) + 200.0
    bound = 1e-9 * np.abs(JTo) + 64 * np.finfo(float).eps * scale[None, :] / np.abs(h)[:, None]
    assert np.all(np.abs(JT - JTo) <= bound)


def test_other_threads_keep_numpys_own_constructors_while_a_trace_runs():
    """``np.zeros`` & co. are replaced on the numpy module while callbacks are traced; only the tracing thread sees the
    replacement: another thread calling ``np.zeros(4)`` from its own code in the middle of a trace gets an
    ndarray, never a traced buffer of somebody else's graph; and a second trace started meanwhile waits for the first."""
    import threading
    from opengoddard_amd import trace
    inside, release = threading.Event(), threading.Event()
    seen = {}
    prob, obj = preallocated_outputs()
    plain_inequality = prob.inequality

    def blocking_inequality(p, o):
        seen["tracer"] = type(np.zeros(3)).__name__           # this thread: a traced buffer
        inside.set()
        assert release.wait(30)
        return plain_inequality(p, o)

    prob.inequality = blocking_inequality
    result = {}
    t = threading.Thread(target=lambda: result.setdefault("P", codegen.trace_problem(prob, obj)))
    t.start()
    assert inside.wait(30)
    try:
        seen["other"] = np.zeros(4)
        seen["other_array"] = np.asarray([1.0, 2.0])
        second = threading.Thread(target=lambda: result.setdefault("Q", codegen.trace_problem(*bryson_denham(8))))
        second.start()
        second.join(0.5)
        assert second.is_alive()                               # the second trace waits for the lock
    finally:
        release.set()
    t.join(60)
    second.join(60)
    assert seen["tracer"] == "Sym"
    assert type(seen["other"]) is np.ndarray and type(seen["other_array"]) is np.ndarray
    assert result["P"].m > 0 and result["Q"].m > 0
    assert np.zeros is trace.np.zeros and type(np.zeros(2)) is np.ndarray      # restored


def test_bryson_denham_known_answer_with_oracle_engine(capsys):
    """The reference converges to 8.0 instead of the analytic 4.0 because the running-cost
    quadrature omits (tf-t0)/2 (quirk Q10); its smoke script prints 7.998497 for 30 nodes."""
    og.ENGINE_FACTORY = np_path.NumpyEngine
    try:
        prob, obj = bryson_denham(30)
        prob.solve(obj)
    finally:
        og.ENGINE_FACTORY = None
    capsys.readouterr()
    cost = float(np_path.cost_add(prob, obj))
    assert abs(cost - 7.998497) < 5e-4


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_sweep_bit_exact_on_edge_problems(name):
    from opengoddard_amd.engine import HipEngine
    prob, obj = CASES[name]()
    eng = HipEngine(prob, obj)
    tw = twin.Twin(prob, obj, program=eng.program, header=eng.header)
    lb, ub = np_path.bounds_arrays(prob)
    x = np.clip(prob.p, lb, ub)
    h = _native.fd_step(x, lb, ub)
    F0, JT = eng.sweep_stacked(x, h)
    F0c, JTc = tw.sweep(x, h)
    assert np.array_equal(F0, F0c) and np.array_equal(JT, JTc)
    assert np.array_equal(eng.eval_stacked(x), F0c)
    eng.close()


@pytest.mark.gpu
def test_gpu_bryson_denham_solve(capsys):
    prob, obj = bryson_denham(30)
    prob.solve(obj)
    capsys.readouterr()
    assert abs(float(np_path.cost_add(prob, obj)) - 7.998497) < 5e-4
    prob._engine.close()
