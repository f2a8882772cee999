"""Fuzz of the layout / tiling logic: randomly shaped multi-phase problems (node counts that are
not multiples of 4 or 16, a single state, no controls, phases of different sizes, smooth and
non-smooth knots, units) with randomly composed elementwise callbacks.  CPU: traced program and
C++ twin against the NumPy oracle.  GPU: HIP evaluation, structured and dense sweeps against the
twin, bit for bit.  These shapes exercise the padding of the MFMA operand image, the tile and
column work lists and the item tables far away from the BASELINE configurations."""
import numpy as np
import pytest

from opengoddard_amd import _native, codegen
from opengoddard_amd.optimize import Condition, Dynamics, Problem
from oracle import np_path, program_eval, twin

SHAPES = [
    # nodes,        states,     controls,  smooth knots
    ([3],           [1],        [0],       []),
    ([4],           [2],        [1],       []),
    ([5, 3],        [1, 1],     [1, 0],    [True]),
    ([17],          [3],        [2],       []),
    ([16, 7],       [2, 2],     [1, 2],    [False]),
    ([33, 9, 18],   [2, 3, 2],  [1, 1, 0], [True, True]),
    ([21, 21],      [4, 4],     [2, 2],    [True]),
    ([64],          [5],        [1],       []),
    ([65, 31],      [2, 2],     [3, 1],    [True]),
]


class Obj:
    pass


def _term(rng, x):
    """A random smooth elementwise function of one array."""
    kind = rng.integers(0, 8)
    if kind == 0:
        return np.sin(x)
    if kind == 1:
        return np.cos(0.5 * x)
    if kind == 2:
        return np.exp(-0.3 * x)
    if kind == 3:
        return np.sqrt(x ** 2 + 1.0)
    if kind == 4:
        return 1.0 / (1.5 + x ** 2)
    if kind == 5:
        return np.arctan(x)
    if kind == 6:
        return np.where(x > 0.1, x, 0.1 * x)
    return np.maximum(x, -0.25) * 0.7


def make_problem(shape, seed):
    nodes, ns, nc, smooth = shape
    rng = np.random.default_rng(seed)
    S = len(nodes)
    times = [0.0] + list(np.cumsum(rng.uniform(0.5, 2.0, S)))
    prob = Problem([float(t) for t in times], list(nodes), list(ns), list(nc), 3)
    for i in range(S):
        for s in range(ns[i]):
            prob.set_unit_states(s, i, float(rng.choice([1.0, 2.0, 0.5, 3.0])))
        for c in range(nc[i]):
            prob.set_unit_controls(c, i, float(rng.choice([1.0, 4.0])))
    if rng.integers(0, 2):
        prob.set_unit_time(float(rng.choice([2.0, 0.5])))
    prob.p[:-S] = rng.uniform(-1.0, 1.0, prob.number_of_variables - S)
    plan_seed = int(rng.integers(0, 2 ** 31))

    def dynamics(prob, obj, section):
        r = np.random.default_rng(plan_seed + section)
        xs = [prob.states(s, section) for s in range(ns[section])]
        us = [prob.controls(c, section) for c in range(nc[section])]
        dx = Dynamics(prob, section)
        for s in range(ns[section]):
            acc = 0.3 * _term(r, xs[int(r.integers(0, len(xs)))])
            if us:
                acc = acc + 0.5 * us[int(r.integers(0, len(us)))]
            if len(xs) > 1:
                acc = acc - 0.2 * xs[(s + 1) % len(xs)] * _term(r, xs[s])
            if r.integers(0, 3) == 0:
                continue                                  # leave this state's rhs at its zero default
            dx[s] = acc
        return dx()

    def equality(prob, obj):
        rows = Condition()
        rows.equal(prob.states(0, 0)[0], 0.1, unit=prob.unit_states[0][0])
        rows.equal(prob.states_all_section(0)[-1], 0.4)
        if S > 1:
            rows.equal(prob.states(0, S - 1)[0], prob.states(0, S - 2)[-1])
        return rows()

    def inequality(prob, obj):
        rows = Condition()
        rows.lower_bound(prob.states_all_section(0), -3.0)
        rows.upper_bound(prob.time_final(-1), 50.0)
        if nc[0]:
            rows.upper_bound(prob.controls(0, 0) ** 2, 4.0)
        if nodes[0] > 3:
            rows.lower_bound(prob.states(0, 0)[1:-1] * 2.0, -9.0)        # interior slice
        return rows()

    def cost(prob, obj):
        return prob.time_final(-1) - prob.states(0, S - 1)[-1]

    def running_cost(prob, obj):
        return 0.5 * prob.states_all_section(0) ** 2

    prob.dynamics = [dynamics] * S
    prob.knot_states_smooth = list(smooth)
    prob.cost = cost
    prob.running_cost = running_cost if seed % 2 == 0 else None
    prob.equality = equality
    prob.inequality = inequality
    for i in range(S):                       # a few active bounds so that h changes sign
        prob.set_states_bounds(0, i, -1.0, 1.0)
    return prob, Obj()


CASES = [(i, shape) for i, shape in enumerate(SHAPES)]


@pytest.mark.parametrize("seed,shape", CASES)
def test_random_layout_cpu_chain(seed, shape):
    prob, obj = make_problem(shape, seed)
    lb, ub = np_path.bounds_arrays(prob)
    x = np.clip(prob.p, lb, ub)
    F = np_path.stacked_values(prob, obj, x)
    P = codegen.trace_problem(prob, obj)
    assert np.array_equal(program_eval.evaluate(P, prob, x), F)
    tw = twin.Twin(prob, obj, program=P)
    assert np.all(np.abs(tw.values(x) - F) <= 1e-11 * np.maximum(1.0, np.abs(F)) + 1e-9)
    h = _native.fd_step(x, lb, ub)
    assert np.array_equal(h, np_path.fd_step(x, lb, ub))
    _, JT = tw.sweep(x, h)
    _, _, JTo = np_path.sweep(prob, obj, x)
    scale = np.maximum(1.0, np.abs(F)) + 50.0 * max(np.abs(D).max() for D in prob.D)
    bound = 1e-9 * np.abs(JTo) + 64 * np.finfo(float).eps * scale[None, :] / np.abs(h)[:, None]
    assert np.all(np.abs(JT - JTo) <= bound)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,shape", CASES)
def test_random_layout_gpu_bit_exact(seed, shape, monkeypatch):
    from opengoddard_amd.engine import HipEngine
    prob, obj = make_problem(shape, seed)
    lb, ub = np_path.bounds_arrays(prob)
    x = np.clip(prob.p, lb, ub)
    h = _native.fd_step(x, lb, ub)
    eng = HipEngine(prob, obj)
    tw = twin.Twin(prob, obj, program=eng.program, header=eng.header)
    F0c, JTc = tw.sweep(x, h)
    F0, JT = eng.sweep_stacked(x, h)
    assert np.array_equal(eng.eval_stacked(x), F0c)
    assert np.array_equal(F0, F0c) and np.array_equal(JT, JTc)
    for lo, hi in ((0, eng.n // 3), (eng.n // 3, eng.n - 1), (eng.n - 1, eng.n)):
        assert np.array_equal(eng.sweep_stacked(x, h, lo, hi)[1], JTc[lo:hi])      # column shards
    eng.close()
    monkeypatch.setenv("OGPSX_SWEEP", "dense")
    eng = HipEngine(prob, obj)
    assert np.array_equal(eng.sweep_stacked(x, h)[1], JTc)
    eng.close()
