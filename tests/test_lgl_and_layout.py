"""Host logic vs the reference's goldens: LGL construction (og_lgl through the C ABI), the
decision-vector layout / getters / index helpers with their quirks, unit scaling, Guess."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from opengoddard_amd import _native
from opengoddard_amd.optimize import Condition, Dynamics, Guess, Problem

LGL_SIZES = (3, 4, 5, 10, 20, 25, 30, 40, 50, 80, 100, 128, 200)


@pytest.mark.parametrize("n", LGL_SIZES)
def test_lgl_matches_reference(n, lgl_golden):
    """north_star: index ordering identical; tau within 1e-15; w and D within 1e-12 *relative*
    (an absolute 1e-12 on D is below the reference's own error)."""
    tau, w, D = _native.lgl(n)
    rt, rw, rD = lgl_golden["tau_%d" % n], lgl_golden["w_%d" % n], lgl_golden["D_%d" % n]
    assert tau.shape == (n,) and w.shape == (n,) and D.shape == (n, n)
    assert np.all(np.diff(tau) > 0) and tau[0] == -1.0 and tau[-1] == 1.0
    assert np.max(np.abs(tau - rt)) <= 1e-15
    assert np.array_equal(tau, -tau[::-1])                     # antisymmetric to the bit
    assert np.max(np.abs(w / rw - 1.0)) <= 1e-12
    nz = rD != 0
    assert np.array_equal(D == 0, ~nz)                         # same structural zeros
    assert np.max(np.abs(D[nz] / rD[nz] - 1.0)) <= 1e-12
    assert D[0, 0] == -n * (n - 1) * 0.25 and D[-1, -1] == n * (n - 1) * 0.25


def test_lgl_known_values_and_errors():
    tau = _native.lgl(10)[0]      # printed by the reference's smoke script (optimize.py:1135-1150)
    ref = [-1, -0.91953391, -0.73877387, -0.47792495, -0.16527896]
    assert np.allclose(tau[:5], ref, atol=5e-9)
    assert abs(_native.lgl(20)[1].sum() - 2.0) < 1e-14          # quadrature weights sum to 2
    D = _native.lgl(30)[2]
    assert np.max(np.abs(D.sum(axis=1))) < 1e-11                # derivative of a constant
    t = _native.lgl(30)[0]
    assert np.max(np.abs(D @ t ** 3 - 3 * t ** 2)) < 1e-11      # exact for polynomials
    with pytest.raises(_native.NativeError):
        _native.lgl(2)
    with pytest.raises(ValueError):
        Problem([0.0, 1.0], [2], [1], [1])                      # quirk Q2


def _load_layout():
    with open(os.path.join(GOLDEN, "layout.json")) as fh:
        return json.load(fh)


def test_layout_getters_and_index_helpers_match_reference():
    data = _load_layout()
    for case in data["layouts"]:
        nodes, ns, nc = case["nodes"], case["ns"], case["nc"]
        p = Problem([float(i) for i in range(len(nodes) + 1)], list(nodes), list(ns), list(nc))
        assert p.div == case["div"]
        assert int(p.number_of_variables) == case["nvar"]
        for i in range(len(nodes)):
            assert list(p.bounds[p.index_time_final(i)]) == case["tf_bounds"][i]
        p.p = np.arange(p.number_of_variables, dtype=float) + 0.5
        for name, args, expect in case["calls"]:
            if isinstance(expect, str) and expect.startswith("raise:"):
                with pytest.raises(Exception) as info:
                    getattr(p, name)(*args)
                assert type(info.value).__name__ == expect[6:], (name, args)
                continue
            got = getattr(p, name)(*args)
            if name == "time_update":
                # depends on tau: the reference's differs from og_lgl by <= 1.2e-16
                assert np.allclose(got, expect, rtol=0, atol=1e-14)
            else:
                assert np.array_equal(np.asarray(got, dtype=float), np.asarray(expect, dtype=float)), \
                    (name, args)


def test_units_bounds_and_time_scaling_match_reference():
    u = _load_layout()["units"]
    p = Problem([0.0, 100.0, 200.0], [5, 4], [2, 2], [1, 1])
    p.set_unit_states_all_section(0, 10.0)
    p.set_unit_controls_all_section(0, 4.0)
    p.set_unit_time(50.0)
    p.set_states_all_section(0, np.linspace(1.0, 9.0, 9))
    p.set_controls(0, 1, np.array([1.0, 2.0, 3.0, 4.0]))
    p.set_states_bounds(1, 0, -5.0, None)
    p.set_controls_bounds_all_section(0, None, 8.0)
    p.set_time_final_bounds(1, None, 300.0)
    assert np.array_equal(p.p, u["p"])
    assert [float(v) for v in p.time_init] == u["time_init"] and float(p.t0) == u["t0"]
    assert np.allclose(p.time_all_section, u["time_all_section"], rtol=0, atol=1e-14)
    got = [[None if b is None else float(b) for b in pair] for pair in p.bounds]
    assert got == u["bounds"]
    assert [float(p.time_start(i)) for i in range(2)] == u["time_start"]
    assert [float(p.time_final(i)) for i in range(2)] == u["time_final"]
    assert np.allclose(p.time_to_tau(p.time_all_section), u["tau_of_time"], rtol=0, atol=1e-14)


def test_guess_helpers_match_reference_bitwise():
    g = _load_layout()["guess"]
    t = np.array(g["t"])
    assert np.array_equal(Guess.linear(t, 1.5, -2.0), g["linear"])
    assert np.array_equal(Guess.cubic(t, 1.0, -0.6, 0.6, 0.25), g["cubic"])
    assert np.array_equal(Guess.constant(t, 3.25), g["constant"])
    assert np.array_equal(Guess.zeros(t), g["zeros"])


def test_condition_and_dynamics_semantics():
    c = Condition()
    c.equal(np.array([3.0, 4.0]), 1.0, unit=2.0)
    c.lower_bound(5.0, 2.0)
    c.upper_bound(np.array([1.0]), 4.0, unit=3.0)
    assert np.array_equal(c(), [1.0, 1.5, 3.0, 1.0])
    z = Condition(4)
    z.change_value(2, 7.0)
    assert np.array_equal(z(), [0, 0, 7.0, 0])
    p = Problem([0.0, 1.0], [4], [2], [1])
    p.set_unit_states(1, 0, 5.0)
    p.unit_time = 2.0
    d = Dynamics(p, 0)
    d[0] = np.arange(4.0)
    out = d()                                  # state 1 never assigned -> zeros (optimize.py:1111)
    assert np.array_equal(out, np.concatenate([np.arange(4.0) * 2.0, np.zeros(4)]))
    with pytest.raises(AssertionError):
        d[2] = 1.0


def test_setter_length_assertion_and_repr():
    p = Problem([0.0, 1.0], [4], [1], [1])
    with pytest.raises(AssertionError):
        p.set_states(0, 0, np.zeros(5))                          # quirk Q6
    assert "number of variables = 9" in repr(p)
