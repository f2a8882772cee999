"""Goddard rocket in TWO phases with one knot (maths of reference ``examples/05_Goddard_1knot.py:10-170``): the
configuration that exercises what the one-phase C2 does not - the built-in knot rows of ``equality_add``
(``knot_states_smooth = [True]``, ``optimize.py:688-696``), a state unit (altitude in units of 0.1: quirk Q7's
``(p u) / u`` round trip), a knot time fixed by an equality on ``time_final(0)``, a user ``cost_derivative`` built with
``Condition(n).change_value`` and quirk Q4: the cost reads ``states_all_section(-1)`` - the LAST CONTROL slice of every
phase scaled by the last state's unit - so what is maximised is the final thrust entry, exactly as the shipped script
does.

States (h, v, m), control T per phase; hdot = v, vdot = (T - drag) / m - g, mdot = -T / c,
drag = Dc v^2 exp(-Hc (h - H0) / H0), g = g0 (H0 / h)^2; 25 + 25 LGL nodes on t in [0, 0.1, 0.3].
"""
import numpy as np

from .goddard import Vehicle


def make_callbacks(api):
    Condition, Dynamics = api.Condition, api.Dynamics

    def dynamics(prob, obj, section):
        h = prob.states(0, section)
        v = prob.states(1, section)
        m = prob.states(2, section)
        T = prob.controls(0, section)
        drag = 1 * obj.Dc * v ** 2 * np.exp(-obj.Hc * (h - obj.H0) / obj.H0)
        grav = obj.g0 * (obj.H0 / h) ** 2
        rhs = Dynamics(prob, section)
        rhs[0] = v
        rhs[1] = (T - drag) / m - grav
        rhs[2] = -T / obj.c
        return rhs()

    def equality(prob, obj):
        h = prob.states_all_section(0)
        v = prob.states_all_section(1)
        m = prob.states_all_section(2)
        rows = Condition()
        rows.equal(h[0], obj.H0)
        rows.equal(v[0], obj.V0)
        rows.equal(m[0], obj.M0)
        rows.equal(v[-1], 0.0)
        rows.equal(m[-1], obj.Mf)
        rows.equal(prob.time_final(0), 0.075)                  # the knot's time
        return rows()

    def inequality(prob, obj):
        h = prob.states_all_section(0)
        v = prob.states_all_section(1)
        m = prob.states_all_section(2)
        T = prob.controls_all_section(0)
        rows = Condition()
        rows.lower_bound(h, obj.H0)
        rows.lower_bound(v, 0.0)
        rows.lower_bound(m, obj.Mf)
        rows.lower_bound(T, 0.0)
        rows.lower_bound(prob.time_final(-1), 0.1)
        rows.lower_bound(prob.time_final(0), 0.05)
        rows.upper_bound(m, obj.M0)
        rows.upper_bound(T, obj.T_max)
        return rows()

    def cost(prob, obj):
        return -prob.states_all_section(-1)[-1]                # quirk Q4: index -1 is the last CONTROL slice

    def cost_derivative(prob, obj):
        grad = Condition(prob.number_of_variables)
        grad.change_value(prob.index_states(0, -1, -1), -1)
        return grad()

    return dynamics, equality, inequality, cost, cost_derivative


def build(api, nodes=None, max_iteration=50):
    prob = api.Problem([0.0, 0.1, 0.3], list(nodes or [25, 25]), [3, 3], [1, 1], max_iteration)
    obj = Vehicle()
    prob.set_unit_states_all_section(0, 0.1)
    G = api.Guess
    t = prob.time_all_section
    prob.set_states_all_section(0, G.cubic(t, 1.0, 0.0, 1.010, 0.0))
    prob.set_states_all_section(1, G.linear(t, 0.0, 0.0))
    prob.set_states_all_section(2, np.hstack((G.linear(prob.time[0], 1.0, 0.6), G.linear(prob.time[1], 0.6, 0.6))))
    prob.set_controls_all_section(0, np.hstack((G.linear(prob.time[0], 3.5, 3.5), G.linear(prob.time[1], 0.0, 0.0))))
    dynamics, equality, inequality, cost, cost_derivative = make_callbacks(api)
    prob.dynamics = [dynamics, dynamics]
    prob.knot_states_smooth = [True]
    prob.cost = cost
    prob.cost_derivative = cost_derivative
    prob.equality = equality
    prob.inequality = inequality
    return prob, obj
