"""C2 - Goddard rocket, single phase (maths of reference
``examples/04_Goddard_0knot.py:10-157``).

States (h, v, m), control T; hdot = v, vdot = (T - drag)/m - g, mdot = -T/c with
drag = Dc v^2 exp(-Hc (h - H0)/H0) and g = g0 (H0/h)^2; maximise the final altitude.
One phase, 50 LGL nodes, t in [0, 0.3].  Literature optimum h_f = 1.01283.
"""
import numpy as np


class Vehicle:
    g0 = 1.0

    def __init__(self):
        self.H0, self.V0, self.M0 = 1.0, 0.0, 1.0
        self.Tc, self.Hc, self.Vc, self.Mc = 3.5, 500, 620, 0.6
        self.c = 0.5 * np.sqrt(self.g0 * self.H0)
        self.Mf = self.Mc * self.M0
        self.Dc = 0.5 * self.Vc * self.M0 / self.g0
        self.T_max = self.Tc * self.g0 * self.M0


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
        rows.upper_bound(m, obj.M0)
        rows.upper_bound(T, obj.T_max)
        return rows()

    def cost(prob, obj):
        return -prob.states_all_section(0)[-1]

    return dynamics, equality, inequality, cost


def build(api, nodes=None, max_iteration=30):
    prob = api.Problem([0.0, 0.3], list(nodes or [50]), [3], [1], max_iteration)
    obj = Vehicle()
    t = prob.time_all_section
    G = api.Guess
    prob.set_states_all_section(0, G.cubic(t, 1.0, 0.0, 1.010, 0.0))
    prob.set_states_all_section(1, G.linear(t, 0.0, 0.0))
    prob.set_states_all_section(2, G.cubic(t, 1.0, -0.6, 0.6, 0.0))
    prob.set_controls_all_section(0, G.cubic(t, 3.5, 0.0, 0.0, 0.0))
    dynamics, equality, inequality, cost = make_callbacks(api)
    prob.dynamics = [dynamics]
    prob.knot_states_smooth = []
    prob.cost = cost
    prob.cost_derivative = None
    prob.equality = equality
    prob.inequality = inequality
    return prob, obj
