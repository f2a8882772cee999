"""C4' / C4 - planar low-thrust orbit raising.

``variant="3x4"`` is the shape the reference ships (SURVEY.md Appendix D4'; maths of reference
``examples/10_Low_Thrust_Orbit_Transfer.py:10-168``): states (r, vr, vt), four non-negative
controls (ur1, ur2, ut1, ut2) acting as ur1-ur2 and ut1-ut2, running cost = sum of controls.

``variant="7x3"`` is BASELINE.json's "1 phase, 7 states, 3 controls, 200 LGL nodes"
configuration, defined here: states (r, theta, vr, vt, m, J1, J2),
controls (ur, ut, delta) with throttle delta in [0, 1]:

    rdot = vr                      thetadot = vt / r
    vrdot = vt^2/r - 1/r^2 + T delta ur / m
    vtdot = -vr vt / r + T delta ut / m
    mdot = -T delta / ve           J1dot = delta           J2dot = ur^2 + ut^2

with the direction constrained by ur^2 + ut^2 <= 1.  Cost: running cost delta (burn time) with
the raw LGL weights, like the shipped example (quirk Q10).
"""
import numpy as np


class Spacecraft:
    def __init__(self):
        self.u_max = 0.01
        self.r0, self.vr0, self.vt0 = 1.0, 0.0, 1.0
        self.rf, self.vrf, self.vtf = 4.0, 0.0, 0.5
        self.tf_max = 55
        # 7x3 extras
        self.thrust = 0.01
        self.ve = 1.5
        self.m0 = 1.0
        self.m_min = 0.1


def make_callbacks_3x4(api):
    Condition, Dynamics = api.Condition, api.Dynamics

    def dynamics(prob, obj, section):
        r = prob.states(0, section)
        vr = prob.states(1, section)
        vt = prob.states(2, section)
        ur1, ur2 = prob.controls(0, section), prob.controls(1, section)
        ut1, ut2 = prob.controls(2, section), prob.controls(3, section)
        rhs = Dynamics(prob, section)
        rhs[0] = vr
        rhs[1] = vt ** 2 / r - 1 / r ** 2 + (ur1 - ur2)
        rhs[2] = -vr * vt / r + (ut1 - ut2)
        return rhs()

    def equality(prob, obj):
        r = prob.states_all_section(0)
        vr = prob.states_all_section(1)
        vt = prob.states_all_section(2)
        rows = Condition()
        rows.equal(r[0], obj.r0)
        rows.equal(vr[0], obj.vr0)
        rows.equal(vt[0], obj.vt0)
        rows.equal(r[-1], obj.rf)
        rows.equal(vr[-1], obj.vrf)
        rows.equal(vt[-1], obj.vtf)
        return rows()

    def inequality(prob, obj):
        r = prob.states_all_section(0)
        ur1, ur2 = prob.controls_all_section(0), prob.controls_all_section(1)
        ut1, ut2 = prob.controls_all_section(2), prob.controls_all_section(3)
        tf = prob.time_final(-1)
        rows = Condition()
        rows.lower_bound(r, obj.r0)
        for u in (ur1, ut1, ur2, ut2):
            rows.lower_bound(u, 0.0)
        rows.lower_bound(tf, 0.0)
        rows.upper_bound(r, obj.rf)
        for u in (ur1, ut1, ur2, ut2):
            rows.upper_bound(u, obj.u_max)
        rows.upper_bound(tf, obj.tf_max)
        return rows()

    def cost(prob, obj):
        return 0.0

    def running_cost(prob, obj):
        ur1, ur2 = prob.controls_all_section(0), prob.controls_all_section(1)
        ut1, ut2 = prob.controls_all_section(2), prob.controls_all_section(3)
        return (ur1 + ur2) + (ut1 + ut2)

    return dynamics, equality, inequality, cost, running_cost


def make_callbacks_7x3(api):
    Condition, Dynamics = api.Condition, api.Dynamics

    def dynamics(prob, obj, section):
        r = prob.states(0, section)
        vr = prob.states(2, section)
        vt = prob.states(3, section)
        m = prob.states(4, section)
        ur, ut = prob.controls(0, section), prob.controls(1, section)
        delta = prob.controls(2, section)
        accel = obj.thrust * delta / m
        rhs = Dynamics(prob, section)
        rhs[0] = vr
        rhs[1] = vt / r
        rhs[2] = vt ** 2 / r - 1 / r ** 2 + accel * ur
        rhs[3] = -vr * vt / r + accel * ut
        rhs[4] = -obj.thrust * delta / obj.ve
        rhs[5] = delta
        rhs[6] = ur ** 2 + ut ** 2
        return rhs()

    def equality(prob, obj):
        first = [(0, obj.r0), (1, 0.0), (2, obj.vr0), (3, obj.vt0), (4, obj.m0), (5, 0.0), (6, 0.0)]
        last = [(0, obj.rf), (2, obj.vrf), (3, obj.vtf)]
        rows = Condition()
        for state, value in first:
            rows.equal(prob.states_all_section(state)[0], value)
        for state, value in last:
            rows.equal(prob.states_all_section(state)[-1], value)
        return rows()

    def inequality(prob, obj):
        r = prob.states_all_section(0)
        m = prob.states_all_section(4)
        ur, ut = prob.controls_all_section(0), prob.controls_all_section(1)
        delta = prob.controls_all_section(2)
        tf = prob.time_final(-1)
        rows = Condition()
        rows.lower_bound(r, obj.r0)
        rows.lower_bound(delta, 0.0)
        rows.lower_bound(m, obj.m_min)
        rows.lower_bound(tf, 0.0)
        rows.upper_bound(delta, 1.0)
        rows.upper_bound(ur ** 2 + ut ** 2, 1.0)
        rows.upper_bound(tf, obj.tf_max)
        return rows()

    def cost(prob, obj):
        return 0.0

    def running_cost(prob, obj):
        return prob.controls_all_section(2)

    return dynamics, equality, inequality, cost, running_cost


def build(api, variant="3x4", nodes=None, max_iteration=10):
    obj = Spacecraft()
    G = api.Guess
    if variant == "3x4":
        prob = api.Problem([0.0, 10.0], list(nodes or [100]), [3], [4], max_iteration)
        t = prob.time_all_section
        prob.set_states_all_section(0, G.linear(t, obj.r0, obj.rf))
        prob.set_states_all_section(1, G.linear(t, obj.vr0, obj.vrf))
        prob.set_states_all_section(2, G.linear(t, obj.vt0, obj.vtf))
        prob.set_controls_all_section(0, G.linear(t, obj.u_max, obj.u_max))
        prob.set_controls_all_section(2, G.linear(t, obj.u_max, obj.u_max))
        callbacks = make_callbacks_3x4(api)
    elif variant == "7x3":
        prob = api.Problem([0.0, 30.0], list(nodes or [200]), [7], [3], max_iteration)
        t = prob.time_all_section
        prob.set_states_all_section(0, G.linear(t, obj.r0, obj.rf))
        prob.set_states_all_section(1, G.linear(t, 0.0, 6.0))
        prob.set_states_all_section(2, G.cubic(t, obj.vr0, 0.02, obj.vrf, 0.0))
        prob.set_states_all_section(3, G.linear(t, obj.vt0, obj.vtf))
        prob.set_states_all_section(4, G.linear(t, obj.m0, 0.85))
        prob.set_states_all_section(5, G.linear(t, 0.0, 20.0))
        prob.set_states_all_section(6, G.linear(t, 0.0, 25.0))
        prob.set_controls_all_section(0, G.constant(t, 0.3))
        prob.set_controls_all_section(1, G.constant(t, 0.8))
        prob.set_controls_all_section(2, G.cubic(t, 0.9, 0.0, 0.5, 0.0))
        prob.set_states_bounds_all_section(4, obj.m_min, obj.m0)
        prob.set_controls_bounds_all_section(2, 0.0, 1.0)
        callbacks = make_callbacks_7x3(api)
    else:
        raise ValueError("variant must be '3x4' or '7x3'")
    dynamics, equality, inequality, cost, running_cost = callbacks
    prob.dynamics = [dynamics]
    prob.knot_states_smooth = []
    prob.cost = cost
    prob.running_cost = running_cost
    prob.equality = equality
    prob.inequality = inequality
    return prob, obj
