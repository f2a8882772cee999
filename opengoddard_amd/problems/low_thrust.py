"""C4' / C4 - planar low-thrust orbit raising.

``variant="3x4"`` is the shape the reference ships (SURVEY.md Appendix D4'; maths of reference
``examples/10_Low_Thrust_Orbit_Transfer.py:10-168``): states (r, vr, vt), four non-negative
controls (ur1, ur2, ut1, ut2) acting as ur1-ur2 and ut1-ut2, running cost = sum of controls.

``variant="7x3"`` is BASELINE.json's "1 phase, 7 states, 3 controls, 200 LGL nodes"
configuration, defined here: the same transfer in THREE dimensions
(cylindrical coordinates about the initial orbit's axis) with a 10 degree plane change and a
power-limited engine - states (r, theta, z, vr, vt, vz, m), controls (ur, ut, uz) = the thrust
vector as a fraction of the maximum, |u| <= 1:

    rdot = vr          thetadot = vt / r          zdot = vz          d = (r^2 + z^2)^(3/2)
    vrdot = vt^2/r - r/d + T ur / m
    vtdot = -vr vt / r  + T ut / m
    vzdot = -z/d        + T uz / m
    mdot  = -(T / ve) (ur^2 + ut^2 + uz^2)

from the unit circular orbit to the circular orbit of radius 4 inclined by 10 degrees (reached at
its node: z = 0, vz = v sin i, vt = v cos i).  Cost: the running cost 100 (ur^2 + ut^2 + uz^2) with the
raw LGL weights, like the shipped example (quirk Q10) - the minimum-energy transfer.  Round 4
replaced round 1's planar throttle-times-direction form (controls ur, ut, delta with the thrust
delta (ur, ut)): its product of controls leaves the direction undetermined wherever the throttle
is zero and scales the pair freely elsewhere, and SLSQP wandered along that valley for thousands
of iterations at every size (n = 301: no exit mode 0 in 2000 iterations).  The quadratic cost
makes the controls unique; the reference's own SLSQP converges on this form in 131 / 241
iterations at 30 / 60 nodes.
"""
import numpy as np


class Spacecraft:
    def __init__(self):
        self.u_max = 0.01
        self.r0, self.vr0, self.vt0 = 1.0, 0.0, 1.0
        self.rf, self.vrf, self.vtf = 4.0, 0.0, 0.5
        self.tf_max = 55
        # 7x3: three-dimensional, power-limited
        self.thrust = 0.02
        self.ve = 1.5
        self.m0 = 1.0
        self.m_min = 0.1
        self.inclination = np.deg2rad(10.0)
        speed = 1.0 / np.sqrt(self.rf)
        self.vt_target = speed * np.cos(self.inclination)
        self.vz_target = speed * np.sin(self.inclination)
        # weight of the 7x3 running cost: its Hessian in the controls, 2 x weight x w_i with LGL weights of 200 nodes
        # between 5e-5 and 1.6e-2, is then of the order of the identity SLSQP starts its quasi-Newton matrix from
        # (measured, MI355X, ftol 1e-6: weight 1 - exit mode 0 after 724 major iterations, 10 - 386, 100 - 127, 1000 - 170;
        # problems/launch4.py has the longer story)
        self.effort_scale = 100.0


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
        r, theta, z, vr, vt, vz, m = (prob.states(i, section) for i in range(7))
        ur, ut, uz = (prob.controls(i, section) for i in range(3))
        d2 = r ** 2 + z ** 2
        d3 = d2 * np.sqrt(d2)
        accel = obj.thrust / m
        rhs = Dynamics(prob, section)
        rhs[0] = vr
        rhs[1] = vt / r
        rhs[2] = vz
        rhs[3] = vt ** 2 / r - r / d3 + accel * ur
        rhs[4] = -vr * vt / r + accel * ut
        rhs[5] = -z / d3 + accel * uz
        rhs[6] = -(obj.thrust / obj.ve) * (ur ** 2 + ut ** 2 + uz ** 2)
        return rhs()

    def equality(prob, obj):
        first = [(0, obj.r0), (1, 0.0), (2, 0.0), (3, obj.vr0), (4, obj.vt0), (5, 0.0), (6, obj.m0)]
        last = [(0, obj.rf), (2, 0.0), (3, obj.vrf), (4, obj.vt_target), (5, obj.vz_target)]
        rows = Condition()
        for state, value in first:
            rows.equal(prob.states_all_section(state)[0], value)
        for state, value in last:
            rows.equal(prob.states_all_section(state)[-1], value)
        return rows()

    def inequality(prob, obj):
        r = prob.states_all_section(0)
        m = prob.states_all_section(6)
        ur, ut, uz = (prob.controls_all_section(i) for i in range(3))
        tf = prob.time_final(-1)
        rows = Condition()
        rows.lower_bound(r, obj.r0)
        rows.lower_bound(m, obj.m_min)
        rows.lower_bound(tf, 0.0)
        rows.upper_bound(ur ** 2 + ut ** 2 + uz ** 2, 1.0)
        rows.upper_bound(tf, obj.tf_max)
        return rows()

    def cost(prob, obj):
        return 0.0

    def running_cost(prob, obj):
        ur, ut, uz = (prob.controls_all_section(i) for i in range(3))
        return obj.effort_scale * (ur ** 2 + ut ** 2 + uz ** 2)

    return dynamics, equality, inequality, cost, running_cost


def build(api, variant="3x4", nodes=None, max_iteration=10, effort_scale=None):
    obj = Spacecraft()
    if effort_scale is not None:
        obj.effort_scale = float(effort_scale)
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
        prob = api.Problem([0.0, 40.0], list(nodes or [200]), [7], [3], max_iteration)
        t = prob.time_all_section
        prob.set_states_all_section(0, G.linear(t, obj.r0, obj.rf))
        prob.set_states_all_section(1, G.linear(t, 0.0, 6.0))
        prob.set_states_all_section(2, G.linear(t, 0.0, 0.0))
        prob.set_states_all_section(3, G.cubic(t, obj.vr0, 0.02, obj.vrf, 0.0))
        prob.set_states_all_section(4, G.linear(t, obj.vt0, obj.vt_target))
        prob.set_states_all_section(5, G.linear(t, 0.0, obj.vz_target))
        prob.set_states_all_section(6, G.linear(t, obj.m0, 0.85))
        prob.set_controls_all_section(0, G.constant(t, 0.1))
        prob.set_controls_all_section(1, G.constant(t, 0.5))
        prob.set_controls_all_section(2, G.constant(t, 0.1))
        prob.set_states_bounds_all_section(6, obj.m_min, obj.m0)
        for control in range(3):
            prob.set_controls_bounds_all_section(control, -1.0, 1.0)
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
