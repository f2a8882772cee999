"""C3' / C3 - two-stage ascent to orbit in polar coordinates.

``variant="5x2"`` is the shape the reference ships (SURVEY.md Appendix D3'; maths of reference
``examples/09_Rocket_Ascent_Polar_TSTO.py:10-325``): states (R, theta, Vr, Vt, m), controls
(Tr, Tt), two phases joined by user-written knot equalities (``knot_states_smooth=[False]``).

``variant="6x3"`` is BASELINE.json's "2-phase, 6-state, 3-control, 80-node" configuration
(SURVEY.md section 8(d), C3): the same vehicle plus a sixth state Q, the dynamic-pressure
integral Qdot = rho (Vr^2 + Vt^2) / 2, and a third thrust component Tn that only enters the
propellant flow mdot = -sqrt(Tr^2 + Tt^2 + Tn^2) / (g0 Isp) and the thrust-magnitude bounds.
Units, bounds and the guess recipe follow the shipped example.
"""
import numpy as np


class Launcher:
    GMe = 3.986004418 * 10 ** 14
    Re = 6371.0 * 1000
    g0 = 9.80665

    def __init__(self):
        self.M0 = [20000.0, 1000.0]
        self.Mdry = [2000, 200]
        self.Cd = [0.2, 0.2]
        self.A = [3.14, 3.14]
        self.Isp = [300.0, 350.0]
        self.Tmax = [self.M0[0] * self.g0 * 1.5, self.M0[1] * self.g0 * 1.5]
        self.MaxG = 8.0
        self.Htarget = 500.0 * 1000
        self.Rtarget = self.Re + self.Htarget
        self.Vtarget = np.sqrt(self.GMe / self.Rtarget)
        self.unit_Q = 1.0e6

    def air_density(self, h):
        h[h < -100.0] = -100.0          # clamp below the surface (in place on a temporary)
        return 1.225 * np.exp(-(1 / 8500.0) * h)


def make_callbacks(api, with_q):
    Condition, Dynamics = api.Condition, api.Dynamics

    def thrust_mag(prob, section_or_none):
        if section_or_none is None:
            Tr, Tt = prob.controls_all_section(0), prob.controls_all_section(1)
            Tn = prob.controls_all_section(2) if with_q else None
        else:
            Tr, Tt = prob.controls(0, section_or_none), prob.controls(1, section_or_none)
            Tn = prob.controls(2, section_or_none) if with_q else None
        if with_q:
            return np.sqrt(Tr ** 2 + Tt ** 2 + Tn ** 2)
        return np.sqrt(Tr ** 2 + Tt ** 2)

    def dynamics(prob, obj, section):
        R = prob.states(0, section)
        Vr = prob.states(2, section)
        Vt = prob.states(3, section)
        m = prob.states(4, section)
        Tr = prob.controls(0, section)
        Tt = prob.controls(1, section)
        rho = obj.air_density(R - obj.Re)
        Dr = 0.5 * rho * Vr * np.sqrt(Vr ** 2 + Vt ** 2) * obj.Cd[section] * obj.A[section]
        Dt = 0.5 * rho * Vt * np.sqrt(Vr ** 2 + Vt ** 2) * obj.Cd[section] * obj.A[section]
        grav = obj.g0 * (obj.Re / R) ** 2
        rhs = Dynamics(prob, section)
        rhs[0] = Vr
        rhs[1] = Vt / R
        rhs[2] = Tr / m - Dr / m - grav + Vt ** 2 / R
        rhs[3] = Tt / m - Dt / m - (Vr * Vt) / R
        rhs[4] = -thrust_mag(prob, section) / obj.g0 / obj.Isp[section]
        if with_q:
            rhs[5] = 0.5 * rho * (Vr ** 2 + Vt ** 2)
        return rhs()

    def equality(prob, obj):
        Vr = prob.states_all_section(2)
        Vt = prob.states_all_section(3)
        R0, R1 = prob.states(0, 0), prob.states(0, 1)
        th0, th1 = prob.states(1, 0), prob.states(1, 1)
        Vr0, Vr1 = prob.states(2, 0), prob.states(2, 1)
        Vt0, Vt1 = prob.states(3, 0), prob.states(3, 1)
        m0, m1 = prob.states(4, 0), prob.states(4, 1)
        uR, uV, uM = prob.unit_states[0][0], prob.unit_states[0][2], prob.unit_states[0][4]
        rows = Condition()
        rows.equal(R0[0], obj.Re, unit=uR)
        rows.equal(th0[0], 0.0)
        rows.equal(Vr0[0], 0.0, unit=uV)
        rows.equal(Vt0[0], 0.0, unit=uV)
        rows.equal(m0[0], obj.M0[0], unit=uM)
        rows.equal(m1[0], obj.M0[1], unit=uM)
        rows.equal(R1[-1], obj.Rtarget, unit=uR)
        rows.equal(Vr[-1], 0.0, unit=uV)
        rows.equal(Vt[-1], obj.Vtarget, unit=uV)
        rows.equal(R1[0], R0[-1], unit=uR)
        rows.equal(th1[0], th0[-1])
        rows.equal(Vr1[0], Vr0[-1], unit=uV)
        rows.equal(Vt1[0], Vt0[-1], unit=uV)
        if with_q:
            Q0, Q1 = prob.states(5, 0), prob.states(5, 1)
            rows.equal(Q0[0], 0.0, unit=prob.unit_states[0][5])
            rows.equal(Q1[0], Q0[-1], unit=prob.unit_states[0][5])
        return rows()

    def inequality(prob, obj):
        R = prob.states_all_section(0)
        Vr = prob.states_all_section(2)
        Vt = prob.states_all_section(3)
        m = prob.states_all_section(4)
        Tr = prob.controls_all_section(0)
        Tt = prob.controls_all_section(1)
        rho = obj.air_density(R - obj.Re)
        a_mag = []
        for stage in (0, 1):
            Dr = 0.5 * rho * Vr * np.sqrt(Vr ** 2 + Vt ** 2) * obj.Cd[stage] * obj.A[stage]
            Dt = 0.5 * rho * Vt * np.sqrt(Vr ** 2 + Vt ** 2) * obj.Cd[stage] * obj.A[stage]
            a_r = (Tr - Dr) / m
            a_t = (Tt - Dt) / m
            a_mag.append(np.sqrt(a_r ** 2 + a_t ** 2))
        rows = Condition()
        rows.lower_bound(R, obj.Re, unit=prob.unit_states[0][0])
        rows.upper_bound(thrust_mag(prob, 0), obj.Tmax[0], unit=prob.unit_controls[0][0])
        rows.upper_bound(thrust_mag(prob, 1), obj.Tmax[1], unit=prob.unit_controls[0][0])
        rows.upper_bound(a_mag[0], obj.MaxG * obj.g0)
        rows.upper_bound(a_mag[1], obj.MaxG * obj.g0)
        return rows()

    def cost(prob, obj):
        return -prob.states(4, 1)[-1] / prob.unit_states[1][4]

    return dynamics, equality, inequality, cost


def build(api, variant="5x2", nodes=None, max_iteration=40):
    with_q = variant == "6x3"
    if variant not in ("5x2", "6x3"):
        raise ValueError("variant must be '5x2' or '6x3'")
    nodes = list(nodes or ([80, 80] if with_q else [20, 20]))
    ns, nc = (6, 3) if with_q else (5, 2)
    prob = api.Problem([0.0, 100, 200], nodes, [ns, ns], [nc, nc], max_iteration)
    obj = Launcher()
    G = api.Guess

    unit_R = obj.Re
    unit_V = np.sqrt(obj.GMe / obj.Re)
    unit_m = obj.M0[0]
    unit_t = unit_R / unit_V
    unit_T = unit_m * unit_R / unit_t ** 2
    for state, unit in enumerate([unit_R, 1, unit_V, unit_V, unit_m]):
        prob.set_unit_states_all_section(state, unit)
    if with_q:
        prob.set_unit_states_all_section(5, obj.unit_Q)
    for control in range(nc):
        prob.set_unit_controls_all_section(control, unit_T)
    prob.set_unit_time(unit_t)

    t = prob.time_all_section
    prob.set_states_all_section(0, G.cubic(t, obj.Re, 0.0, obj.Rtarget, 0.0))
    prob.set_states_all_section(1, G.cubic(t, 0.0, 0.0, np.deg2rad(25.0), 0.0))
    prob.set_states_all_section(2, G.linear(t, 0.0, 0.0))
    prob.set_states_all_section(3, G.linear(t, 0.0, obj.Vtarget))
    # the shipped script stacks two all-phase mass profiles and so only ever uses the first
    prob.set_states_all_section(4, G.cubic(t, obj.M0[0], -0.6, obj.Mdry[0], 0.0))
    thrust = np.hstack([G.cubic(prob.time[i], obj.Tmax[i] * 9 / 10, 0.0, 0.0, 0.0) for i in (0, 1)])
    prob.set_controls_all_section(0, thrust)
    prob.set_controls_all_section(1, thrust)      # the shipped script reuses the radial profile
    if with_q:
        prob.set_states_all_section(5, G.linear(t, 0.0, 0.2 * obj.unit_Q))
        prob.set_controls_all_section(2, 0.05 * thrust)

    prob.set_states_bounds_all_section(0, obj.Re, None)
    prob.set_states_bounds(4, 0, obj.Mdry[0], obj.M0[0])
    prob.set_states_bounds(4, 1, 1.0, obj.M0[1])
    for control in range(nc):
        prob.set_controls_bounds(control, 0, -obj.Tmax[1], obj.Tmax[0])
        prob.set_controls_bounds(control, 1, -obj.Tmax[1], obj.Tmax[1])

    dynamics, equality, inequality, cost = make_callbacks(api, with_q)
    prob.dynamics = [dynamics, dynamics]
    prob.knot_states_smooth = [False]
    prob.cost = cost
    prob.equality = equality
    prob.inequality = inequality
    return prob, obj
