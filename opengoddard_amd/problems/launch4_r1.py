"""C5 - synthetic four-phase knotted launch: 4 phases, 8 states,
4 controls, 128 LGL nodes per phase, n = 6148 decision variables.  Not a shipped example; it
exists to put the sweep in the bandwidth-bound regime and to exercise the built-in knot rows
(``knot_states_smooth = [True, True, True]`` -> 24 continuity rows, quirk Q9).

Vehicle: polar ascent with an out-of-plane velocity component and two integrator states.
States (R, theta, Vr, Vt, Vn, m, Q, Hq); controls (Tr, Tt, Tn, kappa) where kappa in [0, 1]
scales the drag area (1 + kappa).  Per phase i (its own Cd, A, Isp):

    rho = 1.225 exp(-h/8500), h = max(R - Re, -100)       V2 = Vr^2 + Vt^2 + Vn^2
    k = 0.5 rho sqrt(V2) Cd_i A_i (1 + kappa)             g = g0 (Re/R)^2
    Rdot = Vr                thetadot = Vt / R
    Vrdot = Tr/m - k Vr/m - g + (Vt^2 + Vn^2)/R
    Vtdot = Tt/m - k Vt/m - Vr Vt/R
    Vndot = Tn/m - k Vn/m - Vr Vn/R
    mdot = -sqrt(Tr^2 + Tt^2 + Tn^2) / g0 / Isp_i
    Qdot = 0.5 rho V2                                     Hqdot = c_q sqrt(rho) V2 sqrt(V2)
"""
import numpy as np


class Stack:
    GMe = 3.986004418 * 10 ** 14
    Re = 6371.0 * 1000
    g0 = 9.80665

    def __init__(self):
        self.M0 = 60000.0
        self.Mfinal = 4000.0
        self.Cd = [0.25, 0.22, 0.2, 0.2]
        self.A = [7.0, 7.0, 3.14, 3.14]
        self.Isp = [280.0, 300.0, 330.0, 350.0]
        self.Tmax = [self.M0 * self.g0 * 1.4, self.M0 * self.g0 * 0.9,
                     self.M0 * self.g0 * 0.35, self.M0 * self.g0 * 0.1]
        self.MaxG = 6.0
        self.MaxQ = 45000.0
        self.c_q = 1.7415e-4
        self.Rtarget = self.Re + 400.0 * 1000
        self.Vtarget = np.sqrt(self.GMe / self.Rtarget)
        self.unit_Q = 1.0e7
        self.unit_H = 1.0e9

    def air_density(self, h):
        h[h < -100.0] = -100.0
        return 1.225 * np.exp(-(1 / 8500.0) * h)


N_PHASE = 4


def make_callbacks(api):
    Condition, Dynamics = api.Condition, api.Dynamics

    def flow(prob, obj, getter_s, getter_c, stage):
        R, Vr, Vt, Vn, m = (getter_s(i) for i in (0, 2, 3, 4, 5))
        Tr, Tt, Tn, kappa = (getter_c(i) for i in range(4))
        rho = obj.air_density(R - obj.Re)
        V2 = Vr ** 2 + Vt ** 2 + Vn ** 2
        k = 0.5 * rho * np.sqrt(V2) * obj.Cd[stage] * obj.A[stage] * (1.0 + kappa)
        return R, Vr, Vt, Vn, m, Tr, Tt, Tn, rho, V2, k

    def dynamics(prob, obj, section):
        R, Vr, Vt, Vn, m, Tr, Tt, Tn, rho, V2, k = flow(
            prob, obj, lambda s: prob.states(s, section), lambda c: prob.controls(c, section),
            section)
        grav = obj.g0 * (obj.Re / R) ** 2
        rhs = Dynamics(prob, section)
        rhs[0] = Vr
        rhs[1] = Vt / R
        rhs[2] = Tr / m - k * Vr / m - grav + (Vt ** 2 + Vn ** 2) / R
        rhs[3] = Tt / m - k * Vt / m - (Vr * Vt) / R
        rhs[4] = Tn / m - k * Vn / m - (Vr * Vn) / R
        rhs[5] = -np.sqrt(Tr ** 2 + Tt ** 2 + Tn ** 2) / obj.g0 / obj.Isp[section]
        rhs[6] = 0.5 * rho * V2
        rhs[7] = obj.c_q * np.sqrt(rho) * V2 * np.sqrt(V2)
        return rhs()

    def equality(prob, obj):
        u = prob.unit_states[0]
        first = [(0, obj.Re), (1, 0.0), (2, 0.0), (3, 0.0), (4, 0.0), (5, obj.M0), (6, 0.0), (7, 0.0)]
        last = [(0, obj.Rtarget), (2, 0.0), (3, obj.Vtarget), (4, 0.0)]
        rows = Condition()
        for state, value in first:
            rows.equal(prob.states(state, 0)[0], value, unit=u[state])
        for state, value in last:
            rows.equal(prob.states(state, N_PHASE - 1)[-1], value, unit=u[state])
        return rows()

    def inequality(prob, obj):
        rows = Condition()
        rows.lower_bound(prob.states_all_section(0), obj.Re, unit=prob.unit_states[0][0])
        for i in range(N_PHASE):
            Tr, Tt, Tn = (prob.controls(c, i) for c in range(3))
            rows.upper_bound(np.sqrt(Tr ** 2 + Tt ** 2 + Tn ** 2), obj.Tmax[i],
                             unit=prob.unit_controls[0][0])
        for i in range(N_PHASE):
            R, Vr, Vt, Vn, m, Tr, Tt, Tn, rho, V2, k = flow(
                prob, obj, lambda s: prob.states(s, i), lambda c: prob.controls(c, i), i)
            a_r = (Tr - k * Vr) / m
            a_t = (Tt - k * Vt) / m
            a_n = (Tn - k * Vn) / m
            rows.upper_bound(np.sqrt(a_r ** 2 + a_t ** 2 + a_n ** 2), obj.MaxG * obj.g0)
        for i in range(N_PHASE):
            R, Vr, Vt, Vn = (prob.states(s, i) for s in (0, 2, 3, 4))
            rho = obj.air_density(R - obj.Re)
            rows.upper_bound(0.5 * rho * (Vr ** 2 + Vt ** 2 + Vn ** 2), obj.MaxQ, unit=obj.MaxQ)
        return rows()

    def cost(prob, obj):
        return -prob.states(5, N_PHASE - 1)[-1] / prob.unit_states[N_PHASE - 1][5]

    return dynamics, equality, inequality, cost


def build(api, nodes=None, max_iteration=5):
    nodes = list(nodes or [128] * N_PHASE)
    assert len(nodes) == N_PHASE
    prob = api.Problem([0.0, 60.0, 150.0, 300.0, 520.0], nodes, [8] * N_PHASE, [4] * N_PHASE,
                       max_iteration)
    obj = Stack()
    G = api.Guess
    unit_R = obj.Re
    unit_V = np.sqrt(obj.GMe / obj.Re)
    unit_m = obj.M0
    unit_t = unit_R / unit_V
    unit_T = unit_m * unit_R / unit_t ** 2
    for state, unit in enumerate([unit_R, 1, unit_V, unit_V, unit_V, unit_m, obj.unit_Q, obj.unit_H]):
        prob.set_unit_states_all_section(state, unit)
    for control in range(3):
        prob.set_unit_controls_all_section(control, unit_T)
    prob.set_unit_controls_all_section(3, 1.0)
    prob.set_unit_time(unit_t)

    t = prob.time_all_section
    prob.set_states_all_section(0, G.cubic(t, obj.Re, 0.0, obj.Rtarget, 0.0))
    prob.set_states_all_section(1, G.cubic(t, 0.0, 0.0, np.deg2rad(20.0), 0.0))
    prob.set_states_all_section(2, G.cubic(t, 0.0, 900.0 * unit_t, 0.0, 0.0))
    prob.set_states_all_section(3, G.linear(t, 0.0, obj.Vtarget))
    prob.set_states_all_section(4, G.cubic(t, 0.0, 40.0 * unit_t, 0.0, 0.0))
    prob.set_states_all_section(5, G.cubic(t, obj.M0, -0.6, obj.Mfinal, 0.0))
    prob.set_states_all_section(6, G.linear(t, 0.0, 0.3 * obj.unit_Q))
    prob.set_states_all_section(7, G.linear(t, 0.0, 0.2 * obj.unit_H))
    for control, share in ((0, 0.8), (1, 0.55), (2, 0.05)):
        profile = np.hstack([G.cubic(prob.time[i], obj.Tmax[i] * share, 0.0, obj.Tmax[i] * share * 0.5,
                                     0.0) for i in range(N_PHASE)])
        prob.set_controls_all_section(control, profile)
    prob.set_controls_all_section(3, G.linear(t, 0.2, 0.1))

    prob.set_states_bounds_all_section(0, obj.Re, None)
    prob.set_states_bounds_all_section(5, obj.Mfinal * 0.5, obj.M0)
    for control in range(3):
        for i in range(N_PHASE):
            prob.set_controls_bounds(control, i, -obj.Tmax[i], obj.Tmax[i])
    prob.set_controls_bounds_all_section(3, 0.0, 1.0)

    dynamics, equality, inequality, cost = make_callbacks(api)
    prob.dynamics = [dynamics] * N_PHASE
    prob.knot_states_smooth = [True] * (N_PHASE - 1)
    prob.cost = cost
    prob.equality = equality
    prob.inequality = inequality
    return prob, obj
