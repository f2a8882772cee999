"""C5 - synthetic four-phase knotted launch: 4 phases, 8 states,
4 controls, 128 LGL nodes per phase, n = 6148 decision variables.  Not a shipped example; it
exists to put the sweep in the bandwidth-bound regime, to exercise both kinds of knot - the
built-in continuity rows (``knot_states_smooth`` True, quirk Q9) inside a stage and user-written
rows with a mass jump at the staging knot, the pattern of reference
``examples/09_Rocket_Ascent_Polar_TSTO.py:76-132`` - and, since round 4, to be an NLP that SLSQP
actually solves.

Vehicle: a two-stage polar ascent, two phases per stage, with an out-of-plane velocity component
and two integrator states.  States (R, theta, Vr, Vt, Vn, m, Q, Hq); controls (Tr, Tt, Tn, kappa)
where kappa in [0, 1] scales the drag area (1 + kappa).  Per phase i (its own Cd, A, Isp, Tmax):

    rho = 1.225 exp(-h/8500), h = max(R - Re, -100)       V2 = Vr^2 + Vt^2 + Vn^2
    k = 0.5 rho sqrt(V2) Cd_i A_i (1 + kappa)             g = g0 (Re/R)^2
    Rdot = Vr                thetadot = Vt / R
    Vrdot = Tr/m - k Vr/m - g + (Vt^2 + Vn^2)/R
    Vtdot = Tt/m - k Vt/m - Vr Vt/R
    Vndot = Tn/m - k Vn/m - Vr Vn/R
    mdot = -sqrt(Tr^2 + Tt^2 + Tn^2) / g0 / Isp_i
    Qdot = 0.5 rho V2                                     Hqdot = c_q sqrt(rho) V2 sqrt(V2)

Knots: phase 0 -> 1 and 2 -> 3 are smooth (all eight states continuous, built-in rows; the knot
times are pinned to fixed fractions of the stage's burn - a free knot inside a continuous arc would
be a direction the NLP does not see); phase 1 -> 2 is the staging: seven states continuous by
user rows, the mass restarts at the second stage's ignition mass.  Path constraints: thrust
magnitude, acceleration (MaxG), dynamic pressure (MaxQ), R >= Re, burn durations of both stages
bounded above.  Cost: the control effort 100 sum w (Tr^2 + Tt^2 + Tn^2) / unit_T^2 as a running cost
(raw LGL weights, quirk Q10) - a smooth, strictly convex function of the controls, weighted such that its
Hessian in the scaled controls is of the order of the identity SLSQP's quasi-Newton matrix starts from.

Round 1-3's form of this problem (one continuous vehicle of 60 t with Isp 280-350 s to a 400 km
orbit, all knots smooth and free, final mass as the cost) was not a well-posed NLP: its ideal
delta-v is short of orbit unless the mass runs to its lower bound, the three free knot times of a
continuous arc are flat directions, and the cost's only gradient entry sits on one variable - SLSQP
left the guess with steps that overflowed after a handful of iterations (NaN cost from the 5th-8th
iteration on, in SciPy-faithful restatement and HIP core alike, VERDICT r3 missing #1).
"""
import numpy as np


class Stack:
    GMe = 3.986004418 * 10 ** 14
    Re = 6371.0 * 1000
    g0 = 9.80665

    def __init__(self):
        # per phase: stage 1 flies phases 0-1, stage 2 phases 2-3
        self.M0 = [60000.0, 60000.0, 9000.0, 9000.0]          # mass at the stage's ignition
        self.Mmin = [10000.0, 10000.0, 1000.0, 1000.0]        # burn-out mass of the stack in that phase
        self.Isp = [290.0, 310.0, 345.0, 345.0]
        self.Tmax = [60000.0 * self.g0 * 1.5, 60000.0 * self.g0 * 1.5, 9000.0 * self.g0, 9000.0 * self.g0]
        self.Cd = [0.25, 0.22, 0.2, 0.2]
        self.A = [7.0, 7.0, 3.14, 3.14]
        self.MaxG = 6.0
        self.MaxQ = 45000.0
        self.c_q = 1.7415e-4
        self.Rtarget = self.Re + 300.0 * 1000
        self.Vtarget = np.sqrt(self.GMe / self.Rtarget)
        self.unit_Q = 1.0e7
        self.unit_H = 1.0e9
        self.burn_max = [500.0, 660.0]                          # longest burn of stage 1 / stage 2, seconds
        self.knot_fraction = [0.4, 270.0 / 660.0]               # where a stage's inner knot sits in its burn
        # weight of the cost.  SLSQP starts its quasi-Newton matrix from the identity; the Hessian of the cost in the
        # scaled controls is 2 effort_scale w_i with LGL weights w_i of 128 nodes between 1.2e-4 and 2.5e-2, so with a
        # weight of 1 the identity is 40 ... 8 000 times too stiff in every control direction and the dense BFGS update
        # has 2 014 free directions to correct one by one (measured on the MI355X, n = 6148, ftol 1e-6: weight 1 - no
        # exit mode 0 in 12 000 major iterations; 10 - 5 109 iterations; 100 - 2 018 iterations, 160 s)
        self.effort_scale = 100.0

    def air_density(self, h):
        h[h < -100.0] = -100.0
        return 1.225 * np.exp(-(1 / 8500.0) * h)


N_PHASE = 4
CONTINUOUS_AT_STAGING = (0, 1, 2, 3, 4, 6, 7)                   # every state but the mass


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
        first = [(0, obj.Re), (1, 0.0), (2, 0.0), (3, 0.0), (4, 0.0), (5, obj.M0[0]), (6, 0.0), (7, 0.0)]
        last = [(0, obj.Rtarget), (2, 0.0), (3, obj.Vtarget), (4, 0.0)]
        rows = Condition()
        for state, value in first:
            rows.equal(prob.states(state, 0)[0], value, unit=u[state])
        for state, value in last:
            rows.equal(prob.states(state, N_PHASE - 1)[-1], value, unit=u[state])
        # the inner knot of each stage at a fixed fraction of the stage's burn
        t1, t2, t3, t4 = (prob.time_final(i) for i in range(N_PHASE))
        rows.equal(t1, obj.knot_fraction[0] * t2, unit=prob.unit_time)
        rows.equal(t3 - t2, obj.knot_fraction[1] * (t4 - t2), unit=prob.unit_time)
        # staging (knot 1 -> 2): everything but the mass is continuous, the second stage ignites at its own mass
        for state in CONTINUOUS_AT_STAGING:
            rows.equal(prob.states(state, 2)[0], prob.states(state, 1)[-1], unit=u[state])
        rows.equal(prob.states(5, 2)[0], obj.M0[2], unit=u[5])
        return rows()

    def inequality(prob, obj):
        rows = Condition()
        rows.lower_bound(prob.states_all_section(0), obj.Re, unit=prob.unit_states[0][0])
        rows.upper_bound(prob.time_final(1), obj.burn_max[0], unit=prob.unit_time)
        rows.upper_bound(prob.time_final(3) - prob.time_final(1), obj.burn_max[1], unit=prob.unit_time)
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
        return 0.0

    def running_cost(prob, obj):
        u = prob.unit_controls[0][0]
        Tr, Tt, Tn = (prob.controls_all_section(c) for c in range(3))
        return obj.effort_scale * (Tr ** 2 + Tt ** 2 + Tn ** 2) / u ** 2

    return dynamics, equality, inequality, cost, running_cost


def build(api, nodes=None, max_iteration=5, effort_scale=None):
    nodes = list(nodes or [128] * N_PHASE)
    assert len(nodes) == N_PHASE
    obj = Stack()
    if effort_scale is not None:
        obj.effort_scale = float(effort_scale)
    t_stage = obj.burn_max[0]
    t_end = t_stage + obj.burn_max[1]
    knots = [0.0, obj.knot_fraction[0] * t_stage, t_stage, t_stage + obj.knot_fraction[1] * obj.burn_max[1], t_end]
    prob = api.Problem(knots, nodes, [8] * N_PHASE, [4] * N_PHASE, max_iteration)
    G = api.Guess
    unit_R = obj.Re
    unit_V = np.sqrt(obj.GMe / obj.Re)
    unit_m = obj.M0[0]
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
    prob.set_states_all_section(2, G.linear(t, 0.0, 0.0))
    prob.set_states_all_section(3, G.linear(t, 0.0, obj.Vtarget))
    prob.set_states_all_section(4, G.linear(t, 0.0, 0.0))
    mass = [(60000.0, 40000.0), (40000.0, 14000.0), (9000.0, 5000.0), (5000.0, 2000.0)]
    prob.set_states_all_section(5, np.hstack([G.linear(prob.time[i], *mass[i]) for i in range(N_PHASE)]))
    prob.set_states_all_section(6, G.linear(t, 0.0, 0.3 * obj.unit_Q))
    prob.set_states_all_section(7, G.linear(t, 0.0, 0.2 * obj.unit_H))
    for control, share in ((0, 0.6), (1, 0.6), (2, 0.0)):
        profile = np.hstack([G.cubic(prob.time[i], obj.Tmax[i] * share, 0.0, obj.Tmax[i] * share * 0.5,
                                     0.0) for i in range(N_PHASE)])
        prob.set_controls_all_section(control, profile)
    prob.set_controls_all_section(3, G.linear(t, 0.2, 0.1))

    prob.set_states_bounds_all_section(0, obj.Re, None)
    for i in range(N_PHASE):
        prob.set_states_bounds(5, i, obj.Mmin[i], obj.M0[i])
        for control in range(3):
            prob.set_controls_bounds(control, i, -obj.Tmax[i], obj.Tmax[i])
    prob.set_controls_bounds_all_section(3, 0.0, 1.0)

    dynamics, equality, inequality, cost, running_cost = make_callbacks(api)
    prob.dynamics = [dynamics] * N_PHASE
    prob.knot_states_smooth = [True, False, True]
    prob.cost = cost
    prob.running_cost = running_cost
    prob.equality = equality
    prob.inequality = inequality
    return prob, obj
