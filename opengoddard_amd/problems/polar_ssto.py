"""Single-stage ascent to a 400 km circular orbit in polar coordinates, with physical units on every state, both
controls and time (maths of reference ``examples/08_Rocket_Ascent_Polar_SSTO.py:10-215``).  What it exercises beyond C3:
``set_unit_time`` (quirk Q8: the time grid and the final-time entry become non-dimensional, ``time_start(0)`` is divided
by the unit once more), rows scaled by hand with ``unit=`` (one of them with the WRONG state's unit, as shipped: the
final radius is scaled by theta's unit 1), a sliced operand ``m[1:]`` (quirk Q15), a row without any unit (the
acceleration limit in m/s^2) and variable bounds next to general inequalities.  The shipped script defines a cost
gradient but never assigns it: the cost gradient is differenced like everything else.

States (R, theta, Vr, Vt, m), controls (Tr, Tt):
    rho = 1.225 exp(-(R - Re) / 8500)          D_(r,t) = 0.5 rho V_(r,t) sqrt(Vr^2 + Vt^2) Cd A
    Rdot = Vr      thetadot = Vt / R           Vrdot = Tr/m - Dr/m - g0 (Re/R)^2 + Vt^2/R
    Vtdot = Tt/m - Dt/m - Vr Vt / R            mdot = -sqrt(Tr^2 + Tt^2) / g0 / Isp
30 LGL nodes on t in [0, 200] s; maximise the final mass.
"""
import numpy as np


class Launcher:
    GMe = 3.986004418 * 10 ** 14
    Re = 6371.0 * 1000
    g0 = 9.80665

    def __init__(self):
        self.Vr = np.sqrt(self.GMe / self.Re)
        self.H0 = 10.0
        self.V0 = 0.0
        self.M0 = 100000.0
        self.Mp = self.M0 * 0.99
        self.Cd = 0.6
        self.A = 4.0
        self.Isp = 300.0
        self.g0 = 9.80665
        self.Tmax = self.M0 * self.g0 * 1.5
        self.MaxQ = 14000.0
        self.MaxG = 8.0
        self.Htarget = 400.0 * 1000
        self.Rtarget = self.Re + self.Htarget
        self.Vtarget = np.sqrt(self.GMe / self.Rtarget)

    def air_density(self, h):
        beta = 1 / 8500.0
        rho0 = 1.225
        return rho0 * np.exp(-beta * h)


def make_callbacks(api):
    Condition, Dynamics = api.Condition, api.Dynamics

    def drag(obj, R, Vr, Vt):
        rho = obj.air_density(R - obj.Re)
        Dr = 0.5 * rho * Vr * np.sqrt(Vr ** 2 + Vt ** 2) * obj.Cd * obj.A
        Dt = 0.5 * rho * Vt * np.sqrt(Vr ** 2 + Vt ** 2) * obj.Cd * obj.A
        return Dr, Dt

    def dynamics(prob, obj, section):
        R, theta, Vr, Vt, m = (prob.states(i, section) for i in range(5))
        Tr, Tt = prob.controls(0, section), prob.controls(1, section)
        Dr, Dt = drag(obj, R, Vr, Vt)
        grav = obj.g0 * (obj.Re / R) ** 2
        rhs = Dynamics(prob, section)
        rhs[0] = Vr
        rhs[1] = Vt / R
        rhs[2] = Tr / m - Dr / m - grav + Vt ** 2 / R
        rhs[3] = Tt / m - Dt / m - (Vr * Vt) / R
        rhs[4] = -np.sqrt(Tr ** 2 + Tt ** 2) / obj.g0 / obj.Isp
        return rhs()

    def equality(prob, obj):
        R, theta, Vr, Vt, m = (prob.states_all_section(i) for i in range(5))
        u = prob.unit_states[0]
        rows = Condition()
        rows.equal(R[0], obj.Re, unit=u[0])
        rows.equal(theta[0], 0.0, unit=u[1])
        rows.equal(Vr[0], 0.0, unit=u[2])
        rows.equal(Vt[0], 0.0, unit=u[3])
        rows.equal(m[0], obj.M0, unit=u[4])
        rows.equal(R[-1], obj.Rtarget, unit=u[1])              # (as shipped: theta's unit, i.e. metres unscaled)
        rows.equal(Vr[-1], 0.0, unit=u[2])
        rows.equal(Vt[-1], obj.Vtarget, unit=u[3])
        return rows()

    def inequality(prob, obj):
        R, theta, Vr, Vt, m = (prob.states_all_section(i) for i in range(5))
        Tr, Tt = prob.controls_all_section(0), prob.controls_all_section(1)
        Dr, Dt = drag(obj, R, Vr, Vt)
        a_r = (Tr - Dr) / m
        a_t = (Tt - Dt) / m
        a_mag = np.sqrt(a_r ** 2 + a_t ** 2)
        T = np.sqrt(Tr ** 2 + Tt ** 2)
        rows = Condition()
        rows.lower_bound(m[1:], (obj.M0 - obj.Mp), unit=prob.unit_states[0][4])
        rows.lower_bound(Tt, 0.0, unit=prob.unit_controls[0][0])
        rows.upper_bound(m, obj.M0, unit=prob.unit_states[0][4])
        rows.upper_bound(T, obj.Tmax, unit=prob.unit_controls[0][0])
        rows.upper_bound(a_mag, obj.MaxG * obj.g0)
        return rows()

    def cost(prob, obj):
        return -prob.states_all_section(4)[-1] / prob.unit_states[0][4]

    return dynamics, equality, inequality, cost


def build(api, nodes=None, max_iteration=20):
    prob = api.Problem([0.0, 200], list(nodes or [30]), [5], [2], max_iteration)
    obj = Launcher()
    unit_R = obj.Re
    unit_V = np.sqrt(obj.GMe / obj.Re)
    unit_m = obj.M0
    unit_t = unit_R / unit_V
    unit_T = unit_m * unit_R / unit_t ** 2
    for state, unit in enumerate((unit_R, 1, unit_V, unit_V, unit_m)):
        prob.set_unit_states_all_section(state, unit)
    prob.set_unit_controls_all_section(0, unit_T)
    prob.set_unit_controls_all_section(1, unit_T)
    prob.set_unit_time(unit_t)
    G = api.Guess
    t = prob.time_all_section
    prob.set_states_all_section(0, G.cubic(t, obj.Re, 0.0, obj.Rtarget, 0.0))
    prob.set_states_all_section(1, G.cubic(t, 0.0, 0.0, np.deg2rad(25.0), 0.0))
    prob.set_states_all_section(2, G.linear(t, 0.0, 0.0))
    prob.set_states_all_section(3, G.linear(t, 0.0, obj.Vtarget))
    prob.set_states_all_section(4, G.cubic(t, obj.M0, -0.6, obj.M0 - obj.Mp, 0.0))
    prob.set_controls_all_section(0, G.cubic(t, obj.Tmax / 2, 0.0, 0.0, 0.0))
    prob.set_controls_all_section(1, G.cubic(t, obj.Tmax / 2, 0.0, 0.0, 0.0))
    prob.set_states_bounds_all_section(0, obj.Re, None)
    prob.set_controls_bounds_all_section(0, 0.0, obj.Tmax)
    prob.set_controls_bounds_all_section(1, 0.0, obj.Tmax)
    dynamics, equality, inequality, cost = make_callbacks(api)
    prob.dynamics = [dynamics]
    prob.knot_states_smooth = []
    prob.cost = cost
    prob.equality = equality
    prob.inequality = inequality
    return prob, obj
