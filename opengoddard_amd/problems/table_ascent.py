"""Single-stage polar ascent whose aerodynamics come from lookup tables - the pattern of the
reference's example 11 (``examples/11_Polar_TSTO_Taiki.py:14-27, 94-97``: atmosphere and drag
coefficient as ``scipy.interpolate.interp1d`` objects called inside the callbacks), with
synthetic tables so that it can travel to the GPU box.  SURVEY.md section 8(f) rank 3
("table-lookup dynamics").

States (R, theta, Vr, Vt, m), controls (Tr, Tt); density and sound speed are tabulated over
altitude with constant fill values outside the table, Cd is tabulated over Mach number and
extrapolated linearly.
"""
import numpy as np
from scipy import interpolate


class Vehicle:
    GMe = 3.986004418 * 10 ** 14
    Re = 6371.0 * 1000
    g0 = 9.80665

    def __init__(self):
        alt = np.concatenate([np.arange(0.0, 20e3, 1e3), np.arange(20e3, 90e3, 5e3)])
        density = 1.225 * np.exp(-alt / 7600.0) * (1.0 + 0.03 * np.sin(alt / 9000.0))
        sound = 340.3 - 45.0 * np.minimum(alt, 11e3) / 11e3 + 12.0 * np.clip((alt - 20e3) / 30e3, 0.0, 1.0)
        mach = np.array([0.0, 0.3, 0.6, 0.9, 1.0, 1.1, 1.3, 2.0, 3.0, 5.0])
        cd = np.array([0.28, 0.27, 0.29, 0.42, 0.55, 0.58, 0.50, 0.38, 0.30, 0.26])
        self.air_density = interpolate.interp1d(alt, density, bounds_error=False,
                                                fill_value=(density[0], 0.0))
        self.air_sound = interpolate.interp1d(alt, sound, bounds_error=False,
                                              fill_value=(sound[0], sound[-1]))
        self.drag_coefficient = interpolate.interp1d(mach, cd, fill_value="extrapolate")
        self.M0 = 5000.0
        self.Mdry = 600.0
        self.A = 1.2
        self.Isp = 320.0
        self.Tmax = self.M0 * self.g0 * 1.6
        self.MaxG = 7.0
        self.Rtarget = self.Re + 250.0 * 1000
        self.Vtarget = np.sqrt(self.GMe / self.Rtarget)


def make_callbacks(api):
    Condition, Dynamics = api.Condition, api.Dynamics

    def aero(prob, obj, R, Vr, Vt):
        h = R - obj.Re
        rho = obj.air_density(h)
        speed = np.sqrt(Vr ** 2 + Vt ** 2)
        cd = obj.drag_coefficient(speed / obj.air_sound(h))
        return 0.5 * rho * speed * cd * obj.A            # drag force per unit velocity

    def dynamics(prob, obj, section):
        R, Vr, Vt, m = (prob.states(i, section) for i in (0, 2, 3, 4))
        Tr, Tt = prob.controls(0, section), prob.controls(1, section)
        k = aero(prob, obj, R, Vr, Vt)
        grav = obj.g0 * (obj.Re / R) ** 2
        rhs = Dynamics(prob, section)
        rhs[0] = Vr
        rhs[1] = Vt / R
        rhs[2] = Tr / m - k * Vr / m - grav + Vt ** 2 / R
        rhs[3] = Tt / m - k * Vt / m - (Vr * Vt) / R
        rhs[4] = -np.sqrt(Tr ** 2 + Tt ** 2) / obj.g0 / obj.Isp
        return rhs()

    def equality(prob, obj):
        u = prob.unit_states[0]
        rows = Condition()
        for state, value in ((0, obj.Re), (1, 0.0), (2, 0.0), (3, 0.0), (4, obj.M0)):
            rows.equal(prob.states(state, 0)[0], value, unit=u[state])
        for state, value in ((0, obj.Rtarget), (2, 0.0), (3, obj.Vtarget)):
            rows.equal(prob.states(state, 0)[-1], value, unit=u[state])
        return rows()

    def inequality(prob, obj):
        R, Vr, Vt, m = (prob.states_all_section(i) for i in (0, 2, 3, 4))
        Tr, Tt = prob.controls_all_section(0), prob.controls_all_section(1)
        k = aero(prob, obj, R, Vr, Vt)
        a_r = (Tr - k * Vr) / m
        a_t = (Tt - k * Vt) / m
        rows = Condition()
        rows.lower_bound(R, obj.Re, unit=prob.unit_states[0][0])
        rows.lower_bound(m[1:], obj.Mdry, unit=prob.unit_states[0][4])
        rows.upper_bound(np.sqrt(Tr ** 2 + Tt ** 2), obj.Tmax, unit=prob.unit_controls[0][0])
        rows.upper_bound(np.sqrt(a_r ** 2 + a_t ** 2), obj.MaxG * obj.g0)
        rows.upper_bound(0.5 * obj.air_density(R - obj.Re) * (Vr ** 2 + Vt ** 2), 40000.0, unit=40000.0)
        return rows()

    def cost(prob, obj):
        return -prob.states(4, 0)[-1] / prob.unit_states[0][4]

    return dynamics, equality, inequality, cost


def build(api, nodes=None, max_iteration=20):
    prob = api.Problem([0.0, 420.0], list(nodes or [40]), [5], [2], max_iteration)
    obj = Vehicle()
    G = api.Guess
    unit_R = obj.Re
    unit_V = np.sqrt(obj.GMe / obj.Re)
    unit_m = obj.M0
    unit_t = unit_R / unit_V
    unit_T = unit_m * unit_R / unit_t ** 2
    for state, unit in enumerate([unit_R, 1, unit_V, unit_V, unit_m]):
        prob.set_unit_states_all_section(state, unit)
    prob.set_unit_controls_all_section(0, unit_T)
    prob.set_unit_controls_all_section(1, unit_T)
    prob.set_unit_time(unit_t)
    t = prob.time_all_section
    prob.set_states_all_section(0, G.cubic(t, obj.Re, 0.0, obj.Rtarget, 0.0))
    prob.set_states_all_section(1, G.cubic(t, 0.0, 0.0, np.deg2rad(18.0), 0.0))
    prob.set_states_all_section(2, G.cubic(t, 0.0, 700.0 * unit_t, 0.0, 0.0))
    prob.set_states_all_section(3, G.linear(t, 0.0, obj.Vtarget))
    prob.set_states_all_section(4, G.cubic(t, obj.M0, -0.6, obj.Mdry * 1.2, 0.0))
    prob.set_controls_all_section(0, G.cubic(t, obj.Tmax * 0.8, 0.0, obj.Tmax * 0.1, 0.0))
    prob.set_controls_all_section(1, G.cubic(t, obj.Tmax * 0.3, 0.0, obj.Tmax * 0.4, 0.0))
    prob.set_states_bounds_all_section(0, obj.Re, None)
    prob.set_states_bounds_all_section(4, obj.Mdry, obj.M0)
    prob.set_controls_bounds_all_section(0, -obj.Tmax, obj.Tmax)
    prob.set_controls_bounds_all_section(1, -obj.Tmax, obj.Tmax)
    dynamics, equality, inequality, cost = make_callbacks(api)
    prob.dynamics = [dynamics]
    prob.knot_states_smooth = []
    prob.cost = cost
    prob.equality = equality
    prob.inequality = inequality
    return prob, obj
