"""C1 - brachistochrone (maths of reference
``examples/01_Brachistochrone_Problem.py:9-130``).

States (x, y, v), control theta; xdot = v sin(theta), ydot = v cos(theta), vdot = g cos(theta);
minimise the final time.  One phase, 20 LGL nodes, t in [0, 2].  The analytic optimum is
tf = sqrt(pi) for g = l = 1.
"""
import numpy as np


class Bead:
    gravity = 1.0
    goal_x = 1.0
    start_angle = np.deg2rad(30)


def make_callbacks(api):
    Condition, Dynamics = api.Condition, api.Dynamics

    def dynamics(prob, obj, section):
        v = prob.states(2, section)
        theta = prob.controls(0, section)
        rhs = Dynamics(prob, section)
        rhs[0] = v * np.sin(theta)
        rhs[1] = v * np.cos(theta)
        rhs[2] = obj.gravity * np.cos(theta)
        return rhs()

    def equality(prob, obj):
        x = prob.states_all_section(0)
        y = prob.states_all_section(1)
        v = prob.states_all_section(2)
        rows = Condition()
        rows.equal(x[0], 0.0)
        rows.equal(y[0], 0.0)
        rows.equal(v[0], 0.0)
        rows.equal(x[-1], obj.goal_x)
        return rows()

    def inequality(prob, obj):
        y = prob.states_all_section(1)
        theta = prob.controls_all_section(0)
        rows = Condition()
        rows.lower_bound(prob.time_final(-1), 0.1)
        rows.lower_bound(y, 0)
        rows.lower_bound(theta, 0)
        return rows()

    def cost(prob, obj):
        return prob.time_final(-1)

    def cost_derivative(prob, obj):
        grad = Condition(prob.number_of_variables)
        grad.change_value(prob.index_time_final(-1), 1)
        return grad()

    return dynamics, equality, inequality, cost, cost_derivative


def build(api, nodes=None, max_iteration=30):
    prob = api.Problem([0.0, 2.0], list(nodes or [20]), [3], [1], max_iteration)
    obj = Bead()
    t = prob.time_all_section
    prob.set_states_all_section(0, api.Guess.linear(t, 0.0, obj.goal_x))
    prob.set_states_all_section(1, api.Guess.linear(t, 0.0, obj.goal_x / np.sqrt(3)))
    prob.set_controls_all_section(0, api.Guess.linear(t, obj.start_angle, obj.start_angle))
    dynamics, equality, inequality, cost, cost_derivative = make_callbacks(api)
    prob.dynamics = [dynamics]
    prob.knot_states_smooth = []
    prob.cost = cost
    prob.cost_derivative = cost_derivative
    prob.equality = equality
    prob.inequality = inequality
    return prob, obj
