"""Host-side mirror of ``OpenGoddard.optimize`` for the MI355X engine.

Same public surface as the reference module (``Problem``, ``Guess``, ``Condition``,
``Dynamics``; reference ``OpenGoddard/optimize.py:38-1127``) so that OpenGoddard problem scripts
run unmodified, but the hot path is different: ``Problem.solve`` traces the user callbacks once
(:mod:`opengoddard_amd.trace`), lowers them to HIP device code
(:mod:`opengoddard_amd.codegen`) and hands SciPy's SLSQP *analytic-looking* ``jac=`` callables
that are one batched forward-difference sweep on the GPU (:mod:`opengoddard_amd.engine`),
instead of letting SciPy call Python closures ``3n+2`` times per major iteration
(SURVEY.md section 3.3 / 8(b)).

The decision-vector layout, the getters' ``p*unit`` convention and the API quirks Q1-Q16 of
SURVEY.md Appendix A are reproduced on purpose; each method's docstring cites the reference
lines it mirrors.  There is no CPU evaluation path in ``solve``: without the HIP extension and
a GPU it raises.
"""
from __future__ import annotations

import os

import numpy as np

from . import trace as _tr

__all__ = ["Problem", "Guess", "Condition", "Dynamics"]

# Callable ``factory(prob, obj) -> engine`` used by ``Problem.solve``.  ``None`` selects the HIP
# engine.  The test-suite swaps in the NumPy oracle here to exercise host logic without a GPU
# (tests/conftest.py); nothing in this package ever sets it.
ENGINE_FACTORY = None
# 'scipy': SciPy's Fortran SLSQP driven by GPU callbacks (the reference's own core);
# 'hip': the same major iteration with the QP subproblem on the GPU (sqp.py, include/ogsqp.h);
# 'auto': 'hip' from AUTO_HIP_FROM decision variables on, 'scipy' below, and 'scipy' whenever the HIP core cannot take
# the problem (its capacity limits, no torch: Problem.sqp_core_fallback says why).  SciPy's core is O(n^3) per major
# iteration (16 s at n = 1442 against a few ms for the HIP core).  Where the HIP core overtakes it was measured in
# round 4 (tools/small_n_crossover.sh, profiles/r04_small_n.jsonl, same solves, MI355X box): n = 81 0.37 s vs 0.44 s
# (a tie: both are Python start-up), n = 201 0.85 vs 4.40 s, n = 281 0.50 vs 4.79 s, n = 701 0.82 vs 29.5 s - the HIP
# core from everything above the smallest shipped example, which keeps the reference's own arithmetic, iterate for
# iterate.  (Round 3 drew the line at 400.)
DEFAULT_SQP_CORE = "auto"
AUTO_HIP_FROM = 150


def _default_engine(prob, obj, devices=None):
    from .engine import HipEngine        # raises if the HIP library / a GPU is missing
    return HipEngine(prob, obj, devices=devices)


def _noop():
    pass


class Problem:
    """Multi-phase LGL pseudospectral transcription of an optimal-control problem.

    Args mirror the reference constructor (``optimize.py:759-823``): ``time_init`` is the list
    of phase boundary times ``[t0, t1, ..., tS]``; ``nodes``, ``number_of_states``,
    ``number_of_controls`` are per-phase lists; ``maxIterator`` bounds the SLSQP restart loop.
    ``method`` is accepted and ignored exactly like the reference (quirk Q1).

    Decision vector ``p`` (``optimize.py:237-245, 781``): for each phase, ``state0[N] ...
    state_{ns-1}[N] control0[N] ...``; phases concatenated; the last ``S`` entries are the phase
    final times.  Everything in ``p`` is divided by its canonical unit.
    """

    # ------------------------------------------------------------------ construction
    def __init__(self, time_init, nodes, number_of_states, number_of_controls,
                 maxIterator=100, method="LGL"):
        assert isinstance(time_init, list), "error: time_init is not list"
        assert isinstance(nodes, list), "error: nodes are not list"
        assert isinstance(number_of_states, list), "error: number of states are not list"
        assert isinstance(number_of_controls, list), "error: number of controls are not list"
        assert len(time_init) == len(nodes) + 1, "error: time_init length is not match nodes length"
        assert len(nodes) == len(number_of_states), "error: nodes length is not match states length"
        assert len(nodes) == len(number_of_controls), \
            "error: nodes length is not match controls length"
        from . import _native

        self.nodes = nodes
        self.number_of_states = number_of_states
        self.number_of_controls = number_of_controls
        self.number_of_section = len(nodes)
        self.number_of_param = np.array(number_of_states) + np.array(number_of_controls)
        self.div = self._make_param_division(nodes, number_of_states, number_of_controls)
        self.number_of_variables = int(sum(self.number_of_param * nodes)) + self.number_of_section

        self.tau, self.w, self.D, self.time = [], [], [], []
        for i, n in enumerate(nodes):
            if n < 3:   # quirk Q2: the reference's Gauss-Jacobi call rejects fewer than 3 nodes
                raise ValueError("n must be positive.")
            tau, w, D = _native.lgl(n)
            self.tau.append(tau)
            self.w.append(w)
            self.D.append(D)
            self.time.append(self._phase_grid(time_init[i], time_init[i + 1], tau))
        self.maxIterator = maxIterator
        self.iterator = 0
        self.time_init = time_init
        self.t0 = time_init[0]
        self.time_all_section = np.concatenate(self.time)

        self.unit_states = [[1.0] * ns for ns in number_of_states]
        self.unit_controls = [[1.0] * nc for nc in number_of_controls]
        self.unit_time = 1.0

        self.p = np.zeros(self.number_of_variables, dtype=float)
        self.bounds = [(None, None)] * self.number_of_variables
        for i in range(self.number_of_section):
            self.set_time_final_bounds(i, 0.0, None)

        self.dynamics = [None] * self.number_of_section
        self.knot_states_smooth = [True] * (self.number_of_section - 1)
        self.cost = None
        self.running_cost = None
        self.cost_derivative = None
        self.equality = None
        self.inequality = None
        for i in range(self.number_of_section):
            self.set_time_final(i, time_init[i + 1])

    @staticmethod
    def _phase_grid(ta, tb, tau):
        return (tb - ta) / 2.0 * tau + (tb + ta) / 2.0

    # LGL helpers kept under the reference's private names (``optimize.py:183-213``); the
    # numbers come from the native library (csrc/og_lgl.h), not from scipy.special.
    def _nodes_LGL(self, n):
        from . import _native
        return _native.lgl(n)[0]

    def _weight_LGL(self, n):
        from . import _native
        return _native.lgl(n)[1]

    def _differentiation_matrix_LGL(self, n):
        from . import _native
        return _native.lgl(n)[2]

    # ------------------------------------------------------------------ layout
    def _make_param_division(self, nodes, number_of_states, number_of_controls):
        """Cumulative slice ends per phase (``optimize.py:237-245``)."""
        ends, base = [], 0
        for n, ns, nc in zip(nodes, number_of_states, number_of_controls):
            row = [base + n * (k + 1) for k in range(ns + nc)]
            base = row[-1]
            ends.append(row)
        return ends

    def _division_states(self, state, section):
        """(back, front) of a state slice; negative indices leak through like the reference
        (``optimize.py:247-260``, quirk Q4)."""
        assert section < len(self.nodes), "section argument out of own section range"
        assert state < self.number_of_states[section], "states argument out of own states range"
        if state != 0:
            front = self.div[section][state - 1]
        elif section == 0:
            front = 0
        else:
            front = self.div[section - 1][-1]
        return self.div[section][state], front

    def _division_controls(self, control, section):
        """(back, front) of a control slice (``optimize.py:262-269``)."""
        assert section < len(self.nodes), "section argument out of own section range"
        assert control < self.number_of_controls[section], \
            "controls argument out of own controls range"
        at = self.number_of_states[section] + control
        return self.div[section][at], self.div[section][at - 1]

    def _tf_slot(self, section):
        return range(-self.number_of_section, 0)[section]

    # ------------------------------------------------------------------ getters
    def states(self, state, section):
        """State ``state`` of phase ``section`` in physical units (``optimize.py:271-284``)."""
        back, front = self._division_states(state, section)
        return self.p[front:back] * self.unit_states[section][state]

    def controls(self, control, section):
        """Control ``control`` of phase ``section`` in physical units (``optimize.py:302-315``)."""
        back, front = self._division_controls(control, section)
        return self.p[front:back] * self.unit_controls[section][control]

    def states_all_section(self, state):
        """All phases of one state, concatenated (``optimize.py:286-300``)."""
        return _tr.cat([self.states(state, i) for i in range(self.number_of_section)])

    def controls_all_section(self, control):
        """All phases of one control, concatenated (``optimize.py:317-331``)."""
        return _tr.cat([self.controls(control, i) for i in range(self.number_of_section)])

    def time_start(self, section):
        """Start time of a phase (``optimize.py:333-347``; quirk Q8 for phase 0)."""
        if section == 0:
            return self.t0
        slot = range(-self.number_of_section - 1, 0)[section]
        return self.p[slot] * self.unit_time

    def time_final(self, section):
        """Final time of a phase; negative ``section`` counts from the end (``optimize.py:349-360``)."""
        return self.p[self._tf_slot(section)] * self.unit_time

    def time_final_all_section(self):
        """List of all phase final times (``optimize.py:362-375``)."""
        return [self.time_final(i) for i in range(self.number_of_section)]

    # ------------------------------------------------------------------ setters
    def set_states(self, state, section, value):
        """``optimize.py:377-388`` (quirk Q6: length must equal the phase's node count)."""
        assert len(value) == self.nodes[section], "Error: value length is NOT match nodes length"
        back, front = self._division_states(state, section)
        self.p[front:back] = value / self.unit_states[section][state]

    def set_controls(self, control, section, value):
        """``optimize.py:404-415``."""
        assert len(value) == self.nodes[section], "Error: value length is NOT match nodes length"
        back, front = self._division_controls(control, section)
        self.p[front:back] = value / self.unit_controls[section][control]

    def _split_by_phase(self, value_all_section):
        at = 0
        for i, n in enumerate(self.nodes):
            yield i, value_all_section[at:at + n]
            at += n

    def set_states_all_section(self, state, value_all_section):
        """``optimize.py:390-402``."""
        for i, chunk in self._split_by_phase(value_all_section):
            self.set_states(state, i, chunk)

    def set_controls_all_section(self, control, value_all_section):
        """``optimize.py:417-429``."""
        for i, chunk in self._split_by_phase(value_all_section):
            self.set_controls(control, i, chunk)

    def set_time_final(self, section, value):
        """``optimize.py:431-440``."""
        self.p[self._tf_slot(section)] = value / self.unit_time

    # ------------------------------------------------------------------ bounds
    @staticmethod
    def _scaled_pair(lb, ub, unit, lb_default=None):
        lo = lb / unit if lb is not None else lb_default
        hi = ub / unit if ub is not None else None
        return lo, hi

    def set_states_bounds(self, state, section, lb, ub):
        """``optimize.py:442-455``."""
        pair = self._scaled_pair(lb, ub, self.unit_states[section][state])
        back, front = self._division_states(state, section)
        self.bounds[front:back] = [pair] * self.nodes[section]

    def set_states_bounds_all_section(self, state, lb, ub):
        """``optimize.py:457-467``."""
        for i in range(self.number_of_section):
            self.set_states_bounds(state, i, lb, ub)

    def set_controls_bounds(self, control, section, lb, ub):
        """``optimize.py:469-482``."""
        pair = self._scaled_pair(lb, ub, self.unit_controls[section][control])
        back, front = self._division_controls(control, section)
        self.bounds[front:back] = [pair] * self.nodes[section]

    def set_controls_bounds_all_section(self, control, lb, ub):
        """``optimize.py:484-494``."""
        for i in range(self.number_of_section):
            self.set_controls_bounds(control, i, lb, ub)

    def set_time_final_bounds(self, section, lb, ub):
        """``optimize.py:496-507``; a missing lower bound becomes 0.0, not -inf (quirk Q3)."""
        self.bounds[self.index_time_final(section)] = \
            self._scaled_pair(lb, ub, self.unit_time, lb_default=0.0)

    # ------------------------------------------------------------------ time helpers
    def time_to_tau(self, time):
        """Affine map of a time grid onto [-1, 1] (``optimize.py:509-516``)."""
        lo, hi = min(time), max(time)
        mid = (lo + hi) / 2
        return np.array([2 / (hi - lo) * (t - mid) for t in time])

    def time_update(self):
        """Rebuild the physical time grid from the optimised final times (``optimize.py:518-531``;
        starts from literal 0, quirk Q11)."""
        knots = [0] + self.time_final_all_section()
        self.time = [self._phase_grid(knots[i], knots[i + 1], self.tau[i])
                     for i in range(self.number_of_section)]
        return np.concatenate(self.time)

    def time_knots(self):
        """``optimize.py:533-540``."""
        return [0] + self.time_final_all_section()

    # ------------------------------------------------------------------ index helpers
    @staticmethod
    def _pick(front, back, index):
        if index is None:
            return front
        span = back - front
        assert index < span, "Error, index out of range"
        return front + (index + span if index < 0 else index)

    def index_states(self, state, section, index=None):
        """Position in ``p`` of a state sample (``optimize.py:542-559``, quirk Q5)."""
        back, front = self._division_states(state, section)
        return self._pick(front, back, index)

    def index_controls(self, control, section, index=None):
        """``optimize.py:561-568``."""
        back, front = self._division_controls(control, section)
        return self._pick(front, back, index)

    def index_time_final(self, section):
        """``optimize.py:570-572``."""
        return self.number_of_variables + self._tf_slot(section)

    # ------------------------------------------------------------------ canonical units
    def set_unit_states(self, state, section, value):
        """``optimize.py:579-588``."""
        self.unit_states[section][state] = value

    def set_unit_states_all_section(self, state, value):
        """``optimize.py:590-599``."""
        for i in range(self.number_of_section):
            self.unit_states[i][state] = value

    def set_unit_controls(self, control, section, value):
        """``optimize.py:601-610``."""
        self.unit_controls[section][control] = value

    def set_unit_controls_all_section(self, control, value):
        """``optimize.py:612-621``."""
        for i in range(self.number_of_section):
            self.unit_controls[i][control] = value

    def set_unit_time(self, value):
        """Switch to non-dimensional time (``optimize.py:623-639``, quirk Q8): ``time_init``,
        ``time``, ``t0`` and ``time_all_section`` become scaled, the tf entries of ``p`` are
        rewritten."""
        self.unit_time = value
        scaled = np.array(self.time_init) / value
        self.time_init = list(scaled)
        self.time = [self._phase_grid(scaled[i], scaled[i + 1], self.tau[i])
                     for i in range(self.number_of_section)]
        self.t0 = scaled[0]
        self.time_all_section = np.concatenate(self.time)
        for i in range(self.number_of_section):
            self.set_time_final(i, scaled[i + 1] * value)

    # ------------------------------------------------------------------ NLP assembly (traced once)
    def _collocation_derivative(self, section, vec):
        if _tr.is_sym(vec):
            return _tr.matvec(section, vec)
        return self.D[section].dot(vec)

    def _assemble_equality(self, obj):
        """User equalities, collocation defects ``D x - (tf-t0)/2 f``, knot continuity, in the
        reference's row order and operation order (``optimize.py:670-698``; quirks Q7, Q9)."""
        blocks = [self.equality(self, obj)]
        for i in range(self.number_of_section):
            deriv = [self._collocation_derivative(i, self.states(j, i) / self.unit_states[i][j])
                     for j in range(self.number_of_states[i])]
            t_a = self.time_start(i) / self.unit_time
            t_b = self.time_final(i) / self.unit_time
            rhs = self.dynamics[i](self, obj, i)
            blocks.append(_tr.cat(deriv) - (t_b - t_a) / 2.0 * rhs)
        for knot in range(self.number_of_section - 1):
            if self.number_of_states[knot] != self.number_of_states[knot + 1]:
                continue
            for s in range(self.number_of_states[knot]):
                left = self.states(s, knot) / self.unit_states[knot][s]
                right = self.states(s, knot + 1) / self.unit_states[knot][s]
                if self.knot_states_smooth[knot]:
                    blocks.append(left[-1] - right[0])
        return _tr.cat(blocks)

    def _assemble_cost(self, obj):
        """Mayer term plus LGL quadrature of the running cost with the raw weights, no
        ``(tf-t0)/2`` factor (``optimize.py:700-709``, quirk Q10)."""
        mayer = self.cost(self, obj)
        if self.running_cost is None:
            return mayer
        integrand = self.running_cost(self, obj) * np.concatenate(self.w)
        if _tr.is_sym(integrand):
            return mayer + _tr.seqsum(integrand)
        return mayer + sum(integrand)

    # ------------------------------------------------------------------ solve
    def solve(self, obj, display_func=_noop, **options):
        """Run the SLSQP restart loop (``optimize.py:649-755``) with GPU-evaluated callbacks.

        Options honoured are the reference's: ``ftol`` (1e-6) and ``maxiter`` (25); two more select what
        the reference does not have: ``sqp_core`` ("auto", the default: the HIP SQP core - QP subproblems on the GPU,
        sqp.py - from 150 decision variables on, SciPy's Fortran core below; "hip" / "scipy" force one) and
        ``jacobian="exact"`` (forward-mode derivatives of the traced callbacks instead of SciPy's
        forward differences; same optimum, fewer iterations, no FD noise).  Cost,
        equality and inequality values come from a single-column launch of the sweep kernel;
        the three Jacobians SLSQP asks for at each major iteration come from one
        forward-difference sweep over all ``n`` decision-vector columns, with SciPy's step
        rule.  A user ``cost_derivative`` is used as-is, like the
        reference does (``optimize.py:730-733``).
        """
        from scipy import optimize as _sciopt

        assert len(self.dynamics) != 0, "It must be set dynamics"
        assert self.cost is not None, "It must be set cost function"
        assert self.equality is not None, "It must be set equality function"
        assert self.inequality is not None, "It must be set inequality function"

        core = options.pop("sqp_core", None) or os.environ.get("OG_SQP_CORE", DEFAULT_SQP_CORE)
        if core not in ("scipy", "hip", "auto"):
            raise ValueError("sqp_core must be 'scipy', 'hip' or 'auto', got %r" % (core,))
        auto = core == "auto"
        if auto:
            # a stand-in engine of the test-suite (ENGINE_FACTORY) has no device-resident Jacobian for the HIP core
            core = "hip" if (self.number_of_variables >= AUTO_HIP_FROM and ENGINE_FACTORY is None) else "scipy"
        self.sqp_core_used = core
        self.sqp_core_fallback = None

        jacobian = options.pop("jacobian", None) or os.environ.get("OG_JACOBIAN", "fd")
        if jacobian not in ("fd", "exact"):
            raise ValueError("jacobian must be 'fd' or 'exact', got %r" % (jacobian,))

        # devices=[0, 1, ...]: the FD columns of every sweep are split over these GPUs of the node (one process,
        # RCCL all-gather of the packed non-zeros; include/ogpsx.h og_comm_init / og_multi_fd_sweep)
        devices = options.pop("devices", None)
        if devices is None and os.environ.get("OG_DEVICES"):
            devices = [int(v) for v in os.environ["OG_DEVICES"].split(",")]
        if devices is not None and len(devices) > 1 and ENGINE_FACTORY is not None:
            # not silently: a stand-in engine has no devices at all
            import warnings
            warnings.warn("Problem.solve: devices=%r is not used with a stand-in engine" % (list(devices),),
                          RuntimeWarning, stacklevel=2)
        if ENGINE_FACTORY is not None:
            engine = ENGINE_FACTORY(self, obj)
        else:
            engine = _default_engine(self, obj, devices=devices)
        self._engine = engine
        if core == "hip" and auto:
            # 'auto' never makes a problem unsolvable that the reference's core solves (slowly): the HIP core needs
            # torch for its device buffers and has capacity limits (include/ogsqp.h) - when it cannot be built for this
            # problem the solve goes to SciPy's core and says so (sqp_core_used / sqp_core_fallback)
            from . import sqp as _sqp
            reason = _sqp.prepare(engine)
            if reason is not None:
                import warnings
                warnings.warn("Problem.solve: the HIP SQP core is not available for this problem (%s); "
                              "using SciPy's SLSQP core" % reason, RuntimeWarning, stacklevel=2)
                core = self.sqp_core_used = "scipy"
                self.sqp_core_fallback = reason
        if jacobian == "exact":
            if not hasattr(engine, "exact_stacked"):
                raise ValueError("this engine has no exact-Jacobian mode")
            engine.jacobian_mode = "exact"
        lb = np.array([-np.inf if b[0] is None else b[0] for b in self.bounds], dtype=float)
        ub = np.array([np.inf if b[1] is None else b[1] for b in self.bounds], dtype=float)

        def value_of(which):
            def fun(p, prob, obj_):
                self.p = p
                return engine.values(p)[which]
            return fun

        def jacobian_of(which):
            def jac(p, prob, obj_):
                blocks, h = engine.jacobians(p, lb, ub)
                # quirk Q13: the reference leaves prob.p at the last FD column's input
                last = np.array(p, dtype=float, copy=True)
                last[-1] += h[-1]
                self.p = last
                return blocks[which]
            return jac

        cons = ({"type": "eq", "fun": value_of(1), "jac": jacobian_of(1), "args": (self, obj)},
                {"type": "ineq", "fun": value_of(2), "jac": jacobian_of(2), "args": (self, obj)})
        if self.cost_derivative is None:
            cost_jac = jacobian_of(0)
        else:
            def cost_jac(p, prob, obj_):
                self.p = p
                return self.cost_derivative(self, obj)

        ftol = options.setdefault("ftol", 1e-6)
        maxiter = options.setdefault("maxiter", 25)
        if core == "hip":
            from . import sqp as _sqp
            user_gradient = None
            if self.cost_derivative is not None:
                def user_gradient(p):
                    self.p = p
                    return self.cost_derivative(self, obj)
        while self.iterator < self.maxIterator:
            print("---- iteration : {0} ----".format(self.iterator + 1))
            if core == "hip":
                # same major iteration, QP subproblem and BFGS factor on the GPU (sqp.py)
                opt = _sqp.minimize_slsqp_hip(engine, self.p, lb, ub, ftol=ftol, maxiter=maxiter,
                                              cost_derivative=user_gradient, disp=True)
                self.p = opt.last_callback_p
                self.sqp_timing = opt.timing
                self.sqp_timings = getattr(self, "sqp_timings", []) + [opt.timing]
            else:
                opt = _sciopt.minimize(value_of(0), self.p, args=(self, obj), bounds=self.bounds,
                                       constraints=cons, jac=cost_jac, method="SLSQP",
                                       options={"disp": True, "maxiter": maxiter, "ftol": ftol})
            self.last_result = opt
            print(opt.message)
            display_func()
            print("")
            if not opt.status:
                break
            self.iterator += 1

    # ------------------------------------------------------------------ reporting
    def __repr__(self):
        rows = ["---- parameter ----",
                "nodes = %s" % (self.nodes,),
                "number of states    = %s" % (self.number_of_states,),
                "number of controls  = %s" % (self.number_of_controls,),
                "number of sections  = %s" % (self.number_of_section,),
                "number of variables = %s" % (self.number_of_variables,),
                "---- algorithm ----",
                "max iteration = %s" % (self.maxIterator,),
                "---- function  ----",
                "dynamics        = %s" % (self.dynamics,),
                "cost            = %s" % (self.cost,),
                "cost_derivative = %s" % (self.cost_derivative,),
                "equality        = %s" % (self.equality,),
                "inequality      = %s" % (self.inequality,),
                "knot_states_smooth = %s" % (self.dynamics,)]
        return "\n".join(rows) + "\n"

    def to_csv(self, filename="OpenGoddard_output.csv", delimiter=","):
        """Time, states and controls (phase-0 counts) as CSV columns (``optimize.py:844-863``)."""
        cols, names = [self.time_update()], ["time"]
        for i in range(self.number_of_states[0]):
            cols.append(self.states_all_section(i))
            names.append("state%d" % i)
        for i in range(self.number_of_controls[0]):
            cols.append(self.controls_all_section(i))
            names.append("control%d" % i)
        header = "".join(n + ", " for n in names)
        np.savetxt(filename, np.vstack(cols).T, delimiter=delimiter, header=header)
        print("Completed saving \"%s\"" % (filename))

    def plot(self, title_comment=""):
        """Scatter of the raw decision vector with slice boundaries (``optimize.py:865-880``)."""
        import matplotlib.pyplot as plt
        plt.figure()
        plt.title("OpenGoddard inner variables" + title_comment)
        plt.plot(self.p, "o")
        plt.xlabel("variables")
        plt.ylabel("value")
        for i in range(self.number_of_section):
            for edge in self.div[i]:
                plt.axvline(edge, color="C%d" % ((i + 1) % 6), alpha=0.5)
        plt.grid()


class Guess:
    """Initial-guess generators on a time grid (``optimize.py:883-975``)."""

    @classmethod
    def zeros(cls, time):
        return np.zeros(len(time))

    @classmethod
    def constant(cls, time, const):
        return np.ones(len(time)) * const

    @classmethod
    def linear(cls, time, y0, yf):
        """Straight line through ``(time[0], y0)`` and ``(time[-1], yf)``.  Same arithmetic as
        the two-point ``scipy.interpolate.interp1d`` the reference builds (``optimize.py:928-931``):
        ``slope * (t - t_first) + y0``."""
        time = np.asarray(time, dtype=float)
        slope = (np.float64(yf) - np.float64(y0)) / (time[-1] - time[0])
        return slope * (time - time[0]) + np.float64(y0)

    @classmethod
    def cubic(cls, time, y0, yprime0, yf, yprimef):
        """Cubic Hermite profile from end values and end slopes (``optimize.py:933-956``): the
        4x4 collocation system is inverted with ``np.linalg.inv`` like the reference."""
        ta, tb = time[0], time[-1]
        rows = []
        for t in (ta, tb):
            rows.append([1, t, t ** 2, t ** 3])
            rows.append([0, 1, 2 * t, 3 * t ** 2])
        coef = np.linalg.inv(np.array(rows)).dot(np.array([y0, yprime0, yf, yprimef]))
        return coef[0] + coef[1] * time + coef[2] * time ** 2 + coef[3] * time ** 3

    @classmethod
    def plot(cls, x, y, title="", xlabel="", ylabel=""):
        import matplotlib.pyplot as plt
        plt.figure()
        plt.plot(x, y, "-o")
        plt.title(title)
        plt.xlabel(xlabel)
        plt.ylabel(ylabel)
        plt.grid()


class Condition(object):
    """Growable constraint-row vector (``optimize.py:978-1072``).  ``equal(a, b)`` appends
    ``(a-b)/unit``; ``lower_bound(a, b)`` appends ``(a-b)/unit`` (feasible when >= 0);
    ``upper_bound(a, b)`` appends ``(b-a)/unit``."""

    def __init__(self, length=0):
        self._condition = np.zeros(length)

    def add(self, arg, unit=1.0):
        self._condition = _tr.cat([self._condition, arg / unit])

    def equal(self, arg1, arg2, unit=1.0):
        self.add(arg1 - arg2, unit)

    def lower_bound(self, arg1, arg2, unit=1.0):
        self.add(arg1 - arg2, unit)

    def upper_bound(self, arg1, arg2, unit=1.0):
        self.add(arg2 - arg1, unit)

    def change_value(self, index, value):
        self._condition[index] = value

    def __call__(self):
        return self._condition


class Dynamics(object):
    """Per-state right-hand-side holder for one phase (``optimize.py:1075-1127``).  Calling it
    stacks the states' time derivatives scaled by ``unit_time / unit_state``; states that were
    never assigned contribute zeros."""

    def __init__(self, prob, section=0):
        self.section = section
        self.number_of_state = prob.number_of_states[section]
        self.unit_states = prob.unit_states
        self.unit_time = prob.unit_time
        self._rhs = {i: np.zeros(prob.nodes[section]) for i in range(self.number_of_state)}

    def __getitem__(self, key):
        assert key < self.number_of_state, "Error, Dynamics key out of range"
        return self._rhs[key]

    def __setitem__(self, key, value):
        assert key < self.number_of_state, "Error, Dynamics key out of range"
        self._rhs[key] = value

    def __call__(self):
        units = self.unit_states[self.section]
        return _tr.cat([self._rhs[i] * (self.unit_time / units[i])
                        for i in range(self.number_of_state)])
