"""Lower a traced NLP (``trace.Graph``) to the device-function header the sweep kernels include.

Pipeline::

    trace_problem(prob, obj)          run cost / equality+defects+knots / inequality once on Sym
        -> Program                    rows of F = [cost | c_eq | c_ineq] as *pieces*
        -> emit_header(program)       C++17 text: struct OgGen { tables; group_eval; mv_operand }

A *piece* is ``(row_start, length, elem)``: rows ``row_start + k`` for ``k in [0, length)`` are
the element expression ``elem`` evaluated at ``k``.  Element expressions only have leaves that
are affine in ``k`` - ``p[base + stride*k]`` (stride 0 or 1), constant-table entries, and the
collocation products ``(D_phase @ operand)[k]`` that the kernel computes with MFMA - so a piece
is branch-free device code.  ``np.hstack`` of per-phase slices (``states_all_section``) simply
becomes several pieces.  Pieces of equal length are bundled into *groups* that share common
sub-expressions (e.g. all state derivatives of one phase share density / speed terms).

Row order is the reference's: cost; user equalities, defects (phase-major, state-major,
node), knot rows (``OpenGoddard/optimize.py:674-696``); user inequalities (``:727``).
"""
from __future__ import annotations

import hashlib

import numpy as np

from . import trace as _tr

import os

HEAVY_COLUMN_ELEMENTS = 32       # columns with more dependent elements get a whole workgroup
LIGHT_COLS = 4                   # columns per workgroup otherwise (csrc/ogk_kernels.hip OGK_LIGHT_COLS)


def fused_cols(n):
    """Columns per light workgroup of the fused launch: a run of neighbouring columns whose defect items lie in
    one (defect group, 16-node tile), i.e. at most the 16 nodes of one variable's tile.  Every light workgroup
    recomputes the base products of its tile, so fewer, wider workgroups mean fewer redundant MFMA chains.
    Measured with persistent-zero output (bench step, MI355X, 4 / 8 / 16 columns): C3 7.8 / 7.3 / 6.9 us,
    C4 15.8 / 13.8 / 12.4 us, C5 32.8 / 24.3 / 18.7 us.  (With the zero fill of round 1 in the same workgroups
    8 was the optimum and 16 lost.)  ``OG_FUSED_COLS`` overrides (timing experiments)."""
    env = os.environ.get("OG_FUSED_COLS")
    return int(env) if env else 16


def tile_cols():
    """Column tiles per MFMA-tile workgroup of the fused launch (at most SWEEP_WAVES - 1 = 7: the last wavefront
    computes the node tile's base products).  ``OG_TILE_COLS`` overrides (timing experiments)."""
    env = os.environ.get("OG_TILE_COLS")
    return max(1, min(7, int(env))) if env else 7


MAX_GROUP_OUTPUTS = int(os.environ.get("OG_MAX_GROUP_OUTPUTS", "1"))


# ------------------------------------------------------------------------------ element graph
class EGraph:
    def __init__(self):
        self.nodes = []
        self._index = {}

    def add(self, node):
        hit = self._index.get(node)
        if hit is not None:
            return hit
        self.nodes.append(node)
        self._index[node] = len(self.nodes) - 1
        return len(self.nodes) - 1


class MvSlot:
    """One collocation product ``D[phase] @ operand`` (operand: ``length`` elements)."""

    def __init__(self, phase, length, operand_eid, leaf_base):
        self.phase, self.length, self.operand, self.leaf_base = phase, length, operand_eid, leaf_base


class Group:
    def __init__(self, kind, length, phase=-1):
        self.kind = kind              # "rows" | "defect"
        self.length = length
        self.phase = phase
        self.outputs = []             # [(row_start, eid)]
        self.mv_slots = []            # slot ids feeding y[] (defect groups)
        self.tails = []               # defect groups: output s = Y_s - tails[s]
        self.deps = []                # [(kind, base, count)] decision-vector dependencies
        self.out_deps = []            # defect groups: the same, per output (per state)


class Program:
    def __init__(self):
        self.eg = EGraph()
        self.n = 0
        self.m_eq = 0
        self.m_ineq = 0
        self.nodes = []               # per-phase node counts
        self.pieces = []              # [(row_start, length, eid, kind)] kind in cost/eq/ineq
        self.mv = []                  # [MvSlot]
        self.groups = []
        self.cvec = np.zeros(0)       # flat constant table
        self.cvec_off = []            # table id -> offset into cvec
        self.tables = []              # linear lookup tables (trace.Graph.tables)
        self.table_len = []

    @property
    def m(self):
        return 1 + self.m_eq + self.m_ineq


class _Lowerer:
    """trace.Graph (vector nodes) -> EGraph (element nodes) + pieces."""

    def __init__(self, graph, program):
        self.g = graph
        self.P = program
        self.eg = program.eg
        self._pieces = {}
        self._fix = {}
        self._shift = {}
        off, at = [], 0
        for v in graph.cvecs:
            off.append(at)
            at += v.shape[0]
        program.cvec_off = off
        program.cvec = np.concatenate(graph.cvecs) if graph.cvecs else np.zeros(0)
        program.tables = list(graph.tables)
        program.table_len = [int(graph.cvecs[t[0]].shape[0]) for t in graph.tables]
        self._mv_index = {}

    # -- substitution helpers ----------------------------------------------------------------
    def _map(self, eid, leaf_fn, memo):
        hit = memo.get(eid)
        if hit is not None:
            return hit
        node = self.eg.nodes[eid]
        tag = node[0]
        if tag in ("P", "CV", "Y"):
            out = leaf_fn(node)
        elif tag in ("C", "sum"):
            out = eid
        elif tag in ("un", "interp"):
            out = self.eg.add((tag, node[1], self._map(node[2], leaf_fn, memo)))
        elif tag in ("bin", "cmp", "logic"):
            out = self.eg.add((tag, node[1], self._map(node[2], leaf_fn, memo),
                               self._map(node[3], leaf_fn, memo)))
        elif tag == "where":
            out = self.eg.add(("where",) + tuple(self._map(c, leaf_fn, memo) for c in node[1:]))
        else:
            raise AssertionError(tag)
        memo[eid] = out
        return out

    def fix(self, eid, i):
        """Substitute k := i (element becomes k-invariant)."""
        memo = self._fix.setdefault(i, {})

        def leaf(node):
            tag = node[0]
            if tag == "P":
                return self.eg.add(("P", node[1] + node[2] * i, 0))
            if tag == "CV":
                if node[3] == 0:
                    return self.eg.add(node)
                value = self.P.cvec[self.P.cvec_off[node[1]] + node[2] + i]
                return self.eg.add(("C", np.float64(value).tobytes()))
            return self.eg.add(("Y", node[1], node[2] + node[3] * i, 0))
        return self._map(eid, leaf, memo)

    def shift(self, eid, d):
        """Substitute k := k + d."""
        if d == 0:
            return eid
        memo = self._shift.setdefault(d, {})

        def leaf(node):
            tag = node[0]
            if tag == "P":
                return self.eg.add(("P", node[1] + node[2] * d, node[2]))
            if tag == "CV":
                return self.eg.add(("CV", node[1], node[2] + node[3] * d, node[3]))
            return self.eg.add(("Y", node[1], node[2] + node[3] * d, node[3]))
        return self._map(eid, leaf, memo)

    # -- vector nodes -> pieces ---------------------------------------------------------------
    def pieces(self, nid):
        """List of (length, eid); a scalar node gives [(None, eid)]."""
        hit = self._pieces.get(nid)
        if hit is not None:
            return hit
        node = self.g.nodes[nid]
        length = self.g.length[nid]
        tag = node[0]
        if tag == "p":
            out = [(node[2], self.eg.add(("P", node[1], 1)))]
        elif tag == "const":
            out = [(None, self.eg.add(("C", node[1])))]
        elif tag == "cvec":
            out = [(length, self.eg.add(("CV", node[1], 0, 1)))]
        elif tag in ("un", "interp"):
            out = [(ln, self.eg.add((tag, node[1], e))) for ln, e in self.pieces(node[2])]
        elif tag in ("bin", "cmp", "logic"):
            out = [(ln, self.eg.add((tag, node[1], a, b)))
                   for ln, (a, b) in self._align([node[2], node[3]], length)]
        elif tag == "where":
            out = [(ln, self.eg.add(("where",) + tuple(es)))
                   for ln, es in self._align(list(node[1:]), length)]
        elif tag == "idx":
            out = [(None, self._element(node[1], node[2]))]
        elif tag == "slice":
            out = self._subrange(node[1], node[2], node[3])
        elif tag == "cat":
            out = []
            for child in node[1]:
                for ln, e in self.pieces(child):
                    out.append((1 if ln is None else ln, e))
        elif tag == "mv":
            out = [(length, self.eg.add(("Y", self._mv_slot(node[1], node[2]), 0, 1)))]
        elif tag == "seqsum":
            body = tuple((1 if ln is None else ln, e) for ln, e in self.pieces(node[1]))
            out = [(None, self.eg.add(("sum", body)))]
        else:
            raise AssertionError(tag)
        self._pieces[nid] = out
        return out

    def _element(self, nid, i):
        if self.g.length[nid] is None:
            raise _tr.TraceError("indexing a scalar")
        at = 0
        for ln, e in self.pieces(nid):
            ln = 1 if ln is None else ln
            if i < at + ln:
                return self.fix(e, i - at)
            at += ln
        raise IndexError(i)

    def _subrange(self, nid, start, count):
        out, at = [], 0
        for ln, e in self.pieces(nid):
            ln = 1 if ln is None else ln
            lo, hi = max(start, at), min(start + count, at + ln)
            if lo < hi:
                out.append((hi - lo, self.shift(e, lo - at)))
            at += ln
        return out

    def _align(self, nids, length):
        """Common refinement of the operands' piece boundaries.  -> [(len, (eids...))]."""
        lists = []
        for nid in nids:
            pcs = self.pieces(nid)
            ln_total = self.g.length[nid]
            if ln_total is None:
                lists.append(("scalar", pcs[0][1]))
            elif ln_total == 1 and length not in (None, 1):
                lists.append(("scalar", self.fix(pcs[0][1], 0)))
            else:
                lists.append(("vec", [(1 if ln is None else ln, e) for ln, e in pcs]))
        if length is None:
            return [(None, tuple(item[1] for item in lists))]
        cuts = {0, length}
        for kind, val in lists:
            if kind == "vec":
                at = 0
                for ln, _ in val:
                    at += ln
                    cuts.add(at)
        cuts = sorted(cuts)
        out = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            es = []
            for kind, val in lists:
                if kind == "scalar":
                    es.append(val)
                    continue
                at = 0
                for ln, e in val:
                    if at <= lo < at + ln:
                        es.append(self.shift(e, lo - at))
                        break
                    at += ln
                else:
                    raise AssertionError("piece lookup")
            out.append((hi - lo, tuple(es)))
        return out

    def _mv_slot(self, phase, operand_nid):
        key = (phase, operand_nid)
        hit = self._mv_index.get(key)
        if hit is not None:
            return hit
        pcs = self.pieces(operand_nid)
        if len(pcs) != 1 or pcs[0][0] != self.P.nodes[phase]:
            raise _tr.TraceError("collocation operand must be one contiguous phase slice")
        operand = pcs[0][1]
        leaves = sorted(_leaves(self.eg, operand, ("P",)))
        strided = [lf for lf in leaves if lf[2] == 1]
        if len(strided) != 1 or len(leaves) != 1:
            raise _tr.TraceError("collocation operand must depend on exactly one state slice")
        self.P.mv.append(MvSlot(phase, pcs[0][0], operand, strided[0][1]))
        self._mv_index[key] = len(self.P.mv) - 1
        return len(self.P.mv) - 1


def _leaves(eg, eid, tags, seen=None, into_sums=True):
    """Set of leaf nodes with a tag in ``tags`` reachable from ``eid``."""
    seen = set() if seen is None else seen
    out = set()
    stack = [eid]
    while stack:
        e = stack.pop()
        if e in seen:
            continue
        seen.add(e)
        node = eg.nodes[e]
        tag = node[0]
        if tag in tags:
            out.add(node)
        if tag in ("un", "interp"):
            stack.append(node[2])
        elif tag in ("bin", "cmp", "logic"):
            stack.extend(node[2:4])
        elif tag == "where":
            stack.extend(node[1:])
        elif tag == "sum" and into_sums:
            for _, body in node[1]:
                stack.append(body)
    return out


# ------------------------------------------------------------------------------ tracing driver
def trace_problem(prob, obj):
    """Run the NLP assembly once on a symbolic decision vector and lower it to a Program."""
    n = int(prob.number_of_variables)
    sym_p = _tr.new_decision_vector(n)
    graph = sym_p.g
    saved = prob.p
    prob.p = sym_p
    try:
        with _tr.tracing(graph):
            cost = prob._assemble_cost(obj)
            ceq = prob._assemble_equality(obj)
            cineq = prob.inequality(prob, obj)
    finally:
        prob.p = saved

    probe = _tr.Sym(graph, graph.const(0.0))

    def as_sym(v, what):
        if _tr.is_sym(v):
            return v
        lifted = probe._lift(np.asarray(v, dtype=np.float64) if not np.isscalar(v) else v)
        if lifted is NotImplemented:
            raise _tr.TraceError("%s returned an untraceable %r" % (what, type(v)))
        return lifted

    P = Program()
    P.n = n
    P.nodes = [int(v) for v in prob.nodes]
    low = _Lowerer(graph, P)

    cost = as_sym(cost, "cost")
    if cost.length not in (None, 1):
        raise _tr.TraceError("cost must be a scalar")
    cost_eid = low.pieces(cost.id)[0][1]
    if cost.length == 1:
        cost_eid = low.fix(cost_eid, 0)
    P.pieces.append((0, 1, cost_eid, "cost"))

    row = 1
    for kind, value in (("eq", ceq), ("ineq", cineq)):
        start = row
        if isinstance(value, np.ndarray) and value.size == 0:
            pcs = []                                     # quirk Q14: empty constraint set
        else:
            value = as_sym(value, kind)
            pcs = low.pieces(value.id)
        for ln, e in pcs:
            ln = 1 if ln is None else ln
            P.pieces.append((row, ln, e, kind))
            row += ln
        if kind == "eq":
            P.m_eq = row - start
        else:
            P.m_ineq = row - start
    _make_groups(P)
    return P


def _make_groups(P):
    eg = P.eg
    defect = {}
    buckets = {}
    order = []
    for row, ln, e, kind in P.pieces:
        ys = _leaves(eg, e, ("Y",))
        if ys:
            phases = {P.mv[y[1]].phase for y in ys}
            if len(phases) != 1 or any(y[2] != 0 or y[3] != 1 for y in ys) or \
                    ln != P.nodes[next(iter(phases))]:
                raise _tr.TraceError(
                    "a defect row mixes the collocation product with a dynamics term that is not ONE expression over "
                    "all nodes of the phase (a right-hand side assembled from slices - rhs[:k] = ...; rhs[k:] = ... - "
                    "or a collocation product used outside the defect rows): write the right-hand side as one vector "
                    "expression (np.where for a switch over the nodes)")
            ph = next(iter(phases))
            if ph not in defect:
                defect[ph] = Group("defect", ln, ph)
                order.append(defect[ph])
            defect[ph].outputs.append((row, e))
            continue
        key = (kind, ln)
        grp = buckets.get(key)
        if grp is None or len(grp.outputs) >= MAX_GROUP_OUTPUTS:
            grp = Group("rows", ln)
            buckets[key] = grp
            order.append(grp)
        grp.outputs.append((row, e))
    for grp in order:
        if grp.kind == "defect":
            slots = set()
            for _, e in grp.outputs:
                slots |= {y[1] for y in _leaves(eg, e, ("Y",))}
            grp.mv_slots = sorted(slots)
            lo = grp.mv_slots[0]
            if grp.mv_slots != list(range(lo, lo + len(grp.mv_slots))):
                raise _tr.TraceError("collocation products of a phase must be consecutive")
            if len(grp.mv_slots) > 16:
                raise _tr.TraceError("more than 16 states per phase are not supported")
    P.groups = order
    for grp in order:
        if grp.kind == "defect":
            _split_defect_outputs(P, grp)
        grp.deps = _group_dependencies(P, grp)
        if grp.kind == "defect":
            grp.out_deps = [_group_dependencies(P, grp, [t]) for t in grp.tails]


def _split_defect_outputs(P, grp):
    """Defect rows must have the shape  Y_s[k] - T_s[k]  with T_s free of collocation products
    (that is what ``Problem._assemble_equality`` produces); the structured sweep re-evaluates
    T only where it depends on the perturbed variable."""
    eg = P.eg
    if len(grp.outputs) != len(grp.mv_slots):
        raise _tr.TraceError("defect group: one collocation product per state expected")
    for s, (_, e) in enumerate(grp.outputs):
        node = eg.nodes[e]
        ok = node[0] == "bin" and node[1] == "sub"
        if ok:
            y = eg.nodes[node[2]]
            ok = y[0] == "Y" and y[1] == grp.mv_slots[s] and y[2] == 0 and y[3] == 1 and \
                not _leaves(eg, node[3], ("Y",))
        if not ok:
            raise _tr.TraceError("defect row is not of the form  D x - (tf-t0)/2 f")
        grp.tails.append(node[3])


def _group_dependencies(P, grp, roots=None):
    """Which decision variables an element k of the group reads.
    kind 1: p[base + k]            (one column per element, ``count`` = group length)
    kind 0: p[base]                (one column, every element)
    kind 2: p[base .. base+count)  read inside a sum  (each column, every element)"""
    eg = P.eg
    if roots is None:
        roots = grp.tails if grp.kind == "defect" else [e for _, e in grp.outputs]
    deps = set()
    stack, seen = [(r, None) for r in roots], set()
    while stack:
        e, span = stack.pop()
        if (e, span) in seen:
            continue
        seen.add((e, span))
        node = eg.nodes[e]
        tag = node[0]
        if tag == "P":
            if node[2] == 0:
                deps.add((0, node[1], 1))
            elif span is None:
                deps.add((1, node[1], grp.length))
            else:
                deps.add((2, node[1], span))
        elif tag in ("un", "interp"):
            stack.append((node[2], span))
        elif tag in ("bin", "cmp", "logic"):
            stack.append((node[2], span))
            stack.append((node[3], span))
        elif tag == "where":
            for c in node[1:]:
                stack.append((c, span))
        elif tag == "sum":
            for ln, body in node[1]:
                stack.append((body, ln))
    return sorted(deps)


# ------------------------------------------------------------------------------ C++ emission
_UN_C = {"neg": "-(%s)", "sqrt": "ogm::sqrt_(%s)", "exp": "ogm::exp_(%s)", "log": "ogm::log_(%s)",
         "sin": "ogm::sin_(%s)", "cos": "ogm::cos_(%s)", "tan": "ogm::tan_(%s)",
         "abs": "ogm::fabs_(%s)", "atan": "ogm::atan_(%s)", "asin": "ogm::asin_(%s)",
         "acos": "ogm::acos_(%s)", "tanh": "ogm::tanh_(%s)", "sinh": "ogm::sinh_(%s)",
         "cosh": "ogm::cosh_(%s)", "expm1": "ogm::expm1_(%s)", "log1p": "ogm::log1p_(%s)",
         "log2": "ogm::log2_(%s)", "log10": "ogm::log10_(%s)", "cbrt": "ogm::cbrt_(%s)"}
_BIN_C = {"add": "%s + %s", "sub": "%s - %s", "mul": "%s * %s", "div": "%s / %s",
          "atan2": "ogm::atan2_(%s, %s)", "hypot": "ogm::hypot_(%s, %s)", "pow": "ogm::pow_(%s, %s)",
          "mod": "ogm::mod_(%s, %s)", "fmod": "ogm::fmod_(%s, %s)"}
_CMP_C = {"lt": "<", "le": "<=", "gt": ">", "ge": ">=", "eq": "==", "ne": "!="}


def _cdouble(raw):
    v = float(np.frombuffer(raw, dtype=np.float64)[0])
    if v != v:
        return "__builtin_nan(\"\")"
    if v in (float("inf"), float("-inf")):
        return "(-__builtin_inf())" if v < 0 else "__builtin_inf()"
    return "(%s)" % v.hex()


class _Emitter:
    def __init__(self, program):
        self.P = program
        # sequential sums (Python's sum() over the nodes of a running cost, say): every summed vector is a
        # *term block* (length, body expression).  The generated loop asks the accessor for each term
        # (OgGen::term_pick), so that a kernel can hand out base terms it computed cooperatively once instead
        # of letting every lane re-evaluate all of them (csrc/ogk_kernels.hip: XColT).
        self.term_blocks = []          # [(length, body eid)]
        self.term_index = {}
        self.sum_groups = set()        # groups whose code contains a sum
        self.eg = program.eg

    def _idx(self, base, stride, var):
        if stride == 0:
            return "%d" % base
        return "%d + %s" % (base, var) if base else var

    def _emit_expr(self, roots, var, lines, names, indent, ymap):
        """Emit SSA temporaries for every node reachable from ``roots`` (topological order)."""
        eg = self.eg
        order, seen = [], set()

        def visit(e):
            stack = [(e, False)]
            while stack:
                cur, done = stack.pop()
                if done:
                    order.append(cur)
                    continue
                if cur in seen or cur in names:
                    continue
                seen.add(cur)
                stack.append((cur, True))
                node = eg.nodes[cur]
                tag = node[0]
                if tag in ("un", "interp"):
                    stack.append((node[2], False))
                elif tag in ("bin", "cmp", "logic"):
                    stack.append((node[3], False))
                    stack.append((node[2], False))
                elif tag == "where":
                    for c in reversed(node[1:]):
                        stack.append((c, False))
        for r in roots:
            visit(r)
        pad = " " * indent
        for e in order:
            node = eg.nodes[e]
            tag = node[0]
            name = "t%d" % e
            ctype = "const S"
            if tag == "P":
                rhs = "x(%s)" % self._idx(node[1], node[2], var)
            elif tag == "C":
                rhs = _cdouble(node[1])
            elif tag == "CV":
                rhs = "cv[%s]" % self._idx(self.P.cvec_off[node[1]] + node[2], node[3], var)
            elif tag == "Y":
                rhs = "y[%d]" % ymap[node[1]]
            elif tag == "un":
                rhs = _UN_C[node[1]] % names[node[2]]
            elif tag == "interp":
                ix, iy, mode, lo, hi = self.P.tables[node[1]]
                rhs = "ogm::interp_linear(cv + %d, cv + %d, %d, %d, %s, %s, %s)" % (
                    self.P.cvec_off[ix], self.P.cvec_off[iy], self.P.table_len[node[1]], mode,
                    _cdouble(lo), _cdouble(hi), names[node[2]])
            elif tag == "bin":
                a, b = names[node[2]], names[node[3]]
                if node[1] == "max":      # np.maximum: propagate NaN from either side
                    rhs = "((%s >= %s || %s != %s) ? %s : %s)" % (a, b, a, a, a, b)
                elif node[1] == "min":
                    rhs = "((%s <= %s || %s != %s) ? %s : %s)" % (a, b, a, a, a, b)
                else:
                    rhs = _BIN_C[node[1]] % (a, b)
            elif tag == "cmp":
                ctype = "const bool"
                rhs = "%s %s %s" % (names[node[2]], _CMP_C[node[1]], names[node[3]])
            elif tag == "logic":
                ctype = "const bool"
                rhs = "%s %s %s" % (names[node[2]], "&&" if node[1] == "and" else "||",
                                    names[node[3]])
            elif tag == "where":
                rhs = "(%s ? %s : %s)" % (names[node[1]], names[node[2]], names[node[3]])
            elif tag == "sum":
                raise AssertionError("sum nodes are emitted by _emit_sums")
            else:
                raise AssertionError(tag)
            names[e] = name
            lines.append("%s%s %s = %s;" % (pad, ctype, name, rhs))

    def _collect_sums(self, roots):
        eg, out, seen = self.eg, [], set()
        stack = list(roots)
        while stack:
            e = stack.pop()
            if e in seen:
                continue
            seen.add(e)
            node = eg.nodes[e]
            tag = node[0]
            if tag == "sum":
                out.append(e)
            elif tag in ("un", "interp"):
                stack.append(node[2])
            elif tag in ("bin", "cmp", "logic"):
                stack.extend(node[2:4])
            elif tag == "where":
                stack.extend(node[1:])
        return out

    def _emit_sums(self, roots, lines, names, indent):
        pad = " " * indent
        found = self._collect_sums(roots)
        for e in found:
            node = self.eg.nodes[e]
            name = "t%d" % e
            # Python's sum(): 0 + v[0] + v[1] + ... left to right
            lines.append("%sS %s = 0.0;" % (pad, name))
            for ln, body in node[1]:
                if self._collect_sums([body]):
                    raise _tr.TraceError("nested sums are not supported")
                key = (ln, body)
                if key not in self.term_index:
                    self.term_index[key] = len(self.term_blocks)
                    self.term_blocks.append(key)
                tb = self.term_index[key]
                # the additions stay in order (Python's left-to-right sum); where the terms come from is the
                # accessor's business (default: evaluated in place, sum_term)
                # An accessor that holds the block's base terms (term_cache) supplies them with the one term that
                # reads its perturbed variable swapped in: the loop is then LDS reads and adds (_cached_sum_lines)
                lines += ["%s{" % pad,
                          "%s    const double* tcp = term_cache_of(x, %d, 0);" % (pad, tb),
                          "%s    if (tcp) {" % pad,
                          "%s        const int qd = term_q_of(x, %d, 0);" % (pad, tb),
                          "%s        const S td = term_v_of(x, %d, 0);" % (pad, tb),
                          # (the terms are loaded in groups whatever q is, a group ahead of the additions that use it:
                          # left to itself the compiler turns every select into a load under an exec mask.  The chain of a
                          # sequential sum is as long as the sum - 512 terms at C5 - and the longest thing a light
                          # workgroup of the fused launch does: 11 ns per term like this, 33 with a select on every term
                          # and the loads waited for group by group)
                          ] + self._cached_sum_lines(pad + "        ", name, ln) + [
                          "%s    } else {" % pad,
                          "%s        _Pragma(\"unroll 8\")" % pad,
                          "%s        for (int q = 0; q < %d; ++q) %s = %s + sum_term(%d, q, x, cv);" % (pad, ln, name, name, tb),
                          "%s    }" % pad,
                          "%s}" % pad]
            names[e] = name
        return bool(found)

    @staticmethod
    def _cached_sum_lines(pad, name, ln, G=4):
        """``name += term`` over the ``ln`` cached terms at ``tcp`` with term ``qd`` replaced by ``td``, in order.
        Two register buffers of G terms take turns (the loads of one are under way while the other is added; the
        cache is readable 16 doubles past its end, so no load needs a clamp or a predicate).  Only a group in which
        some lane of the wavefront has its perturbed term pays for the selects: everywhere else the chain is one
        addition per term."""
        def load(buf, at):
            return '%s_Pragma("unroll") for (int u = 0; u < %d; ++u) %s[u] = tcp[%s + u];' % (pad, G, buf, at)

        def keep(buf):
            return '%s_Pragma("unroll") for (int u = 0; u < %d; ++u) OG_KEEP(%s[u]);' % (pad, G, buf)

        def group(buf, at, cnt):
            return ['%sif (OG_ANY((unsigned)(qd - (%s)) < %du)) {' % (pad, at, cnt),
                    '%s    _Pragma("unroll") for (int u = 0; u < %d; ++u) %s = %s + (%s + u == qd ? td : S(%s[u]));'
                    % (pad, cnt, name, name, at, buf),
                    '%s} else {' % pad,
                    '%s    _Pragma("unroll") for (int u = 0; u < %d; ++u) %s = %s + S(%s[u]);' % (pad, cnt, name, name, buf),
                    '%s}' % pad]
        # (a buffer is waited for BEFORE the other one's loads go out: the wait is then for everything outstanding,
        # which the compiler gets right, and those loads have the G additions that follow to arrive)
        L = ['%sdouble ta_[%d], tb_[%d];' % (pad, G, G), load("ta_", "0"), '%sint q0 = 0;' % pad,
             '%sfor (; q0 + %d <= %d; q0 += %d) {' % (pad, 2 * G, ln, 2 * G),
             "    " + keep("ta_"), "    " + load("tb_", "q0 + %d" % G)] + ["    " + l for l in group("ta_", "q0", G)] + \
            ["    " + keep("tb_"), "    " + load("ta_", "q0 + %d" % (2 * G))] + \
            ["    " + l for l in group("tb_", "q0 + %d" % G, G)] + ['%s}' % pad]
        rest = ln % (2 * G)
        if rest >= G:
            L += [keep("ta_"), load("tb_", "q0 + %d" % G)] + group("ta_", "q0", G)
            if rest > G:
                L += [keep("tb_")] + group("tb_", "q0 + %d" % G, rest - G)
        elif rest:
            L += [keep("ta_")] + group("ta_", "q0", rest)
        return L

    def term_functions(self):
        """``sum_term(tb, q, x, cv)``: term q of block tb; ``sum_term_reads(tb, q, j)``: does it read p[j]?"""
        eg = self.eg
        offs, at = [], 0
        for ln, _ in self.term_blocks:
            offs.append(at)
            at += ln
        L = ["    static constexpr int N_TBLK = %d;" % len(self.term_blocks),
             "    static constexpr int N_TERMS = %d;" % at,
             _int_table("TERM_OFF", offs), _int_table("TERM_LEN", [ln for ln, _ in self.term_blocks]),
             "    template <class X> OG_HDI static typename X::scalar sum_term(const int tb, const int q, const X& x, "
             "const double* cv) {",
             "        typedef typename X::scalar S;",
             "        (void)q; (void)cv; (void)x;",
             "        switch (tb) {"]
        for tb, (ln, body) in enumerate(self.term_blocks):
            L.append("        case %d: {" % tb)
            inner = {}
            self._emit_expr([body], "q", L, inner, 12, {})
            L += ["            return %s;" % inner[body], "        }"]
        L += ["        default: return S(0.0);", "        }", "    }",
              # which term of block tb reads p[j]: its index, -1 none, -2 more than one (no caching for that lane)
              "    OG_HDI static int sum_term_q(const int tb, const int j) {",
              "        (void)j;",
              "        switch (tb) {"]
        for tb, (ln, body) in enumerate(self.term_blocks):
            leaves = sorted({(node[1], node[2]) for node in _leaves(eg, body, ("P",))})
            L.append("        case %d: {" % tb)
            L.append("            int q = -1;")
            for base, stride in leaves:
                if stride == 0:
                    L.append("            if (j == %d) return -2;" % base)
                elif stride == 1:
                    L.append("            if (j >= %d && j < %d) { if (q >= 0 && q != j - %d) return -2; q = j - %d; }"
                             % (base, base + ln, base, base))
                else:
                    L.append("            if (j >= %d && j <= %d && (j - %d) %% %d == 0) return -2;"
                             % (min(base, base + stride * (ln - 1)), max(base, base + stride * (ln - 1)), base, stride))
            L += ["            return q;", "        }"]
        L += ["        default: return -2;", "        }", "    }",
              # an accessor with term_cache()/term_q()/term_v() members supplies cached terms; any other: in place
              "    template <class X> OG_HDI static auto term_cache_of(const X& x, const int tb, int) -> "
              "decltype(x.term_cache(tb)) { return x.term_cache(tb); }",
              "    template <class X> OG_HDI static const double* term_cache_of(const X&, const int, long) { return nullptr; }",
              "    template <class X> OG_HDI static auto term_q_of(const X& x, const int tb, int) -> "
              "decltype(x.term_q(tb)) { return x.term_q(tb); }",
              "    template <class X> OG_HDI static int term_q_of(const X&, const int, long) { return -1; }",
              "    template <class X> OG_HDI static auto term_v_of(const X& x, const int tb, int) -> "
              "decltype(x.term_v(tb)) { return x.term_v(tb); }",
              "    template <class X> OG_HDI static typename X::scalar term_v_of(const X&, const int, long) "
              "{ return typename X::scalar(0.0); }"]
        return L

    def group_function(self, gi, grp):
        if grp.kind == "defect":
            # T_s = (tf-t0)/2 * f_s at node k; the defect is  y[s] - T_s
            lines = ["    template <class X> OG_HDI static void tail%d(const int k, const X& x, "
                     "const double* cv, typename X::scalar* T) {" % gi,
                     "        typedef typename X::scalar S;",
                     "        (void)k; (void)cv;"]
            names = {}
            if self._emit_sums(grp.tails, lines, names, 8):
                self.sum_groups.add(gi)
            self._emit_expr(grp.tails, "k", lines, names, 8, {})
            for s, e in enumerate(grp.tails):
                lines.append("        T[%d] = %s;" % (s, names[e]))
            lines.append("    }")
            for si, e in enumerate(grp.tails):       # one state's term alone: a short chain
                lines += ["    template <class X> OG_HDI static typename X::scalar tail%d_%d(const int k, "
                          "const X& x, const double* cv) {" % (gi, si),
                          "        typedef typename X::scalar S;",
                          "        (void)k; (void)cv;"]
                nm = {}
                self._emit_sums([e], lines, nm, 8)
                self._emit_expr([e], "k", lines, nm, 8, {})
                lines += ["        return %s;" % nm[e], "    }"]
            lines += ["    template <class X> OG_HDI static void group%d(const int k, const X& x, "
                      "const typename X::scalar* y, const double* cv, typename X::scalar* out) {" % gi,
                      "        typename X::scalar T[%d];" % len(grp.tails),
                      "        tail%d(k, x, cv, T);" % gi]
            for s in range(len(grp.tails)):
                lines.append("        out[%d] = y[%d] - T[%d];" % (s, s, s))
            lines.append("    }")
            return lines
        roots = [e for _, e in grp.outputs]
        if len(roots) == 1:
            # value-returning form: no address-taken temporaries in the kernels that evaluate single items
            lines = ["    template <class X> OG_HDI static typename X::scalar group%d_v(const int k, const X& x, "
                     "const double* cv) {" % gi,
                     "        typedef typename X::scalar S;",
                     "        (void)k; (void)cv;"]
            names = {}
            if self._emit_sums(roots, lines, names, 8):
                self.sum_groups.add(gi)
            self._emit_expr(roots, "k", lines, names, 8, {})
            lines += ["        return %s;" % names[roots[0]], "    }",
                      "    template <class X> OG_HDI static void group%d(const int k, const X& x, "
                      "const typename X::scalar* y, const double* cv, typename X::scalar* out) {" % gi,
                      "        (void)y;", "        out[0] = group%d_v(k, x, cv);" % gi, "    }"]
            return lines
        lines = ["    template <class X> OG_HDI static void group%d(const int k, const X& x, "
                 "const typename X::scalar* y, const double* cv, typename X::scalar* out) {" % gi,
                 "        typedef typename X::scalar S;",
                 "        (void)k; (void)y; (void)cv;"]
        names = {}
        if self._emit_sums(roots, lines, names, 8):
            self.sum_groups.add(gi)
        self._emit_expr(roots, "k", lines, names, 8, {})
        for o, (_, e) in enumerate(grp.outputs):
            lines.append("        out[%d] = %s;" % (o, names[e]))
        lines.append("    }")
        return lines

    def operand_function(self):
        lines = ["    template <class X> OG_HD static typename X::scalar mv_operand(const int slot, "
                 "const int k, const X& x, const double* cv) {",
                 "        typedef typename X::scalar S;",
                 "        (void)cv;",
                 "        switch (slot) {"]
        for si, slot in enumerate(self.P.mv):
            lines.append("        case %d: {" % si)
            names = {}
            self._emit_expr([slot.operand], "k", lines, names, 12, {})
            lines.append("            return %s;" % names[slot.operand])
            lines.append("        }")
        lines += ["        default: return 0.0;", "        }", "    }"]
        return lines


def _int_table(name, values):
    """Table as a host+device accessor (a local constexpr array is usable from device code
    without a separate device-side definition)."""
    values = list(values) or [0]
    body = ", ".join(str(int(v)) for v in values)
    return ("    OG_HD static int %s(const int i) { constexpr int t[%d] = {%s}; return t[i]; }"
            % (name, len(values), body))


def _column_items(P):
    """Per-column work lists of the structured sweep: every (group, output, element) that reads p[j].
    Entries whose J_T position is written by the MFMA tiles (row of state s, column in state s's own
    slice) are left out."""
    col_elems = [set() for _ in range(P.n)]
    for gi, g in enumerate(P.groups):
        per_output = g.out_deps if g.kind == "defect" else [g.deps] * len(g.outputs)
        for o, deps in enumerate(per_output):
            lo = hi = -1
            if g.kind == "defect":
                sl = P.mv[g.mv_slots[o]]
                lo, hi = sl.leaf_base, sl.leaf_base + sl.length
            for kind, base, cnt in deps:
                for j in range(base, base + cnt):
                    if lo <= j < hi:
                        continue
                    if kind == 1:
                        col_elems[j].add((gi, o, j - base))
                    else:
                        col_elems[j].update((gi, o, k) for k in range(g.length))
    return col_elems


def sparsity(P):
    """Static pattern of the transposed Jacobian: ``(indptr, rows)`` with ``rows[indptr[j]:indptr[j+1]]`` the
    rows of F that can depend on p[j] - for column j first the collocation block its state slice owns
    (``N`` consecutive defect rows, written by the MFMA tiles), then the row items in work-list order.
    This is the order of the packed non-zeros (``og_pack_dev``, include/ogpsx.h); every other entry of
    J_T is an exact zero in every sweep."""
    col_elems = _column_items(P)
    own = {}
    for gi, g in enumerate(P.groups):
        for o, si in enumerate(g.mv_slots):
            sl = P.mv[si]
            row0 = g.outputs[o][0]
            for j in range(sl.leaf_base, sl.leaf_base + sl.length):
                own[j] = (row0, row0 + sl.length)
    indptr, rows = [0], []
    for j in range(P.n):
        lo, hi = own.get(j, (0, 0))
        rows.extend(range(lo, hi))
        rows.extend(P.groups[gi].outputs[o][0] + k for gi, o, k in sorted(col_elems[j]))
        indptr.append(len(rows))
    return np.asarray(indptr, dtype=np.int64), np.asarray(rows, dtype=np.int32)


def emit_header(P):
    """C++17 source of ``struct OgGen`` for this program (host+device, no includes of its own
    beyond og_math.h)."""
    em = _Emitter(P)
    max_out = max(len(g.outputs) for g in P.groups)
    n_rowitems = sum(g.length for g in P.groups if g.kind == "rows")
    L = ["// generated by opengoddard_amd.codegen -- do not edit",
         "#pragma once",
         "#include \"og_math.h\"",
         "",
         "struct OgGen {",
         "    static constexpr int N_VAR = %d;" % P.n,
         "    static constexpr int M = %d;" % P.m,
         "    static constexpr int M_EQ = %d;" % P.m_eq,
         "    static constexpr int M_INEQ = %d;" % P.m_ineq,
         "    static constexpr int N_PHASE = %d;" % len(P.nodes),
         "    static constexpr int N_MV = %d;" % len(P.mv),
         "    static constexpr int MAX_NODES = %d;" % max(P.nodes),
         "    static constexpr int N_GROUPS = %d;" % len(P.groups),
         "    static constexpr int N_CVEC = %d;" % P.cvec.shape[0],
         "    static constexpr int MAX_OUT = %d;" % max_out,
         "    static constexpr int MAX_NMV = %d;" % max([len(g.mv_slots) for g in P.groups] + [1]),
         "    static constexpr int N_ROW_ITEMS = %d;" % n_rowitems,
         _int_table("PHASE_NODES", P.nodes),
         _int_table("MV_LEAF", [s.leaf_base for s in P.mv]),
         _int_table("G_KIND", [1 if g.kind == "defect" else 0 for g in P.groups]),
         _int_table("G_LEN", [g.length for g in P.groups]),
         _int_table("G_NOUT", [len(g.outputs) for g in P.groups]),
         _int_table("G_PHASE", [g.phase for g in P.groups]),
         _int_table("G_MV0", [g.mv_slots[0] if g.mv_slots else 0 for g in P.groups]),
         _int_table("G_NMV", [len(g.mv_slots) for g in P.groups])]
    rows = []
    for g in P.groups:
        r = [row for row, _ in g.outputs]
        rows += r + [0] * (max_out - len(r))
    L.append(_int_table("G_ROW_FLAT", rows))
    L.append("    OG_HD static int G_ROW(const int g, const int o) { return G_ROW_FLAT(g * MAX_OUT + o); }")
    # prefix of row-group item counts: row item ri -> (group, k)
    starts, at = [], 0
    for g in P.groups:
        starts.append(at if g.kind == "rows" else -1)
        if g.kind == "rows":
            at += g.length
    L.append(_int_table("G_ITEM0", starts))
    # dependency table (group-major) and per-group ranges into it
    dep_g, dep_kind, dep_base, dep_cnt, g_dep0, g_ndep = [], [], [], [], [], []
    for gi, g in enumerate(P.groups):
        g_dep0.append(len(dep_g))
        g_ndep.append(len(g.deps))
        for kind, base, cnt in g.deps:
            dep_g.append(gi), dep_kind.append(kind), dep_base.append(base), dep_cnt.append(cnt)
    L += [_int_table("DEP_KIND", dep_kind), _int_table("DEP_BASE", dep_base),
          _int_table("DEP_CNT", dep_cnt)]
    # collocation slots: owning group, offset of their base product in the y0 scratch
    slot_group = [0] * len(P.mv)
    for gi, g in enumerate(P.groups):
        for sl in g.mv_slots:
            slot_group[sl] = gi
    y0_off, at = [], 0
    for sl in P.mv:
        y0_off.append(at)
        at += sl.length
    col_elems = _column_items(P)
    col_ptr, elem_g, elem_o, elem_k = [0], [], [], []
    for j in range(P.n):
        for gi, o, k in sorted(col_elems[j]):
            elem_g.append(gi), elem_o.append(o), elem_k.append(k)
        col_ptr.append(len(elem_g))
    counts = np.diff(col_ptr)
    # the collocation tile (defect group, 16-node tile) a column's defect items live in: the fused launch
    # gives a light workgroup the base products of exactly one such tile.  Columns whose defect items
    # spread over more than one tile get a workgroup of their own (like the columns with many items).
    col_tile = []
    for j in range(P.n):
        keys = {(gi, k >> 4) for gi, o, k in col_elems[j] if P.groups[gi].kind == "defect"}
        col_tile.append(None if not keys else (next(iter(keys)) if len(keys) == 1 else "many"))
    heavy = [int(j) for j in range(P.n) if counts[j] > HEAVY_COLUMN_ELEMENTS or col_tile[j] == "many"]
    heavy.sort(key=lambda j: -counts[j])
    # light columns in runs of at most LIGHT_COLS neighbours that share a tile
    light_groups, run, heavy_set = [], None, set(heavy)
    for j in range(P.n):
        if j in heavy_set:
            run = None
            continue
        key = col_tile[j]
        if run is not None and run[0] + run[1] == j and run[1] < fused_cols(P.n) and \
                (key is None or run[2] is None or run[2] == key):
            run[1] += 1
            run[2] = run[2] if run[2] is not None else key
        else:
            run = [j, 1, key]
            light_groups.append(run)
    # rows of a J_T row written by the MFMA tiles (j inside a collocated state slice)
    own_lo, own_hi = [0] * P.n, [0] * P.n
    mv_diag, mv_generic = [], []
    for si, sl in enumerate(P.mv):
        g = P.groups[slot_group[si]]
        row0 = g.outputs[si - g.mv_slots[0]][0]
        for j in range(sl.leaf_base, sl.leaf_base + sl.length):
            own_lo[j], own_hi[j] = row0, row0 + sl.length
        # how does the dynamics term of this group depend on columns of the slot's own slice?
        diag = any(kind == 1 and base == sl.leaf_base for kind, base, cnt in g.deps)
        other = any(not (kind == 1 and base == sl.leaf_base) and
                    base < sl.leaf_base + sl.length and base + cnt > sl.leaf_base
                    for kind, base, cnt in g.deps)
        mv_diag.append(int(diag))
        mv_generic.append(int(other))
    L.append("    static constexpr int N_HEAVY = %d;" % len(heavy))
    L += [_int_table("MV_Y0", y0_off),
          "    static constexpr int N_Y0 = %d;" % max(at, 1)]
    L.append("")
    for gi, g in enumerate(P.groups):
        L += em.group_function(gi, g)
        L.append("")
    L += em.operand_function()
    L.append("")
    L += em.term_functions()
    L.append("")
    L.append("    template <class X> OG_HDI static void defect_tail(const int g, const int k, "
             "const X& x, const double* cv, typename X::scalar* T) {")
    L.append("        switch (g) {")
    for gi, g in enumerate(P.groups):
        if g.kind == "defect":
            L.append("        case %d: tail%d(k, x, cv, T); break;" % (gi, gi))
    L += ["        default: break;", "        }", "    }", ""]
    # one J_T entry's worth of work: value of output o of group g at element k, and its row
    L.append("    template <class X> OG_HDI static typename X::scalar item_value(const int g, const int o, "
             "const int k, const X& x, const double* y0, const double* cv, int* row) {")
    L.append("        typedef typename X::scalar S;")
    L.append("        (void)o; (void)y0;")
    L.append("        switch (g) {")
    for gi, g in enumerate(P.groups):
        L.append("        case %d: {" % gi)
        if g.kind == "defect":
            L.append("            switch (o) {")
            for si in range(len(g.tails)):
                L.append("            case %d: *row = %d + k; return S(x.ldy(y0 + %d + k)) - tail%d_%d(k, x, cv);"
                         % (si, g.outputs[si][0], y0_off[g.mv_slots[si]], gi, si))
            L += ["            default: break;", "            }", "            break;"]
        else:
            if len(g.outputs) != 1:
                raise _tr.TraceError("row groups must have one output (OG_MAX_GROUP_OUTPUTS=1)")
            L += ["            *row = %d + k; return group%d_v(k, x, cv);" % (g.outputs[0][0], gi)]
        L.append("        }")
    L += ["        default: break;", "        }", "        *row = 0;", "        return S(0.0);", "    }", ""]
    # the same split in two: everything of an item except the base collocation product it subtracts from
    # (row items: the whole value, *yoff = -1) - the long chain that does not depend on the product - and the
    # offset of that product in the y0 scratch.  value = yoff >= 0 ? y0[yoff] - tail : tail.
    L.append("    template <class X> OG_HDI static typename X::scalar item_tail(const int g, const int o, "
             "const int k, const X& x, const double* cv, int* row, int* yoff) {")
    L.append("        typedef typename X::scalar S;")
    L.append("        (void)o;")
    L.append("        switch (g) {")
    for gi, g in enumerate(P.groups):
        L.append("        case %d: {" % gi)
        if g.kind == "defect":
            L.append("            switch (o) {")
            for si in range(len(g.tails)):
                L.append("            case %d: *row = %d + k; *yoff = %d + k; return tail%d_%d(k, x, cv);"
                         % (si, g.outputs[si][0], y0_off[g.mv_slots[si]], gi, si))
            L += ["            default: break;", "            }", "            break;"]
        else:
            L += ["            *row = %d + k; *yoff = -1; return group%d_v(k, x, cv);" % (g.outputs[0][0], gi)]
        L.append("        }")
    L += ["        default: break;", "        }", "        *row = 0;", "        *yoff = -1;", "        return S(0.0);",
          "    }", ""]
    # the dynamics term of one collocation slot (one state) at node k
    L.append("    template <class X> OG_HDI static typename X::scalar tail_one(const int slot, const int k, "
             "const X& x, const double* cv) {")
    L.append("        switch (slot) {")
    for gi, g in enumerate(P.groups):
        for si, sl in enumerate(g.mv_slots):
            L.append("        case %d: return tail%d_%d(k, x, cv);" % (sl, gi, si))
    L += ["        default: return typename X::scalar(0.0);", "        }", "    }", ""]
    L.append("    template <class X> OG_HDI static void group_eval(const int g, const int k, "
             "const X& x, const typename X::scalar* y, const double* cv, typename X::scalar* out) {")
    L.append("        switch (g) {")
    for gi in range(len(P.groups)):
        L.append("        case %d: group%d(k, x, y, cv, out); break;" % (gi, gi))
    L += ["        default: break;", "        }", "    }", "};", ""]
    L += _sweep_records(P, col_ptr, elem_g, elem_o, elem_k, own_lo, own_hi, heavy, slot_group,
                        y0_off, mv_diag, mv_generic, g_dep0, g_ndep, light_groups, em.sum_groups)
    return "\n".join(L)


SWEEP_WAVES = 8          # wavefronts per ogk_sweep workgroup (csrc/ogk_kernels.hip)


def _sweep_records(P, col_ptr, elem_g, elem_o, elem_k, own_lo, own_hi, heavy, slot_group, y0_off,
                   mv_diag, mv_generic, g_dep0, g_ndep, light_groups, sum_groups=()):
    """Wide, aligned device tables so that a workgroup of the structured sweep learns everything
    about its column / item / MFMA tile from ONE load each (every dependent global load costs
    a few hundred cycles, and the sweep of a small problem is a chain of them)."""
    def table(ctype, name, rows):
        rows = rows or [[0] * (4 if ctype == "int4" else 8)]
        body = ",\n".join("    {%s}" % ", ".join(str(int(v)) for v in r) for r in rows)
        return ["static __device__ const %s %s[%d] = {" % (ctype, name, len(rows)), body, "};"]
    heavy_set = set(heavy)
    col = [[col_ptr[j], col_ptr[j + 1], own_lo[j], own_hi[j] | ((1 << 30) if j in heavy_set else 0)]
           for j in range(P.n)]
    elem = [[g, o, k, P.groups[g].outputs[max(o, 0)][0] + k] for g, o, k in zip(elem_g, elem_o, elem_k)]
    tiles, slots = [], []
    for si, sl in enumerate(P.mv):
        t16 = (sl.length + 15) // 16
        for mtg in range((t16 + SWEEP_WAVES - 1) // SWEEP_WAVES):
            for nt in range(t16):
                tiles.append([si, mtg, nt, 0])
        gi = slot_group[si]
        g = P.groups[gi]
        row0 = g.outputs[si - g.mv_slots[0]][0]
        slots.append([sl.length, gi, sl.leaf_base, row0, y0_off[si], sl.phase,
                      mv_diag[si] | (mv_generic[si] << 1), (g_dep0[gi] << 12) | g_ndep[gi]])
    rowwaves = []
    for gi, g in enumerate(P.groups):
        if g.kind == "rows":
            for k0 in range(0, g.length, 64):
                rowwaves.append([gi, k0, g.length, 1 if gi in sum_groups else 0])   # .w: its code contains a sum
    evalblk = []
    for gi, g in enumerate(P.groups):
        if g.kind == "defect":
            for nt in range((g.length + 15) // 16):
                evalblk.append([gi, nt, g.mv_slots[0], len(g.mv_slots)])
    L = ["#if defined(__HIPCC__)"]
    L += table("int4", "OGT_EVALBLK", evalblk)
    L.append("static constexpr int OGT_N_EVALBLK = %d;" % len(evalblk))
    # fused launch: {first column, columns, defect group or -1, node tile} per light workgroup; the first
    # columns again as a host table (ogk_launch picks the groups a column range touches)
    # (everything a workgroup needs about its tile in ONE record: index tables looked up with a runtime
    # group number end up as stack copies in the kernel)
    lgrp, lrng = [], []
    for j0, cnt, key in light_groups:
        for c in range(fused_cols(P.n)):
            # {items begin, end, entries of the column's own collocation block (they precede its items in the
            #  packed order)}
            lrng.append([col_ptr[j0 + c], col_ptr[j0 + c + 1], own_hi[j0 + c] - own_lo[j0 + c], 0]
                        if c < cnt else [0, 0, 0, 0])
        # bit 16 of the last field: some item of the workgroup contains a sum (base terms are then cached in LDS)
        terms = int(any(elem_g[e] in sum_groups for e in range(col_ptr[j0], col_ptr[j0 + cnt]))) << 16
        if key:
            g = P.groups[key[0]]
            lgrp.append([j0, cnt, y0_off[g.mv_slots[0]], key[1], g.mv_slots[0], len(g.mv_slots), g.length,
                         g.phase | terms])
        else:
            lgrp.append([j0, cnt, 0, 0, 0, 0, 0, terms])
    # heavy columns of the fused launch are cut into parts that look like light workgroups: one part per
    # (defect group, 16-node tile) the column has items in - the tile's D^T panel and the group's operands go
    # through LDS, one wavefront runs the tile's base products - plus one tile-less part for its row items.
    # A part's items come in *slots*: runs of at most 16 (defect: one output over the tile's nodes) or 32 (rows:
    # one row group) items that share their code, one wavefront each (lanes: items at x0 + h e_j | the same at x0).
    # OGT_HELEM holds the heavy columns' items in that order (OGT_ELEM keeps the order mode 1 uses).
    hpart, hslot, helem = [], [], []
    for j in heavy:
        entries = [(elem_g[e], elem_o[e], elem_k[e]) for e in range(col_ptr[j], col_ptr[j + 1])]
        # position of an item in the column's packed order (codegen.sparsity): own block first, then OGT_ELEM order
        ppos = {ent: (own_hi[j] - own_lo[j]) + i for i, ent in enumerate(entries)}
        tiles_of = {}
        for gi, o, k in entries:
            if P.groups[gi].kind == "defect":
                tiles_of.setdefault((gi, k >> 4), {}).setdefault(o, []).append(k)
        for (gi, nt_), by_out in sorted(tiles_of.items()):
            g = P.groups[gi]
            first_slot = len(hslot)
            for o, ks in sorted(by_out.items()):
                hslot.append([len(helem), len(ks), 0, 0])
                helem += [[gi, o, k, ppos[(gi, o, k)]] for k in sorted(ks)]
            hpart.append([j, first_slot, len(hslot), y0_off[g.mv_slots[0]], nt_, g.mv_slots[0], len(g.mv_slots),
                          g.length | (g.phase << 20) | ((1 << 30) if gi in sum_groups else 0)])
        rows_by_group = {}
        for gi, o, k in entries:
            if P.groups[gi].kind != "defect":
                rows_by_group.setdefault((gi, o), []).append(k)
        if rows_by_group:
            first_slot = len(hslot)
            for (gi, o), ks in sorted(rows_by_group.items()):
                ks = sorted(ks)
                for c0 in range(0, len(ks), 32):
                    chunk = ks[c0:c0 + 32]
                    hslot.append([len(helem), len(chunk), 0, 0])
                    helem += [[gi, o, k, ppos[(gi, o, k)]] for k in chunk]
            hpart.append([j, first_slot, len(hslot), 0, 0, 0, 0,
                          (1 << 30) if any(gi in sum_groups for gi, _ in rows_by_group) else 0])
    L += ["struct ogt_int8 { int v[8]; };",
          "static __device__ const ogt_int8 OGT_HPART[%d] = {" % max(len(hpart), 1),
          ",\n".join("    {{%s}}" % ", ".join(str(int(v)) for v in r) for r in (hpart or [[0] * 8])),
          "};",
          "static constexpr int OGT_N_HPART = %d;" % len(hpart),
          "static constexpr int OGT_LGRP_COLS = %d;" % fused_cols(P.n),
          "static __device__ const ogt_int8 OGT_LGRP[%d] = {" % max(len(lgrp), 1),
          ",\n".join("    {{%s}}" % ", ".join(str(int(v)) for v in r) for r in (lgrp or [[0] * 8])),
          "};"]
    # OGT_HELEM: {group, output, element, position in the column's packed order}
    L += table("int4", "OGT_HSLOT", hslot)    # {first item in OGT_HELEM, items} per slot of a heavy part
    L += table("int4", "OGT_HELEM", helem)
    L += table("int4", "OGT_LRNG", lrng)      # {items begin, end} per (group, column)
    L += ["static constexpr int OGT_N_LGRP = %d;" % len(light_groups),
          "static const int OGH_LGRP_J[%d] = {%s};" % (len(light_groups) + 1, ", ".join(
              [str(r[0]) for r in light_groups] + [str(P.n)]))]
    # Light workgroups whose items contain a sequential sum (a running cost: a chain of as many dependent additions
    # as the sum has terms) are the longest of the launch: they are dispatched FIRST among the light workgroups - the
    # launch's block index is mapped through these two lists (the groups with a sum, the others, each in column order;
    # the host passes how many of the first lie below the launch's column range, ogk_launch)
    with_sum = [i for i, r in enumerate(lgrp) if r[7] & (1 << 16)]
    without = [i for i, r in enumerate(lgrp) if not r[7] & (1 << 16)]
    L += ["static __device__ const int OGT_LSUM[%d] = {%s};" % (max(len(with_sum), 1), ", ".join(map(str, with_sum or [0]))),
          "static __device__ const int OGT_LPLAIN[%d] = {%s};" % (max(len(without), 1), ", ".join(map(str, without or [0]))),
          "static const unsigned char OGH_LGRP_SUM[%d] = {%s};" % (max(len(lgrp), 1), ", ".join(
              "1" if r[7] & (1 << 16) else "0" for r in (lgrp or [[0] * 8])))]
    L += table("int4", "OGT_ROWWAVE", rowwaves)
    L.append("static constexpr int OGT_N_ROWWAVES = %d;" % len(rowwaves))
    L += table("int4", "OGT_COL", col)
    L += table("int4", "OGT_ELEM", elem)
    L += table("int4", "OGT_TILE", tiles)
    # fused launch: {slot, first column tile, node tile, column tiles} - at most SWEEP_WAVES - 1 column tiles
    # per workgroup (the last wavefront computes the base products of the node tile), spread evenly
    ftiles = []
    for si, sl in enumerate(P.mv):
        t16 = (sl.length + 15) // 16
        ngrp = -(-t16 // tile_cols())
        per = -(-t16 // ngrp)
        for c0 in range(0, t16, per):
            for nt in range(t16):
                ftiles.append([si, c0, nt, min(per, t16 - c0)])
    L += table("int4", "OGT_FTILE", ftiles)
    L.append("static constexpr int OGT_N_FTILES = %d;" % len(ftiles))
    L += table("ogt_int8", "OGT_SLOT", [[r] for r in []] or None) if False else \
        ["static __device__ const ogt_int8 OGT_SLOT[%d] = {" % max(len(slots), 1),
         ",\n".join("    {{%s}}" % ", ".join(str(int(v)) for v in r) for r in (slots or [[0] * 8])),
         "};"]
    L += ["static constexpr int OGT_N_TILES = %d;" % len(tiles),
          "static __device__ const int OGT_HEAVY[%d] = {%s};" % (
              max(len(heavy), 1), ", ".join(str(j) for j in heavy) or "0"),
          "#endif", ""]
    return L


def program_hash(source):
    return hashlib.sha256(source.encode()).hexdigest()[:16]
