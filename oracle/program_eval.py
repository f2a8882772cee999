"""TEST INFRASTRUCTURE - NumPy interpreter for a lowered ``codegen.Program``.

Evaluates the traced-and-lowered row pieces with the same NumPy ufuncs, in the same order, as
the user's callbacks would run, so ``evaluate(program, prob, x)`` must equal
``np_path.stacked_values(prob, obj, x)`` *bit for bit*.  That isolates tracer / lowering bugs
(wrong slice, wrong row order, lost quirk) from arithmetic differences, before any C++ or HIP
is involved.  Only tests import this.
"""
from __future__ import annotations

import numpy as np

_UN = {"neg": np.negative, "sqrt": np.sqrt, "exp": np.exp, "log": np.log, "sin": np.sin,
       "cos": np.cos, "tan": np.tan, "abs": np.abs, "atan": np.arctan, "asin": np.arcsin,
       "acos": np.arccos, "tanh": np.tanh, "sinh": np.sinh, "cosh": np.cosh, "expm1": np.expm1,
       "log1p": np.log1p, "log2": np.log2, "log10": np.log10, "cbrt": np.cbrt}
_BIN = {"add": np.add, "sub": np.subtract, "mul": np.multiply, "div": np.true_divide,
        "max": np.maximum, "min": np.minimum, "atan2": np.arctan2, "hypot": np.hypot, "pow": np.power,
        "mod": np.remainder, "fmod": np.fmod}
_CMP = {"lt": np.less, "le": np.less_equal, "gt": np.greater, "ge": np.greater_equal,
        "eq": np.equal, "ne": np.not_equal}


class _Eval:
    def __init__(self, P, x, D):
        self.P, self.x, self.D = P, np.asarray(x, dtype=float), D
        self.yvec = {}

    def mv(self, slot):
        if slot not in self.yvec:
            s = self.P.mv[slot]
            operand = self.elem(s.operand, s.length, {})
            self.yvec[slot] = self.D[s.phase].dot(np.broadcast_to(operand, (s.length,)))
        return self.yvec[slot]

    def elem(self, eid, length, memo):
        hit = memo.get(eid)
        if hit is not None:
            return hit
        node = self.P.eg.nodes[eid]
        tag = node[0]
        k = np.arange(length)
        if tag == "P":
            out = self.x[node[1] + node[2] * k] if node[2] else np.float64(self.x[node[1]])
        elif tag == "C":
            out = np.frombuffer(node[1], dtype=np.float64)[0]
        elif tag == "CV":
            base = self.P.cvec_off[node[1]] + node[2]
            out = self.P.cvec[base + node[3] * k] if node[3] else np.float64(self.P.cvec[base])
        elif tag == "Y":
            y = self.mv(node[1])
            out = y[node[2] + node[3] * k] if node[3] else np.float64(y[node[2]])
        elif tag == "un":
            out = _UN[node[1]](self.elem(node[2], length, memo))
        elif tag == "interp":
            ix, iy, mode, lo, hi = self.P.tables[node[1]]
            n = self.P.table_len[node[1]]
            xg = self.P.cvec[self.P.cvec_off[ix]:self.P.cvec_off[ix] + n]
            yg = self.P.cvec[self.P.cvec_off[iy]:self.P.cvec_off[iy] + n]
            xn = np.atleast_1d(np.asarray(self.elem(node[2], length, memo), dtype=float))
            idx = np.searchsorted(xg, xn).clip(1, n - 1).astype(int)        # SciPy _call_linear
            slope = (yg[idx] - yg[idx - 1]) / (xg[idx] - xg[idx - 1])
            out = slope * (xn - xg[idx - 1]) + yg[idx - 1]
            if mode != 1:
                flo = np.frombuffer(lo, dtype=np.float64)[0] if mode == 0 else np.nan
                fhi = np.frombuffer(hi, dtype=np.float64)[0] if mode == 0 else np.nan
                out[xn < xg[0]] = flo
                out[xn > xg[-1]] = fhi
            if np.ndim(self.elem(node[2], length, memo)) == 0:
                out = np.float64(out[0])
        elif tag == "bin":
            out = _BIN[node[1]](self.elem(node[2], length, memo), self.elem(node[3], length, memo))
        elif tag == "cmp":
            out = _CMP[node[1]](self.elem(node[2], length, memo), self.elem(node[3], length, memo))
        elif tag == "logic":
            f = np.logical_and if node[1] == "and" else np.logical_or
            out = f(self.elem(node[2], length, memo), self.elem(node[3], length, memo))
        elif tag == "where":
            c, a, b = (self.elem(e, length, memo) for e in node[1:])
            out = np.where(c, a, b)
        elif tag == "sum":
            total = 0
            for ln, body in node[1]:
                vec = np.broadcast_to(self.elem(body, ln, {}), (ln,))
                for v in vec:
                    total = total + v
            out = np.float64(total)
        else:
            raise AssertionError(tag)
        memo[eid] = out
        return out


def evaluate(program, prob, x):
    """F(x) = [cost | c_eq | c_ineq] from the lowered pieces."""
    ev = _Eval(program, x, prob.D)
    F = np.full(program.m, np.nan)
    for row, ln, eid, _kind in program.pieces:
        F[row:row + ln] = ev.elem(eid, ln, {})
    return F
