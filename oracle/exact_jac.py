"""TEST INFRASTRUCTURE - independent check of the exact-Jacobian mode: complex-step derivatives.

``jacobian(program, prob, x)`` evaluates the lowered program (the same pieces ``program_eval``
interprets) on ``x + i*eps*e_j`` with NumPy's complex arithmetic: for every analytic operation the
imaginary part divided by ``eps`` is the derivative along x_j to machine precision (no subtraction,
eps = 1e-30).  It shares nothing with ``csrc/og_dual.h``: the derivative rules of
+ - * / sqrt exp log sin cos tan arctan arcsin arccos are NumPy's own complex functions.  The
non-analytic operations take their branch from the real part, as a one-sided difference would:
``abs``, ``maximum``/``minimum``, comparisons, ``where``, table lookup, ``arctan2``.

Only tests import this module.
"""
from __future__ import annotations

import numpy as np

from . import program_eval

EPS = 1e-30


def _re(v):
    return np.real(v)


def _abs(z):
    return np.where(_re(z) > 0, z, np.where(_re(z) < 0, -z, _re(z) * 0.0 + 0j))


def _max(a, b):                      # np.maximum with NaN propagation, branch on the real parts
    ra, rb = _re(a), _re(b)
    return np.where((ra >= rb) | (ra != ra), a + 0j, b + 0j)


def _min(a, b):
    ra, rb = _re(a), _re(b)
    return np.where((ra <= rb) | (ra != ra), a + 0j, b + 0j)


def _atan2(y, x):
    ry, rx = _re(y), _re(x)
    return np.arctan2(ry, rx) + 1j * (rx * np.imag(y) - ry * np.imag(x)) / (rx * rx + ry * ry)


def _cbrt(z):                      # np.cbrt has no complex loop: real cube root of the real part, analytic derivative
    c = np.cbrt(_re(z))
    with np.errstate(all="ignore"):
        return c + 1j * np.where(c != 0, np.imag(z) / (3.0 * c * c), 0.0)


def _hypot(a, b):
    h = np.hypot(_re(a), _re(b))
    with np.errstate(all="ignore"):
        return h + 1j * np.where(h != 0, (_re(a) * np.imag(a) + _re(b) * np.imag(b)) / h, 0.0)


def _mod(a, b, which):
    """remainder / fmod: piecewise a - q b with q = floor (trunc) of a / b constant between the jumps"""
    ra, rb = _re(a), _re(b)
    with np.errstate(all="ignore"):
        q = np.floor(ra / rb) if which == "mod" else np.trunc(ra / rb)
        val = np.remainder(ra, rb) if which == "mod" else np.fmod(ra, rb)
        return val + 1j * (np.imag(a) - np.where(np.isfinite(q), q, 0.0) * np.imag(b))


class _ComplexEval(program_eval._Eval):
    def __init__(self, P, x, D):
        super().__init__(P, np.real(x), D)
        self.x = np.asarray(x, dtype=complex)

    def elem(self, eid, length, memo):
        hit = memo.get(eid)
        if hit is not None:
            return hit
        node = self.P.eg.nodes[eid]
        tag = node[0]
        k = np.arange(length)
        if tag == "P":                      # (the real interpreter casts scalars to float64)
            out = self.x[node[1] + node[2] * k] if node[2] else self.x[node[1]]
        elif tag == "Y":
            y = self.mv(node[1])
            out = y[node[2] + node[3] * k] if node[3] else y[node[2]]
        elif tag == "un" and node[1] == "abs":
            out = _abs(self.elem(node[2], length, memo))
        elif tag == "un" and node[1] == "cbrt":
            out = _cbrt(self.elem(node[2], length, memo))
        elif tag == "bin" and node[1] == "hypot":
            out = _hypot(self.elem(node[2], length, memo), self.elem(node[3], length, memo))
        elif tag == "bin" and node[1] in ("mod", "fmod"):
            out = _mod(self.elem(node[2], length, memo), self.elem(node[3], length, memo), node[1])
        elif tag == "bin" and node[1] in ("max", "min", "atan2"):
            a, b = self.elem(node[2], length, memo), self.elem(node[3], length, memo)
            out = {"max": _max, "min": _min, "atan2": _atan2}[node[1]](a, b)
        elif tag == "cmp":
            f = program_eval._CMP[node[1]]
            out = f(_re(self.elem(node[2], length, memo)), _re(self.elem(node[3], length, memo)))
        elif tag == "interp":
            ix, iy, mode, lo, hi = self.P.tables[node[1]]
            n = self.P.table_len[node[1]]
            xg = self.P.cvec[self.P.cvec_off[ix]:self.P.cvec_off[ix] + n]
            yg = self.P.cvec[self.P.cvec_off[iy]:self.P.cvec_off[iy] + n]
            z = self.elem(node[2], length, memo)
            xn = np.atleast_1d(np.asarray(z, dtype=complex))
            idx = np.searchsorted(xg, _re(xn), side="right").clip(1, n - 1).astype(int)   # right-hand segment
            slope = (yg[idx] - yg[idx - 1]) / (xg[idx] - xg[idx - 1])
            out = slope * (xn - xg[idx - 1]) + yg[idx - 1]
            if mode != 1:
                flo = np.frombuffer(lo, dtype=np.float64)[0] if mode == 0 else np.nan
                fhi = np.frombuffer(hi, dtype=np.float64)[0] if mode == 0 else np.nan
                out[_re(xn) < xg[0]] = flo
                out[_re(xn) > xg[-1]] = fhi
            if np.ndim(z) == 0:
                out = out[0]
        elif tag == "sum":
            total = 0j
            for ln, body in node[1]:
                vec = np.broadcast_to(self.elem(body, ln, {}), (ln,))
                for v in vec:
                    total = total + v
            out = total
        else:
            return super().elem(eid, length, memo)
        memo[eid] = out
        return out


def jacobian(program, prob, x, columns=None):
    """``JT[r, :] = dF/dx_columns[r]`` (F = [cost | c_eq | c_ineq]) by complex-step differentiation."""
    x = np.asarray(x, dtype=float)
    columns = range(x.size) if columns is None else columns
    JT = np.empty((len(columns), program.m))
    for r, j in enumerate(columns):
        z = x.astype(complex)
        z[j] += 1j * EPS
        ev = _ComplexEval(program, z, prob.D)
        F = np.full(program.m, np.nan, dtype=complex)
        for row, ln, eid, _kind in program.pieces:
            F[row:row + ln] = ev.elem(eid, ln, {})
        JT[r] = np.imag(F) / EPS
    return JT
