"""Callback tracer: turns the user's NumPy-style OpenGoddard callbacks into an expression graph.

The reference evaluates ``dynamics / equality / inequality / cost / running_cost`` as arbitrary
Python on every one of the 3n+2 callback evaluations of an SLSQP major iteration
(reference ``OpenGoddard/optimize.py:670-715``).  A GPU sweep needs the same
arithmetic as device code.  During ``Problem.solve`` the decision vector ``prob.p`` is replaced
*once* by a :class:`Sym` vector; the unmodified callbacks then run on symbolic slices, and
every NumPy ufunc / operator they apply is recorded in exactly the order NumPy would execute
it.  :mod:`opengoddard_amd.codegen` lowers the recorded graph to HIP device functions.

Supported (this is what the shipped examples use, SURVEY.md section 0 finding 1): ``+ - * /``,
unary minus, ``**`` with exponent 2 / 0.5 / 1 / -1 (NumPy's fast paths), ``np.sqrt exp log sin
cos tan arctan arcsin arccos arctan2 abs square deg2rad rad2deg maximum minimum where``, linear
``scipy.interpolate.interp1d`` objects, comparisons, integer / negative
indexing, unit-step slicing, boolean-mask assignment with a scalar (``h[h < a] = a``,
reference ``examples/09_Rocket_Ascent_Polar_TSTO.py:36``), ``np.hstack / concatenate / append``.
Anything else raises :class:`TraceError` - there is no silent CPU fallback.
"""
from __future__ import annotations

import numbers
import os
import threading

import numpy as np


# what to write instead, by what the message is about (there is no CPU fallback: the error has to say how to go on)
_HINTS = (
    ("control flow", "use np.where(condition, a, b) (both branches are evaluated), np.maximum / np.minimum / np.clip, or a "
                     "mask assignment `v[v < lo] = lo`; a condition on a CONSTANT (problem data, a node count) is fine"),
    ("NumPy routine", "traceable are the elementwise ufuncs, np.where / clip / hstack / concatenate / append / stack / vstack, "
                      "np.sum / mean / prod / min / max / dot / matmul / cumsum / diff / roll / flip / take / interp / trapz, "
                      "slicing and index arrays; express the routine through those, or precompute what depends on problem "
                      "data only outside the callback"),
    ("NumPy function", "traceable are the elementwise ufuncs, np.where / clip / hstack / concatenate / append / stack / vstack, "
                       "np.sum / mean / prod / min / max / dot / matmul / cumsum / diff / roll / flip / take / interp / trapz, "
                       "slicing and index arrays; express the routine through those, or precompute what depends on problem "
                       "data only outside the callback"),
    ("ufunc", "see the list of traced functions in DESIGN.md section 1 (exp, log, sin ... arctan2, hypot, cbrt, x ** y); a "
              "special function can be tabulated and read through np.interp or a linear scipy.interpolate.interp1d"),
    ("interp", "linear interpolation (np.interp, scipy.interpolate.interp1d(kind='linear')) is traced; tabulate a smoother "
               "interpolant on a finer grid"),
    ("keyword", "call the reduction with axis= only (no out=, keepdims=, where=, dtype=)"),
    ("float(", "keep the value a traced scalar - arithmetic, comparisons inside np.where and indexing results work on it"),
)


def _user_frame():
    """(file, line, source text) of the innermost frame that is the user's code: not this package, not NumPy / SciPy."""
    import linecache
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    frame = sys._getframe(2)
    while frame is not None:
        fname = frame.f_code.co_filename
        low = fname.replace("\\", "/")
        if not (os.path.abspath(fname).startswith(here) or "/numpy/" in low or "/scipy/" in low or low.startswith("<")):
            return fname, frame.f_lineno, (linecache.getline(fname, frame.f_lineno) or "").strip()
        frame = frame.f_back
    return None


class TraceError(RuntimeError):
    """A callback did something the tracer cannot turn into device code.  The message names the line of the user's
    callback that did it and what to write instead (VERDICT r5 #8d: with no CPU fallback the error is the documentation)."""

    def __init__(self, message=""):
        message = str(message)
        where = None
        try:
            where = _user_frame()
        except Exception:
            where = None
        self.user_frame = where
        if where and "\n  at " not in message:
            message += "\n  at %s:%d:  %s" % where
            for key, hint in _HINTS:
                if key in message:
                    message += "\n  instead: " + hint
                    break
        super().__init__(message)


class TraceAttributeError(TraceError, AttributeError):
    """An ndarray attribute the tracer does not model.  Also an ``AttributeError``, so that duck-typing probes
    (``hasattr(v, "dtype")``, ``getattr(v, "shape", None)``) answer False / the default instead of raising."""


# ----------------------------------------------------------------------------- graph nodes
class Graph:
    """Hash-consed expression DAG.  A node is a tuple; ``length`` None means scalar."""

    def __init__(self):
        self.nodes = []          # id -> tuple
        self.length = []         # id -> int | None
        self.isbool = []         # id -> bool
        self._index = {}
        self.cvecs = []          # constant vectors referenced by ("cvec", i)
        self._cvec_index = {}
        self.tables = []         # linear lookup tables: (cvec id of x grid, cvec id of y, mode, lo, hi)

    def add(self, node, length, isbool=False):
        key = (node, length, isbool)
        hit = self._index.get(key)
        if hit is not None:
            return hit
        self.nodes.append(node)
        self.length.append(length)
        self.isbool.append(isbool)
        self._index[key] = len(self.nodes) - 1
        return len(self.nodes) - 1

    def const(self, value):
        value = float(value)
        # key on the bit pattern so that -0.0 and 0.0, and NaNs, stay distinct / hashable
        return self.add(("const", np.float64(value).tobytes()), None)

    def cvec(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        key = arr.tobytes()
        idx = self._cvec_index.get(key)
        if idx is None:
            idx = len(self.cvecs)
            self.cvecs.append(arr.copy())
            self._cvec_index[key] = idx
        return self.add(("cvec", idx), int(arr.shape[0]))


def _graph_parts(g, nid, start, count):
    """Node ids (vectors, or scalars counting as one element) whose concatenation is ``nid[start:start+count]``.
    Looks through ``cat`` and ``slice`` nodes and cuts constant vectors directly, so that repeated item assignment
    into one buffer keeps a FLAT graph (depth O(1), not one nesting level per assignment)."""
    if count <= 0:
        return []
    node, total = g.nodes[nid], g.length[nid]
    if total is None:
        return [nid]
    if node[0] == "cat":
        out, at = [], 0
        for child in node[1]:
            ln = g.length[child]
            ln = 1 if ln is None else ln
            lo, hi = max(start, at), min(start + count, at + ln)
            if lo < hi:
                out.extend(_graph_parts(g, child, lo - at, hi - lo))
            at += ln
        return out
    if node[0] == "slice":
        return _graph_parts(g, node[1], node[2] + start, count)
    if start == 0 and count == total:
        return [nid]
    if node[0] == "cvec":
        return [g.cvec(g.cvecs[node[1]][start:start + count])]
    return [g.add(("slice", nid, start, count), count)]


def _graph_cat(g, ids):
    """One node for the concatenation of ``ids``: nested concatenations are flattened and neighbouring constants
    merged into one constant vector."""
    flat = []
    for nid in ids:
        if g.nodes[nid][0] == "cat":
            flat.extend(g.nodes[nid][1])
        else:
            flat.append(nid)
    merged, run = [], []

    def flush():
        if len(run) == 1:
            merged.append(run[0][0])
        elif run:
            merged.append(g.cvec(np.concatenate([v for _, v in run])))
        del run[:]
    for nid in flat:
        node = g.nodes[nid]
        if node[0] == "cvec":
            run.append((nid, g.cvecs[node[1]]))
        elif node[0] == "const":
            run.append((nid, np.array([const_value(node)])))
        else:
            flush()
            merged.append(nid)
    flush()
    total = sum(1 if g.length[nid] is None else g.length[nid] for nid in merged)
    if len(merged) == 1 and g.length[merged[0]] is not None:
        return merged[0]
    return g.add(("cat", tuple(merged)), total)


def const_value(node):
    return float(np.frombuffer(node[1], dtype=np.float64)[0])


_FOLD_UN = {"neg": np.negative, "sqrt": np.sqrt, "exp": np.exp, "log": np.log, "sin": np.sin, "cos": np.cos,
            "tan": np.tan, "abs": np.absolute, "atan": np.arctan, "asin": np.arcsin, "acos": np.arccos,
            "tanh": np.tanh, "sinh": np.sinh, "cosh": np.cosh, "expm1": np.expm1, "log1p": np.log1p,
            "log2": np.log2, "log10": np.log10, "cbrt": np.cbrt}
_FOLD_BIN = {"add": np.add, "sub": np.subtract, "mul": np.multiply, "div": np.true_divide, "max": np.maximum,
             "min": np.minimum, "atan2": np.arctan2, "hypot": np.hypot, "mod": np.remainder}
_FOLD_CMP = {"lt": np.less, "le": np.less_equal, "gt": np.greater, "ge": np.greater_equal, "eq": np.equal,
             "ne": np.not_equal}


def concrete_value(g, nid, _memo=None):
    """The NumPy value of node ``nid`` when it does not depend on the decision vector (built from constants by
    slicing, indexing, concatenation and elementwise operations only), else None.  A callback may keep plain numbers in
    a buffer it got from ``np.zeros`` - a lookup table, a flag - and then ask for ``float(buf[i])``, ``if buf[0] > 0``,
    ``buf.astype(int)`` or hand it to a NumPy routine: such a buffer is a traced CONSTANT, and these requests have a
    concrete answer (they used to trace, when ``np.zeros`` still returned an ndarray)."""
    memo = {} if _memo is None else _memo
    if nid in memo:
        return memo[nid]
    node = g.nodes[nid]
    kind = node[0]
    out = None
    with np.errstate(all="ignore"):
        if kind == "const":
            out = np.float64(const_value(node))
        elif kind == "cvec":
            out = g.cvecs[node[1]]
        elif kind == "slice":
            v = concrete_value(g, node[1], memo)
            out = None if v is None else v[node[2]:node[2] + node[3]]
        elif kind == "idx":
            v = concrete_value(g, node[1], memo)
            out = None if v is None else v[node[2]]
        elif kind == "cat":
            parts = [concrete_value(g, c, memo) for c in node[1]]
            out = None if any(v is None for v in parts) else np.concatenate([np.atleast_1d(v) for v in parts])
        elif kind == "un" and node[1] in _FOLD_UN:
            v = concrete_value(g, node[2], memo)
            out = None if v is None else _FOLD_UN[node[1]](v)
        elif kind in ("bin", "cmp") and node[1] in (_FOLD_BIN if kind == "bin" else _FOLD_CMP):
            a, b = concrete_value(g, node[2], memo), concrete_value(g, node[3], memo)
            out = None if a is None or b is None else (_FOLD_BIN if kind == "bin" else _FOLD_CMP)[node[1]](a, b)
    memo[nid] = out
    return out


_UNARY = {
    np.negative: "neg", np.sqrt: "sqrt", np.exp: "exp", np.log: "log", np.sin: "sin",
    np.cos: "cos", np.tan: "tan", np.absolute: "abs", np.fabs: "abs", np.square: "square",
    np.positive: "pos", np.reciprocal: "recip", np.arctan: "atan", np.arcsin: "asin",
    np.arccos: "acos", np.tanh: "tanh", np.sinh: "sinh", np.cosh: "cosh", np.expm1: "expm1",
    np.log1p: "log1p", np.log2: "log2", np.log10: "log10", np.cbrt: "cbrt",
}
_BINARY = {
    np.add: "add", np.subtract: "sub", np.multiply: "mul", np.true_divide: "div",
    np.maximum: "max", np.minimum: "min", np.arctan2: "atan2", np.hypot: "hypot",
    np.remainder: "mod",                                 # np.mod is np.remainder: the sign of the divisor (Python's %)
}
_COMPARE = {
    np.less: "lt", np.less_equal: "le", np.greater: "gt", np.greater_equal: "ge",
    np.equal: "eq", np.not_equal: "ne",
}
_LOGICAL = {np.logical_and: "and", np.logical_or: "or", np.bitwise_and: "and", np.bitwise_or: "or"}

_DEG2RAD = float(np.pi / 180.0)    # NumPy: deg2rad(x) = x * (pi/180), rad2deg(x) = x * (180/pi)
_RAD2DEG = float(180.0 / np.pi)


def _norm_index(i, length):
    if length is None:
        raise TraceError("indexing a scalar")
    i = int(i)
    if i < 0:
        i += length
    if not 0 <= i < length:
        raise IndexError("index %d out of range for traced vector of length %d" % (i, length))
    return i


class Sym:
    """A traced value: scalar (``length is None``) or 1-D vector of ``length`` elements."""

    __array_priority__ = 1000.0

    # NumPy aliasing, reproduced: a Sym is a *name* for a node of the graph.  ``y = x`` makes both names one
    # object, so masked assignment and in-place arithmetic through either is seen by both (as with an ndarray);
    # ``+x``, ``x ** 1``, ``np.positive(x)`` and ``x.copy()`` give a NEW object (NumPy copies there); a slice
    # ``x[a:b]`` is a VIEW: writing through it rewrites the parent, and a later write to the parent shows in the
    # view (the view re-derives its node when the parent has moved on).
    def __init__(self, graph, nid, view_of=None):
        self.g = graph
        self._id = nid
        self._version = 0
        self._view_of = view_of            # (parent Sym, start, length) for slice views
        self._seen = view_of[0]._version if view_of else 0
        self._frozen = None                # why writing through this name cannot be traced (strided / column views)

    @property
    def id(self):
        v = self._view_of
        if v is not None:
            parent, start, ln = v
            pid = parent.id                      # a parent that is a view itself catches up first (and says so)
            if parent._version != self._seen:
                self._id = self.g.add(("slice", pid, start, ln), ln)
                self._seen = parent._version
                self._version += 1               # ... so that views of THIS view re-derive as well
        return self._id

    @id.setter
    def id(self, nid):
        """In-place change of what this name stands for (masked assignment, ``+=`` ...)."""
        if self._frozen:
            raise TraceError("writing through %s is not traceable (NumPy would change the array it views)"
                             % self._frozen)
        self._id = nid
        self._version += 1
        v = self._view_of
        if v is not None:
            parent, start, ln = v
            n = parent.length
            pieces = []
            if start > 0:
                pieces.append(self.g.add(("slice", parent.id, 0, start), start))
            pieces.append(nid)
            if start + ln < n:
                pieces.append(self.g.add(("slice", parent.id, start + ln, n - start - ln), n - start - ln))
            parent.id = pieces[0] if len(pieces) == 1 else self.g.add(("cat", tuple(pieces)), n)
            self._seen = parent._version

    # ------------------------------------------------------------------ basics
    @property
    def length(self):
        return self.g.length[self.id]

    @property
    def shape(self):
        return () if self.length is None else (self.length,)

    @property
    def ndim(self):
        return 0 if self.length is None else 1

    @property
    def size(self):
        return 1 if self.length is None else self.length

    def __len__(self):
        if self.length is None:
            raise TypeError("len() of traced scalar")
        return self.length

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def _concrete(self):
        """NumPy value when this expression is a constant of the trace (see :func:`concrete_value`), else None."""
        return concrete_value(self.g, self.id)

    def __bool__(self):
        v = self._concrete()
        if v is not None:
            return bool(v)                               # (NumPy's own rule for arrays of more than one element)
        raise TraceError("Python control flow on a traced value (if/while/and/or on the decision "
                         "variables) cannot be turned into a GPU kernel")

    def __float__(self):
        v = self._concrete()
        if v is not None:
            return float(v)
        raise TraceError("float() of a traced value: the callback needs a concrete number")

    def __int__(self):
        v = self._concrete()
        if v is not None:
            return int(v)
        raise TraceError("int() of a traced value: the callback needs a concrete number")

    def __index__(self):
        v = self._concrete()
        if v is not None and np.ndim(v) == 0 and float(v) == int(v):
            return int(v)
        raise TraceError("a traced value used as an index: the callback needs a concrete integer")

    def __array__(self, *a, **k):
        v = self._concrete()
        if v is not None:
            return np.array(v, *a, **{key: val for key, val in k.items() if key == "dtype"})
        raise TraceError("a traced value was passed to a NumPy routine the tracer does not "
                         "understand (only elementwise ufuncs, hstack/concatenate/append, where)")

    def __repr__(self):
        return "Sym(#%d, len=%s)" % (self.id, self.length)

    def copy(self):
        return Sym(self.g, self.id)

    # ndarray odds and ends that mean nothing special for a 1-D vector
    @property
    def T(self):
        return self

    def ravel(self, order="C"):
        return self

    def flatten(self, order="C"):
        return self.copy()

    def astype(self, dtype, **kw):
        if np.dtype(dtype) != np.float64:
            v = self._concrete()
            if v is not None:
                return np.asarray(v).astype(dtype, **kw)       # a constant-only buffer: a plain array again
            raise TraceError("astype(%s) of a traced value: callbacks are traced in float64" % (dtype,))
        return self.copy()

    def reshape(self, *shape):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        if tuple(shape) in ((-1,), (self.size,)):
            return self
        raise TraceError("reshape%r of a traced vector: only 1-D values are traceable" % (tuple(shape),))

    def __getattr__(self, name):
        # (only reached for attributes that do not exist) an ndarray method the tracer does not model must be a
        # tracing error like every other untraceable construct, not an AttributeError from nowhere
        if name.startswith("__") or name in ("g", "_id", "_version", "_view_of", "_seen", "_frozen"):
            raise AttributeError(name)
        raise TraceAttributeError("ndarray.%s is not traceable on a decision-variable expression" % name)

    # ------------------------------------------------------------------ lifting
    def _lift(self, other):
        """-> Sym or NotImplemented.  ndarray of size 1 is a scalar like NumPy broadcasting."""
        if isinstance(other, Sym):
            if other.g is not self.g:
                raise TraceError("mixing values from two different traces")
            return other
        if isinstance(other, (numbers.Real, np.bool_)):
            return Sym(self.g, self.g.const(other))
        if isinstance(other, np.ndarray):
            if other.ndim == 0 or other.size == 1 and other.ndim <= 1:
                return Sym(self.g, self.g.const(other.reshape(-1)[0]))
            if other.ndim == 1:
                return Sym(self.g, self.g.cvec(other))
            raise TraceError("only 1-D arrays can be combined with traced values")
        if isinstance(other, (list, tuple)):
            return self._lift(np.asarray(other, dtype=np.float64))
        return NotImplemented

    def _bcast_len(self, other):
        la, lb = self.length, other.length
        if la is None:
            return lb
        if lb is None or la == lb:
            return la
        if la == 1:
            return lb
        if lb == 1:
            return la
        raise TraceError("shape mismatch in traced elementwise op: %d vs %d" % (la, lb))

    def _unary(self, op):
        if op == "pos":
            return Sym(self.g, self.id)                 # NumPy: +x is a copy
        if op == "square":
            return self._binary("mul", self)
        if op == "recip":
            return Sym(self.g, self.g.const(1.0))._binary("div", self)
        return Sym(self.g, self.g.add(("un", op, self.id), self.length))

    def _binary(self, op, other, swap=False):
        if isinstance(other, SymMat):
            return NotImplemented                       # the matrix's reflected operator takes it
        other = self._lift(other)
        if other is NotImplemented:
            return NotImplemented
        a, b = (other, self) if swap else (self, other)
        length = a._bcast_len(b)
        return Sym(self.g, self.g.add(("bin", op, a.id, b.id), length))

    def _compare(self, op, other, swap=False):
        other = self._lift(other)
        if other is NotImplemented:
            return NotImplemented
        a, b = (other, self) if swap else (self, other)
        return Sym(self.g, self.g.add(("cmp", op, a.id, b.id), a._bcast_len(b), True))

    # ------------------------------------------------------------------ operators
    def __neg__(self): return self._unary("neg")
    def __pos__(self): return Sym(self.g, self.id)
    def __abs__(self): return self._unary("abs")
    def __add__(self, o): return self._binary("add", o)
    def __radd__(self, o): return self._binary("add", o, swap=True)
    def __sub__(self, o): return self._binary("sub", o)
    def __rsub__(self, o): return self._binary("sub", o, swap=True)
    def __mul__(self, o): return self._binary("mul", o)
    def __rmul__(self, o): return self._binary("mul", o, swap=True)
    def __truediv__(self, o): return self._binary("div", o)
    def __rtruediv__(self, o): return self._binary("div", o, swap=True)
    def __mod__(self, o): return self._binary("mod", o)
    def __rmod__(self, o): return self._binary("mod", o, swap=True)
    def __matmul__(self, o): return matmul(self, o)
    def __rmatmul__(self, o): return matmul(o, self)
    def __lt__(self, o): return self._compare("lt", o)
    def __le__(self, o): return self._compare("le", o)
    def __gt__(self, o): return self._compare("gt", o)
    def __ge__(self, o): return self._compare("ge", o)
    def __eq__(self, o): return self._compare("eq", o)
    def __ne__(self, o): return self._compare("ne", o)
    __hash__ = None                                     # elementwise ==, like ndarray: not hashable

    def _inplace(self, op, o):
        out = self._binary(op, o)
        if out is NotImplemented:
            return NotImplemented
        if out.length != self.length:
            raise TraceError("in-place operation would change the shape of a traced vector")
        self.id = out.id                                # every alias (and the parent of a view) sees it
        return self

    def __iadd__(self, o): return self._inplace("add", o)
    def __isub__(self, o): return self._inplace("sub", o)
    def __imul__(self, o): return self._inplace("mul", o)
    def __itruediv__(self, o): return self._inplace("div", o)
    def __and__(self, o): return self._logical("and", o)
    def __or__(self, o): return self._logical("or", o)

    def _logical(self, op, other):
        if not (isinstance(other, Sym) and self.g.isbool[self.id] and self.g.isbool[other.id]):
            raise TraceError("& and | are only traced between comparison results")
        return Sym(self.g, self.g.add(("logic", op, self.id, other.id),
                                      self._bcast_len(other), True))

    def __pow__(self, e):
        # NumPy's scalar-exponent fast paths (numpy/_core/src/multiarray/number.c fast_scalar_power):
        # 2 -> square (x*x), 0.5 -> sqrt, 1 -> +x, -1 -> reciprocal: reproduced bit for bit.  Every other
        # exponent goes to libm's pow() in NumPy, which has no bit-reproducible device twin; it is traced as
        # repeated multiplication for small integers (|e| <= 16: x*x*x, 1/(x*x*x)) and as exp(e * log(x))
        # otherwise - a few ulp from pow(), far inside the 1e-9 residual tolerance, NaN for a negative base
        # with a fractional exponent exactly like pow().
        if isinstance(e, (numbers.Real, np.ndarray)) and np.ndim(e) == 0:
            e = float(e)
            if e == 2.0:
                return self._binary("mul", self)
            if e == 1.0:
                return Sym(self.g, self.id)
            if e == 0.5:
                return self._unary("sqrt")
            if e == -1.0:
                return self._unary("recip")
            if e == 0.0:
                return _ones_like(self)
            if e == int(e) and abs(e) <= 16:
                k, base, acc = int(abs(e)), self, None
                while k:                                # square-and-multiply, left to right in the bits of k
                    if k & 1:
                        acc = base if acc is None else acc._binary("mul", base)
                    k >>= 1
                    if k:
                        base = base._binary("mul", base)
                return acc if e > 0 else acc._unary("recip")
            return self._unary("log")._binary("mul", e)._unary("exp")
        # a traced (or array-valued) exponent: one node, evaluated as exp(y log x) with pow's special cases
        # (csrc/og_math.h pow_: |y log x| ulp from libm's pow at worst)
        out = self._binary("pow", e)
        if out is NotImplemented:
            raise TraceError("x ** y: unsupported exponent %r" % (type(e),))
        return out

    def __rpow__(self, base):
        out = self._binary("pow", base, swap=True)
        if out is NotImplemented:
            raise TraceError("x ** y: unsupported base %r" % (type(base),))
        return out

    # ------------------------------------------------------------------ indexing
    def __getitem__(self, key):
        n = self.length
        if isinstance(key, (numbers.Integral, np.integer)):
            return Sym(self.g, self.g.add(("idx", self.id, _norm_index(key, n)), None))
        if isinstance(key, slice):
            if n is None:
                raise TraceError("slicing a traced scalar")
            start, stop, step = key.indices(n)
            if step != 1:
                # x[::-1], x[::2]: a copy-like gather of single elements (NumPy makes a view; nothing in a callback
                # writes through a strided view, and the tracer refuses it: the result is not assignable)
                out = take(self, list(range(start, stop, step)))
                if isinstance(out, Sym):
                    out._frozen = "a strided view x[%s:%s:%s]" % (key.start, key.stop, key.step)
                return out
            ln = max(0, stop - start)
            if start == 0 and ln == n:
                return Sym(self.g, self.id, view_of=(self, 0, n))
            return Sym(self.g, self.g.add(("slice", self.id, start, ln), ln), view_of=(self, start, ln))
        if isinstance(key, (list, tuple, np.ndarray)) and not isinstance(key, Sym):
            idx = np.asarray(key)
            if idx.ndim == 1 and idx.dtype == np.bool_:
                if idx.size != n:
                    raise IndexError("boolean index of length %d on a traced vector of length %d" % (idx.size, n))
                return take(self, np.nonzero(idx)[0].tolist())
            if idx.ndim == 1 and np.issubdtype(idx.dtype, np.integer):
                return take(self, idx.tolist())
        raise TraceError("unsupported index %r on a traced vector" % (key,))

    # ------------------------------------------------------------------ reductions (ndarray methods)
    @staticmethod
    def _axis_1d(axis, what):
        if axis not in (None, 0, -1):
            raise TraceError("%s(axis=%r) of a 1-D traced vector" % (what, axis))

    def sum(self, axis=None):
        self._axis_1d(axis, "sum")
        return pairwise_sum(self)

    def mean(self, axis=None):
        self._axis_1d(axis, "mean")
        if self.length is None:
            return Sym(self.g, self.id)
        return pairwise_sum(self)._binary("div", float(self.length))

    def dot(self, other):
        return dot(self, other)

    def prod(self, axis=None):
        self._axis_1d(axis, "prod")
        return prod(self)

    def min(self, axis=None):
        self._axis_1d(axis, "min")
        return reduce_tree(self, "min")

    def max(self, axis=None):
        self._axis_1d(axis, "max")
        return reduce_tree(self, "max")

    def cumsum(self, axis=None):
        self._axis_1d(axis, "cumsum")
        return cumsum(self)

    def fill(self, value):
        self._assign_range(0, self.size, value)

    def __setitem__(self, key, value):
        # in-place boolean-mask assignment: x[mask] = scalar  ->  x = where(mask, scalar, x)
        if isinstance(key, Sym) and self.g.isbool[key.id]:
            v = self._lift(value)
            if v is NotImplemented or v.length not in (None, 1):
                raise TraceError("masked assignment needs a scalar right-hand side")
            if key.length not in (None, 1, self.length):
                raise TraceError("mask length mismatch")
            self.id = self.g.add(("where", key.id, v.id, self.id), self.length)
            return
        if isinstance(key, Sym):
            raise TraceError("assignment through a traced index is not traceable (only a traced comparison as a mask)")
        n = self.length
        if n is None:
            raise TraceError("item assignment to a traced scalar")
        if isinstance(key, type(Ellipsis)):
            key = slice(None)
        if isinstance(key, tuple):
            if len(key) == 1:
                return self.__setitem__(key[0], value)
            raise TraceError("%d indices on a 1-D traced vector" % len(key))
        if isinstance(key, (numbers.Integral, np.integer)):
            i = _norm_index(key, n)
            self._assign_range(i, 1, value, scalar_target=True)
            return
        if isinstance(key, slice):
            start, stop, step = key.indices(n)
            if step == 1:
                self._assign_range(start, max(0, stop - start), value)
                return
            targets = list(range(start, stop, step))
        else:
            idx = np.asarray(key)
            if idx.ndim == 1 and idx.dtype == np.bool_:
                if idx.size != n:
                    raise IndexError("boolean index of length %d on a traced vector of length %d" % (idx.size, n))
                targets = np.nonzero(idx)[0].tolist()
            elif idx.ndim == 1 and np.issubdtype(idx.dtype, np.integer):
                targets = [_norm_index(i, n) for i in idx.tolist()]
            else:
                raise TraceError("unsupported index %r in an assignment to a traced vector" % (key,))
        # scattered targets: element by element, in order (NumPy: the last write to a repeated index wins)
        v = self._lift(value)
        if v is NotImplemented:
            raise TraceError("cannot assign %r into a traced vector" % (type(value),))
        if v.length not in (None, 1, len(targets)):
            raise ValueError("shape mismatch: value of length %d assigned to %d elements" % (v.length, len(targets)))
        for k, i in enumerate(targets):
            self._assign_range(i, 1, v if v.length is None else v[0 if v.length == 1 else k], scalar_target=True)

    def _assign_range(self, start, count, value, scalar_target=False):
        """``self[start:start+count] = value`` (NumPy broadcasting of a scalar / one-element value)."""
        g, n = self.g, self.length
        if count == 0:
            return
        v = self._lift(value)
        if v is NotImplemented:
            if isinstance(value, SymMat):
                raise TraceError("assigning a 2-D traced value into a 1-D traced vector")
            raise TraceError("cannot assign %r into a traced vector" % (type(value),))
        if v.length is None or (v.length == 1 and count != 1):
            vid = v.id if v.length is None else g.add(("idx", v.id, 0), None)
            if count == 1:
                new = [vid]
            elif g.nodes[vid][0] == "const":
                new = [g.cvec(np.full(count, const_value(g.nodes[vid])))]
            else:
                # a traced scalar spread over the range: s * 1.0 is s for every s (NaN and -0.0 included)
                new = [Sym(g, g.cvec(np.ones(count)))._binary("mul", Sym(g, vid)).id]
        elif v.length == count:
            new = [g.add(("idx", v.id, 0), None)] if (scalar_target and count == 1) else _graph_parts(g, v.id, 0, count)
        else:
            raise ValueError("could not broadcast a traced value of length %d into %d elements" % (v.length, count))
        cur = self.id
        ids = _graph_parts(g, cur, 0, start) + new + _graph_parts(g, cur, start + count, n - start - count)
        self.id = _graph_cat(g, ids)

    # ------------------------------------------------------------------ NumPy protocol
    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__" or kwargs.get("out") is not None:
            raise TraceError("ufunc method %s.%s is not traceable" % (ufunc.__name__, method))
        if ufunc in _UNARY:
            return inputs[0]._unary(_UNARY[ufunc])
        if ufunc is np.deg2rad or ufunc is np.radians:
            return inputs[0]._binary("mul", _DEG2RAD)
        if ufunc is np.rad2deg or ufunc is np.degrees:
            return inputs[0]._binary("mul", _RAD2DEG)
        if ufunc is np.sign:
            x = inputs[0]
            return where(x > 0.0, 1.0, where(x < 0.0, -1.0, x._binary("mul", 0.0)))
        if any(isinstance(v, SymMat) for v in inputs):
            return SymMat._ufunc(ufunc, inputs)
        if ufunc is np.heaviside:
            x, h0 = inputs
            if not isinstance(x, Sym):
                raise TraceError("np.heaviside with a traced second argument only")
            # NumPy: NaN -> NaN, 0 -> h0, < 0 -> 0, > 0 -> 1
            return where(x < 0.0, 0.0, where(x > 0.0, 1.0, where(x == 0.0, h0, x)))
        if ufunc is np.matmul:
            return matmul(inputs[0], inputs[1])
        if ufunc is np.fmod:
            a, b = inputs
            me, other, swap = (a, b, False) if isinstance(a, Sym) else (b, a, True)
            out = me._binary("fmod", other, swap)
            if out is NotImplemented:
                raise TraceError("unsupported operand for fmod")
            return out
        if ufunc in (np.fmax, np.fmin):                 # (NaN handling aside: a NaN row is caught either way)
            ufunc = np.maximum if ufunc is np.fmax else np.minimum
        if ufunc is np.power:
            if isinstance(inputs[0], Sym):
                return inputs[0].__pow__(inputs[1])
            return inputs[1].__rpow__(inputs[0])
        if ufunc is np.float_power:
            return np.power(*inputs)
        if ufunc is np.exp2:
            return np.power(2.0, inputs[0])
        table = _BINARY if ufunc in _BINARY else _COMPARE if ufunc in _COMPARE else \
            _LOGICAL if ufunc in _LOGICAL else None
        if table is None:
            raise TraceError("NumPy ufunc %s is not traceable" % ufunc.__name__)
        a, b = inputs
        if isinstance(a, Sym):
            me, other, swap = a, b, False
        else:
            me, other, swap = b, a, True
        if table is _BINARY:
            out = me._binary(table[ufunc], other, swap)
        elif table is _COMPARE:
            out = me._compare(table[ufunc], other, swap)
        else:
            out = me._logical(table[ufunc], other)
        if out is NotImplemented:
            raise TraceError("unsupported operand for %s" % ufunc.__name__)
        return out

    def __array_function__(self, func, types, args, kwargs):
        return _array_function(func, args, kwargs)


def _axis_of(kwargs, args, pos):
    """The ``axis`` of a reduction call (keyword or positional); every other keyword is refused by name."""
    extra = set(kwargs) - {"axis"}
    if extra:
        raise TraceError("keyword%s %s of a NumPy reduction are not traceable"
                         % ("s" if len(extra) > 1 else "", ", ".join(sorted(extra))))
    if "axis" in kwargs:
        return kwargs["axis"]
    return args[pos] if len(args) > pos else None


def _array_function(func, args, kwargs):
    """NumPy functions (``__array_function__`` protocol) on traced vectors and matrices."""
    name = getattr(func, "__name__", str(func))
    # constants of the trace (buffers that only ever received plain numbers) go in as the arrays they are; a call whose
    # traced arguments are ALL constants is NumPy's own call
    def plain(v, depth=0):
        if isinstance(v, Sym):
            c = v._concrete()
            return v if c is None else np.array(c)
        if isinstance(v, (list, tuple)) and depth < 2 and any(isinstance(it, Sym) for it in v):
            return type(v)(plain(it, depth + 1) for it in v)
        return v
    p_args = tuple(plain(v) for v in args)
    p_kwargs = {k: plain(v) for k, v in kwargs.items()}
    # ... except for the routines that BUILD the array a callback goes on to fill (`out = np.concatenate((np.zeros(2),
    # np.zeros(3))); out[1:] = traced`): their result has to stay a traced constant that records assignments - a plain
    # ndarray would hand the traced value to NumPy's own __setitem__
    builders = (np.hstack, np.concatenate, np.append, np.vstack, getattr(np, "row_stack", None), np.stack,
                np.column_stack, np.zeros_like, np.ones_like, np.empty_like, np.full_like)
    if func not in builders and not _has_traced(list(p_args)) and not _has_traced(list(p_kwargs.values())):
        return func(*p_args, **p_kwargs)
    if func in (np.hstack, np.concatenate):
        axis = kwargs.get("axis", 0) if func is np.concatenate else 0
        items = list(args[0])
        if any(isinstance(it, SymMat) for it in items):
            if func is np.concatenate and axis in (0, -2):
                return SymMat.vstack(items)
            raise TraceError("np.%s of 2-D traced values along axis %r is not traceable" % (name, axis))
        if axis not in (0, -1, None):
            raise TraceError("np.concatenate(axis=%r) of 1-D traced values" % (axis,))
        return cat(items)
    if func is np.append:
        return cat([args[0], args[1]])
    if func in (np.vstack, getattr(np, "row_stack", None)):
        return SymMat.vstack(list(args[0]))
    if func is np.stack:
        axis = kwargs.get("axis", args[1] if len(args) > 1 else 0)
        mat = SymMat.vstack(list(args[0]), rows_only=True)
        if axis in (0, -2):
            return mat
        if axis in (1, -1):
            return mat.T
        raise TraceError("np.stack(axis=%r) of traced vectors" % (axis,))
    if func is np.column_stack:
        return SymMat.vstack(list(args[0]), rows_only=True).T
    if func is np.where and len(args) == 3:
        return where(*args)
    if func is np.clip:
        x, lo, hi = args[0], args[1], args[2]
        return np.minimum(np.maximum(x, lo), hi)
    if func in (np.sum, np.mean, np.prod, np.min, np.amin, np.max, np.amax) and len(args) >= 1:
        axis = _axis_of(kwargs, args, 1)
        op = {np.sum: "sum", np.mean: "mean", np.prod: "prod", np.min: "min", np.amin: "min", np.max: "max",
              np.amax: "max"}[func]
        x = args[0]
        if isinstance(x, SymMat):
            return x._reduce(op, axis)
        if axis not in (None, 0, -1):
            raise TraceError("np.%s(axis=%r) of a 1-D traced vector" % (name, axis))
        return getattr(x, op)()
    if func is np.dot and len(args) == 2 and not kwargs:
        return dot(args[0], args[1])
    if func is np.matmul and len(args) == 2 and not kwargs:
        return matmul(args[0], args[1])
    if func is np.cumsum and len(args) == 1 and not kwargs:
        return cumsum(args[0])
    if func is np.roll and len(args) == 2 and not kwargs:
        return roll(args[0], args[1])
    if func is np.flip and len(args) == 1 and not kwargs:
        return args[0][::-1]
    if func is np.take and len(args) == 2 and not kwargs:
        return args[0][np.asarray(args[1])]
    if func is np.interp and len(args) == 3 and not kwargs:
        return np_interp(args[0], args[1], args[2])
    if func is np.diff and len(args) == 1 and not kwargs:
        return args[0][1:] - args[0][:-1]
    if func in (getattr(np, "trapz", None), getattr(np, "trapezoid", None)) and func is not None:
        return trapezoid(*args, **kwargs)
    if func in (np.zeros_like, np.ones_like, np.empty_like, np.full_like):
        extra = set(kwargs) - {"dtype", "fill_value"}
        if extra or (kwargs.get("dtype") is not None and np.dtype(kwargs["dtype"]) != np.float64):
            raise TraceError("np.%s of a traced value with %s" % (name, ", ".join(sorted(kwargs))))
        fill = {np.zeros_like: 0.0, np.ones_like: 1.0, np.empty_like: 0.0}.get(func)
        if func is np.full_like:
            fill = kwargs.get("fill_value", args[1] if len(args) > 1 else None)
        return _filled_like(args[0], fill)
    if func in (np.shape,):
        return args[0].shape
    if func in (np.size,):
        return args[0].size
    if func in (np.ndim,):
        return args[0].ndim
    if func is np.copy:
        return args[0].copy()
    if func is np.ravel:
        return args[0].ravel()
    if func is np.transpose and len(args) == 1:
        return args[0].T
    if func is np.atleast_1d and len(args) == 1:
        x = args[0]
        return cat([x]) if isinstance(x, Sym) and x.length is None else x
    if func is np.squeeze and len(args) == 1 and not kwargs:
        x = args[0]
        return x[0] if isinstance(x, Sym) and x.length == 1 else x
    raise TraceError("NumPy function %s is not traceable" % name)


def _ones_like(s):
    if s.length is None:
        return Sym(s.g, s.g.const(1.0))
    return Sym(s.g, s.g.cvec(np.ones(s.length)))


def is_sym(x):
    return isinstance(x, Sym)


def _find_graph(items):
    for it in items:
        if isinstance(it, Sym):
            return it.g
    return None


def cat(items):
    """``np.hstack`` that also understands traced values (scalars count as one element)."""
    g = _find_graph(items)
    if g is None:                         # plain NumPy: exactly what the reference does
        return np.hstack(items) if len(items) else np.zeros(0)
    ids, total = [], 0
    probe = Sym(g, g.const(0.0))
    for it in items:
        if isinstance(it, np.ndarray) and it.size == 0:
            continue
        if isinstance(it, (list, tuple)):
            it = np.asarray(it, dtype=np.float64)
            if it.size == 0:
                continue
        s = probe._lift(it) if not isinstance(it, Sym) else it
        if s is NotImplemented:
            raise TraceError("cannot concatenate %r with traced values" % (type(it),))
        if isinstance(it, np.ndarray) and it.ndim == 1 and it.size == 1:
            # keep a length-1 *vector* (not a broadcastable scalar) for layout purposes
            pass
        n = 1 if s.length is None else s.length
        if n == 0:
            continue
        ids.append(s.id)
        total += n
    if len(ids) == 1 and g.length[ids[0]] is not None:
        return Sym(g, ids[0])
    return Sym(g, g.add(("cat", tuple(ids)), total))


def where(cond, a, b):
    g = _find_graph([cond, a, b])
    if g is None:
        return np.where(cond, a, b)
    probe = Sym(g, g.const(0.0))
    c, a, b = (probe._lift(v) for v in (cond, a, b))
    if not g.isbool[c.id]:
        raise TraceError("np.where condition must be a traced comparison")
    length = c._bcast_len(a)
    length2 = a._bcast_len(b)
    length = length if length2 is None else (length2 if length is None else max(length, length2))
    return Sym(g, g.add(("where", c.id, a.id, b.id), length))


def take(vec, indices):
    """``vec[[i0, i1, ...]]``: the elements one by one, concatenated (a fancy-index result is a copy in NumPy too)."""
    n = vec.length
    if n is None:
        raise TraceError("indexing a scalar")
    items = [vec[_norm_index(i, n)] for i in indices]
    if not items:
        return np.zeros(0)
    return cat(items)


def pairwise_sum(vec):
    """``np.sum`` / ``ndarray.sum`` of a traced vector with NumPy's own order of additions: ``pairwise_sum`` of
    numpy/_core/src/umath/loops_utils.h.src - fewer than 8 elements left to right; up to 128 eight running sums
    over strides of eight, combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the tail left to right; above 128
    the halves (the first rounded down to a multiple of 8) recursively.  Expanded into additions of single elements:
    the traced value equals NumPy's bit for bit (tests/test_oracle_and_codegen.py)."""
    if not isinstance(vec, Sym):
        return np.sum(vec)
    n = vec.length
    if n is None:
        return Sym(vec.g, vec.id)
    elems = [vec[i] for i in range(n)]

    def pw(lo, cnt):
        if cnt < 8:
            # NumPy starts this loop from -0.0, which changes nothing except the sign of an all-(-0.0) sum
            acc = elems[lo] if cnt else Sym(vec.g, vec.g.const(0.0))
            for i in range(1, cnt):
                acc = acc + elems[lo + i]
            return acc
        if cnt <= 128:
            r = [elems[lo + k] for k in range(8)]
            i = 8
            while i < cnt - (cnt % 8):
                for k in range(8):
                    r[k] = r[k] + elems[lo + i + k]
                i += 8
            res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
            while i < cnt:
                res = res + elems[lo + i]
                i += 1
            return res
        half = cnt // 2
        half -= half % 8
        return pw(lo, half) + pw(lo + half, cnt - half)

    if n == 0:
        return Sym(vec.g, vec.g.const(0.0))
    return pw(0, n)


def dot(a, b):
    """1-D ``np.dot``: NumPy hands this to BLAS ddot, whose order of additions belongs to the BLAS build (SIMD
    width, unrolling); traced as ``np.sum(a * b)`` - the same value to a few ulp of the sum of magnitudes, not bit
    for bit (tests state the bound)."""
    g = _find_graph([a, b])
    if g is None:
        return np.dot(a, b)
    probe = Sym(g, g.const(0.0))
    a, b = (probe._lift(v) for v in (a, b))
    if a is NotImplemented or b is NotImplemented:
        raise TraceError("np.dot: unsupported operand")
    prod = a * b
    return prod if prod.length is None else pairwise_sum(prod)


def reduce_tree(vec, op):
    """``np.min`` / ``np.max``: any order of the comparisons gives the same number (NaN propagates through
    maximum / minimum either way)."""
    if not isinstance(vec, Sym):
        return np.min(vec) if op == "min" else np.max(vec)
    n = vec.length
    if n is None:
        return Sym(vec.g, vec.id)
    if n == 0:
        raise ValueError("zero-size array to reduction operation which has no identity")
    level = [vec[i] for i in range(n)]
    while len(level) > 1:
        nxt = [level[i]._binary(op, level[i + 1]) for i in range(0, len(level) - 1, 2)]
        if len(level) % 2:
            nxt.append(level[-1])
        level = nxt
    return level[0]


def cumsum(vec):
    """``np.cumsum``: running sums left to right (``add.accumulate``)."""
    if not isinstance(vec, Sym):
        return np.cumsum(vec)
    n = vec.length
    if n is None:
        return cat([vec])
    out, acc = [], None
    for i in range(n):
        acc = vec[i] if acc is None else acc + vec[i]
        out.append(acc)
    return cat(out)


def roll(vec, shift):
    if not isinstance(vec, Sym):
        return np.roll(vec, shift)
    n = vec.length
    if n is None or n == 0:
        return vec
    k = int(shift) % n
    if k == 0:
        return +vec
    return cat([vec[n - k:], vec[:n - k]])


def np_interp(x, xp, fp):
    """``np.interp(x, xp, fp)`` with a traced ``x`` and constant tables: piecewise linear, the end values outside
    ``[xp[0], xp[-1]]`` - the table-lookup node of ``interp1d`` with those fills (same formula
    ``slope * (x - x_lo) + y_lo``; NumPy's own loop differs from it in the last bit at some points)."""
    if not isinstance(x, Sym):
        return np.interp(x, xp, fp)
    if isinstance(xp, Sym) or isinstance(fp, Sym):
        raise TraceError("np.interp with traced tables is not traceable")
    xp = np.asarray(xp, dtype=np.float64)
    fp = np.asarray(fp, dtype=np.float64)
    return interp_linear(xp, fp, 0, fp[0], fp[-1], x)


def matvec(phase, operand):
    """Collocation derivative ``D[phase] @ operand`` as a single traced node."""
    return Sym(operand.g, operand.g.add(("mv", int(phase), operand.id), operand.length))


def seqsum(vec):
    """Python's builtin ``sum``: left-to-right, starting from int 0 (reference
    ``OpenGoddard/optimize.py:708``)."""
    return Sym(vec.g, vec.g.add(("seqsum", vec.id), None))


def interp_linear(xgrid, ygrid, mode, fill_below, fill_above, arg):
    """Traced ``scipy.interpolate.interp1d(kind="linear")`` call on a traced argument."""
    g = arg.g
    xg = np.ascontiguousarray(xgrid, dtype=np.float64)
    yg = np.ascontiguousarray(ygrid, dtype=np.float64)
    if xg.ndim != 1 or yg.shape != xg.shape or xg.size < 2:
        raise TraceError("only 1-D linear interp1d tables are traceable")
    if np.any(np.diff(xg) <= 0):
        raise TraceError("interp1d grid must be strictly increasing to be traceable")
    ix = g.nodes[g.cvec(xg)][1]
    iy = g.nodes[g.cvec(yg)][1]
    key = (ix, iy, int(mode), np.float64(fill_below).tobytes(), np.float64(fill_above).tobytes())
    if key not in g.tables:
        g.tables.append(key)
    tid = g.tables.index(key)
    return Sym(g, g.add(("interp", tid, arg.id), arg.length))


class intercept_interp1d:
    """While tracing, route ``interp1d_object(traced_value)`` to :func:`interp_linear` (the
    shipped example 11 calls SciPy interpolants inside its callbacks)."""

    def __enter__(self):
        try:
            from scipy.interpolate import _polyint
        except Exception:
            self._cls = None
            return self
        self._cls = _polyint._Interpolator1D
        self._orig = self._cls.__call__
        orig = self._orig

        def call(interp, x):
            if not isinstance(x, Sym):
                return orig(interp, x)
            if getattr(interp, "_kind", None) != "linear" or np.ndim(interp.y) != 1:
                raise TraceError("only linear, 1-D scipy.interpolate.interp1d objects are traceable")
            if getattr(interp, "_extrapolate", False):
                mode, lo, hi = 1, 0.0, 0.0
            elif interp.bounds_error:
                mode, lo, hi = 2, np.nan, np.nan
            else:
                mode = 0
                lo = float(np.asarray(interp._fill_value_below).reshape(-1)[0])
                hi = float(np.asarray(interp._fill_value_above).reshape(-1)[0])
            return interp_linear(interp.x, interp.y, mode, lo, hi, x)
        self._cls.__call__ = call
        return self

    def __exit__(self, *exc):
        if self._cls is not None:
            self._cls.__call__ = self._orig
        return False


def prod(vec):
    """``np.prod``: ``multiply.reduce`` runs left to right (only additions are summed pairwise in NumPy)."""
    if not isinstance(vec, Sym):
        return np.prod(vec)
    n = vec.length
    if n is None:
        return Sym(vec.g, vec.id)
    if n == 0:
        return Sym(vec.g, vec.g.const(1.0))
    acc = vec[0]
    for i in range(1, n):
        acc = acc * vec[i]
    return acc


def matmul(a, b):
    """``a @ b`` with traced operands: vector @ vector (as ``np.dot``), constant matrix @ traced vector, traced
    vector @ constant matrix - each entry one ``np.dot`` (BLAS order in NumPy: equal to rounding, not bit for bit)."""
    if isinstance(a, SymMat) or isinstance(b, SymMat):
        raise TraceError("@ with a 2-D traced operand is not traceable (constant matrix @ traced vector is)")
    if isinstance(a, Sym) and a.length is None or isinstance(b, Sym) and b.length is None:
        raise ValueError("matmul: input operand does not have enough dimensions")
    if isinstance(a, Sym) and isinstance(b, Sym):
        return dot(a, b)
    if isinstance(b, Sym):
        A = np.asarray(a, dtype=np.float64)
        if A.ndim == 1:
            return dot(A, b)
        if A.ndim == 2 and A.shape[1] == b.length:
            return cat([dot(A[i], b) for i in range(A.shape[0])])
    else:
        B = np.asarray(b, dtype=np.float64)
        if B.ndim == 1:
            return dot(a, B)
        if B.ndim == 2 and B.shape[0] == a.length:
            return cat([dot(a, np.ascontiguousarray(B[:, j])) for j in range(B.shape[1])])
    raise TraceError("@: only 1-D traced vectors with constant 1-D / 2-D operands of matching shape are traceable")


def trapezoid(y, x=None, dx=1.0, axis=-1):
    """``np.trapz`` / ``np.trapezoid`` of 1-D data, NumPy's own formula and order:
    ``(d * (y[1:] + y[:-1]) / 2.0).sum()`` with ``d = np.diff(x)`` or the scalar ``dx``."""
    if isinstance(y, SymMat) or isinstance(x, SymMat):
        raise TraceError("np.trapz of a 2-D traced value")
    if axis not in (-1, 0):
        raise TraceError("np.trapz(axis=%r) of a 1-D traced vector" % (axis,))
    if not isinstance(y, Sym):
        g = _find_graph([x, dx])
        if g is None:
            return (getattr(np, "trapezoid", None) or np.trapz)(y, x=x, dx=dx)
        y = Sym(g, g.const(0.0))._lift(np.asarray(y, dtype=np.float64))
    if x is None:
        d = dx
    else:
        xs = x if isinstance(x, Sym) else np.asarray(x, dtype=np.float64)
        d = xs[1:] - xs[:-1]
    return pairwise_sum(d * (y[1:] + y[:-1]) / 2.0)


def _filled_like(like, fill):
    if isinstance(like, SymMat):
        return SymMat([_filled_like(r, fill) for r in like.rows])
    g = like.g
    if isinstance(fill, Sym):
        base = Sym(g, g.cvec(np.ones(like.size))) if like.length is not None else Sym(g, g.const(1.0))
        return base * fill
    if like.length is None:
        return Sym(g, g.const(float(fill)))
    return Sym(g, g.cvec(np.full(like.length, float(fill))))


class SymMat:
    """A small 2-D traced array: a list of equally long traced rows.  Enough for what callbacks do with one - build it
    (``np.array([a, b])``, ``np.vstack``, ``np.stack``, ``np.zeros((r, n))`` filled row by row), index it, transpose it,
    combine it elementwise and reduce it along an axis, in NumPy's order of operations.  ``m[i]`` IS the row (writes go
    through, as with NumPy's view); a column ``m[:, j]`` and the transpose are read-only copies here."""

    __array_priority__ = 1001.0

    def __init__(self, rows):
        rows = list(rows)
        if not rows:
            raise TraceError("an empty 2-D traced array")
        self.g = rows[0].g
        width = rows[0].length
        for r in rows:
            if not isinstance(r, Sym) or r.length is None or r.length != width:
                raise TraceError("the rows of a 2-D traced array must be traced vectors of one length")
        self.rows = rows

    # -------------------------------------------------------------- construction
    @staticmethod
    def _row(g, item, width=None):
        probe = Sym(g, g.const(0.0))
        if isinstance(item, Sym):
            if item.length is None:
                if width is None:
                    raise TraceError("a traced scalar where a row was expected")
                return _filled_like(Sym(g, g.cvec(np.zeros(width))), item)
            return item
        arr = np.asarray(item, dtype=np.float64)
        if arr.ndim == 0 and width is not None:
            return Sym(g, g.cvec(np.full(width, float(arr))))
        if arr.ndim != 1:
            raise TraceError("cannot make a row of a 2-D traced array out of an array of shape %r" % (arr.shape,))
        return Sym(g, g.cvec(arr))

    @classmethod
    def vstack(cls, items, rows_only=False):
        g = None
        for it in items:
            if isinstance(it, (Sym, SymMat)):
                g = it.g
                break
        rows = []
        for it in items:
            if isinstance(it, SymMat):
                if rows_only:
                    raise TraceError("np.stack of 2-D traced arrays")
                rows.extend(Sym(g, r.id) for r in it.rows)
            elif isinstance(it, Sym):
                rows.append(Sym(g, it.id) if it.length is not None else cat([it]))
            else:
                arr = np.asarray(it, dtype=np.float64)
                if arr.ndim == 2 and not rows_only:
                    rows.extend(cls._row(g, arr[i]) for i in range(arr.shape[0]))
                elif arr.ndim == 0:
                    rows.append(Sym(g, g.cvec(arr.reshape(1))))
                else:
                    rows.append(cls._row(g, arr))
        return cls(rows)

    @classmethod
    def filled(cls, g, shape, fill):
        return cls([Sym(g, g.cvec(np.full(int(shape[1]), float(fill)))) for _ in range(int(shape[0]))])

    # -------------------------------------------------------------- basics
    @property
    def shape(self):
        return (len(self.rows), self.rows[0].length)

    ndim = 2

    @property
    def size(self):
        return len(self.rows) * self.rows[0].length

    def __len__(self):
        return len(self.rows)

    def __iter__(self):
        return iter(self.rows)

    def __bool__(self):
        raise TraceError("Python control flow on a traced value")

    def __array__(self, *a, **k):
        raise TraceError("a 2-D traced value was passed to a NumPy routine the tracer does not understand")

    def __repr__(self):
        return "SymMat(%d x %d)" % self.shape

    def copy(self):
        return SymMat([r.copy() for r in self.rows])

    @property
    def T(self):
        r, c = self.shape
        cols = []
        for j in range(c):
            col = cat([self.rows[i][j] for i in range(r)])
            col._frozen = "a transposed view m.T"
            cols.append(col)
        return SymMat(cols)

    def transpose(self):
        return self.T

    def ravel(self, order="C"):
        if order != "C":
            raise TraceError("ravel(order=%r) of a 2-D traced array" % (order,))
        return cat([Sym(self.g, r.id) for r in self.rows])

    flatten = ravel

    def reshape(self, *shape):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        if tuple(shape) in ((-1,), (self.size,)):
            return self.ravel()
        if tuple(shape) == self.shape:
            return self
        raise TraceError("reshape%r of a 2-D traced array" % (tuple(shape),))

    def __getattr__(self, name):
        if name.startswith("__") or name in ("g", "rows"):
            raise AttributeError(name)
        raise TraceAttributeError("ndarray.%s is not traceable on a 2-D traced array" % name)

    # -------------------------------------------------------------- indexing
    def _row_index(self, i):
        return _norm_index(i, len(self.rows))

    def __getitem__(self, key):
        if isinstance(key, (numbers.Integral, np.integer)):
            return self.rows[self._row_index(key)]
        if isinstance(key, slice):
            picked = self.rows[key]
            return SymMat(picked) if picked else np.zeros((0, self.shape[1]))
        if isinstance(key, tuple) and len(key) == 2:
            ri, ci = key
            if isinstance(ri, (numbers.Integral, np.integer)):
                return self.rows[self._row_index(ri)][ci]
            if isinstance(ri, slice):
                picked = self.rows[ri]
                if isinstance(ci, (numbers.Integral, np.integer)):
                    col = cat([r[ci] for r in picked])
                    if isinstance(col, Sym):
                        col._frozen = "a column view m[:, j]"
                    return col
                if isinstance(ci, slice) and ci == slice(None):
                    return SymMat(picked)
                sub = [r[ci] for r in picked]
                for r in sub:
                    if isinstance(r, Sym):
                        r._frozen = "a sub-block view m[a:b, c:d]"
                return SymMat(sub)
        raise TraceError("unsupported index %r on a 2-D traced array" % (key,))

    def __setitem__(self, key, value):
        if isinstance(key, (numbers.Integral, np.integer)):
            self.rows[self._row_index(key)][:] = value
            return
        if isinstance(key, slice):
            key = (key, slice(None))
        if isinstance(key, tuple) and len(key) == 2:
            ri, ci = key
            if isinstance(ri, (numbers.Integral, np.integer)):
                self.rows[self._row_index(ri)][ci] = value
                return
            if isinstance(ri, slice):
                picked = self.rows[ri]
                if isinstance(value, SymMat):
                    if len(value.rows) != len(picked):
                        raise ValueError("shape mismatch in an assignment to a 2-D traced array")
                    for r, v in zip(picked, value.rows):
                        r[ci] = v
                    return
                if isinstance(ci, (numbers.Integral, np.integer)):
                    v = value if isinstance(value, Sym) else np.asarray(value, dtype=np.float64)
                    scalar = (isinstance(v, Sym) and v.length is None) or (not isinstance(v, Sym) and v.ndim == 0)
                    if not scalar and len(v) != len(picked):
                        raise ValueError("shape mismatch in an assignment to a column of a 2-D traced array")
                    for k, r in enumerate(picked):
                        r[ci] = v if scalar else v[k]
                    return
                arr = value if isinstance(value, Sym) else np.asarray(value, dtype=np.float64)
                if not isinstance(arr, Sym) and arr.ndim == 2:
                    if arr.shape[0] != len(picked):
                        raise ValueError("shape mismatch in an assignment to a 2-D traced array")
                    for r, v in zip(picked, arr):
                        r[ci] = v
                    return
                for r in picked:                         # one row (or a scalar) broadcast over the rows
                    r[ci] = arr
                return
        raise TraceError("unsupported index %r in an assignment to a 2-D traced array" % (key,))

    # -------------------------------------------------------------- elementwise
    def _operand_rows(self, other):
        """The other operand of an elementwise operation, row by row (NumPy broadcasting against (r, c))."""
        r, c = self.shape
        if isinstance(other, SymMat):
            if other.shape == (r, c):
                return other.rows
            if other.shape == (1, c):
                return [other.rows[0]] * r
            if other.shape == (r, 1):
                return [row[0] for row in other.rows]
            raise TraceError("shape mismatch in a traced elementwise operation: %r vs %r" % (self.shape, other.shape))
        if isinstance(other, Sym):
            if other.length in (None, 1, c):
                return [other] * r
            raise TraceError("shape mismatch in a traced elementwise operation: %r vs (%d,)" % (self.shape, other.length))
        if isinstance(other, (numbers.Real, np.bool_)):
            return [other] * r
        arr = np.asarray(other, dtype=np.float64)
        if arr.ndim <= 1 and arr.size in (1, c):
            return [arr if arr.size == c and arr.ndim == 1 else float(arr.reshape(-1)[0])] * r
        if arr.ndim == 2 and arr.shape == (r, c):
            return [arr[i] for i in range(r)]
        if arr.ndim == 2 and arr.shape == (r, 1):
            return [float(arr[i, 0]) for i in range(r)]
        if arr.ndim == 2 and arr.shape == (1, c):
            return [arr[0]] * r
        raise TraceError("shape mismatch in a traced elementwise operation: %r vs %r" % (self.shape, arr.shape))

    def _map(self, fn, other=None, swap=False):
        if other is None:
            return SymMat([fn(row) for row in self.rows])
        others = self._operand_rows(other)
        out = []
        for row, o in zip(self.rows, others):
            v = fn(o, row) if swap else fn(row, o)
            if v is NotImplemented:
                raise TraceError("unsupported operand in a 2-D traced operation")
            out.append(v if v.length is not None else cat([v]))
        return SymMat(out)

    @staticmethod
    def _ufunc(ufunc, inputs):
        if len(inputs) == 1:
            return inputs[0]._map(lambda r: ufunc(r))
        a, b = inputs
        if isinstance(a, SymMat):
            return a._map(lambda x, y: ufunc(x, y), b)
        return b._map(lambda x, y: ufunc(x, y), a, swap=True)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__" or kwargs.get("out") is not None:
            raise TraceError("ufunc method %s.%s is not traceable" % (ufunc.__name__, method))
        return SymMat._ufunc(ufunc, inputs)

    def __array_function__(self, func, types, args, kwargs):
        return _array_function(func, args, kwargs)

    def __neg__(self): return self._map(lambda r: -r)
    def __pos__(self): return self.copy()
    def __abs__(self): return self._map(abs)
    def __add__(self, o): return self._map(lambda a, b: a + b, o)
    def __radd__(self, o): return self._map(lambda a, b: a + b, o, swap=True)
    def __sub__(self, o): return self._map(lambda a, b: a - b, o)
    def __rsub__(self, o): return self._map(lambda a, b: a - b, o, swap=True)
    def __mul__(self, o): return self._map(lambda a, b: a * b, o)
    def __rmul__(self, o): return self._map(lambda a, b: a * b, o, swap=True)
    def __truediv__(self, o): return self._map(lambda a, b: a / b, o)
    def __rtruediv__(self, o): return self._map(lambda a, b: a / b, o, swap=True)
    def __pow__(self, o): return self._map(lambda a, b: a ** b, o)
    def __rpow__(self, o): return self._map(lambda a, b: a ** b, o, swap=True)
    def __mod__(self, o): return self._map(lambda a, b: a % b, o)
    def __lt__(self, o): return self._map(lambda a, b: a < b, o)
    def __le__(self, o): return self._map(lambda a, b: a <= b, o)
    def __gt__(self, o): return self._map(lambda a, b: a > b, o)
    def __ge__(self, o): return self._map(lambda a, b: a >= b, o)
    __hash__ = None

    def _inplace(self, fn, o):
        others = self._operand_rows(o)
        for row, v in zip(self.rows, others):
            fn(row, v)
        return self

    def __iadd__(self, o): return self._inplace(lambda r, v: r.__iadd__(v), o)
    def __isub__(self, o): return self._inplace(lambda r, v: r.__isub__(v), o)
    def __imul__(self, o): return self._inplace(lambda r, v: r.__imul__(v), o)
    def __itruediv__(self, o): return self._inplace(lambda r, v: r.__itruediv__(v), o)

    def __matmul__(self, o):
        return matmul(self, o)

    def __rmatmul__(self, o):
        return matmul(o, self)

    # -------------------------------------------------------------- reductions
    def _reduce(self, op, axis):
        """NumPy's order: along the last (contiguous) axis a row is reduced like a 1-D vector (pairwise sums); along
        axis 0 the rows are combined one after the other, elementwise; over both axes the C-ordered buffer is one
        1-D reduction."""
        r, c = self.shape
        if axis is None:
            flat = self.ravel()
            return getattr(flat, op)()
        if axis in (1, -1):
            return cat([getattr(Sym(self.g, row.id), op)() for row in self.rows])
        if axis in (0, -2):
            if op in ("sum", "mean"):
                acc = Sym(self.g, self.rows[0].id)
                for row in self.rows[1:]:
                    acc = acc + row
                return acc / float(r) if op == "mean" else acc
            if op == "prod":
                acc = Sym(self.g, self.rows[0].id)
                for row in self.rows[1:]:
                    acc = acc * row
                return acc
            acc = Sym(self.g, self.rows[0].id)
            for row in self.rows[1:]:
                acc = acc._binary(op, row)
            return acc
        raise TraceError("%s(axis=%r) of a 2-D traced array" % (op, axis))

    def sum(self, axis=None): return self._reduce("sum", axis)
    def mean(self, axis=None): return self._reduce("mean", axis)
    def prod(self, axis=None): return self._reduce("prod", axis)
    def min(self, axis=None): return self._reduce("min", axis)
    def max(self, axis=None): return self._reduce("max", axis)


# ----------------------------------------------------------------------------- NumPy constructors while tracing
_ACTIVE_GRAPH = None
_OWN_MODULES = ("numpy", "scipy", __name__.rsplit(".", 1)[0] + ".optimize", __name__, __name__.rsplit(".", 1)[0] + ".codegen")


def _from_user_code(depth=2):
    """Was the patched constructor called by a callback (and not by NumPy / SciPy internals or this package's own
    mirror of the reference classes, which must keep getting plain arrays)?"""
    import sys
    mod = sys._getframe(depth).f_globals.get("__name__", "")
    return not any(mod == own or mod.startswith(own + ".") for own in _OWN_MODULES)


def _has_traced(obj, depth=0):
    if isinstance(obj, (Sym, SymMat)):
        return True
    if isinstance(obj, (list, tuple)) and depth < 3:
        return any(_has_traced(o, depth + 1) for o in obj)
    return False


# One trace at a time, and only the tracing thread sees the patched constructors: ``np.zeros`` & co. are replaced on the
# numpy MODULE for the duration of a trace (there is no narrower hook for ``np.zeros(n)`` inside user code), so a second
# thread that traces, or merely calls ``np.zeros`` from code outside the allow-list, must not be handed a traced buffer
# of somebody else's graph.  ``_TRACE_LOCK`` serialises traces; ``_TRACING_THREAD`` is the one thread whose
# calls are intercepted - every other thread gets NumPy's own function.
_TRACE_LOCK = threading.RLock()
_TRACING_THREAD = None


class tracing_numpy:
    """While the callbacks are traced, the NumPy constructors a callback uses to make its OUTPUT buffer -
    ``np.zeros / ones / empty / full (n)`` and ``((r, n))`` - return traced constants, so that ``out[i] = expression``
    records an assignment instead of asking NumPy to store a traced value in a float array; ``np.array`` /
    ``np.asarray`` of traced pieces build traced vectors / 2-D arrays.  Calls from NumPy's and SciPy's own code and
    from this package's mirror classes are passed through untouched."""

    NAMES = ("zeros", "ones", "empty", "full", "array", "asarray", "asanyarray")

    def __init__(self, graph):
        self.graph = graph

    def __enter__(self):
        global _ACTIVE_GRAPH, _TRACING_THREAD
        _TRACE_LOCK.acquire()
        self._saved_graph, self._saved_thread = _ACTIVE_GRAPH, _TRACING_THREAD
        _ACTIVE_GRAPH = self.graph
        _TRACING_THREAD = threading.get_ident()
        self._orig = {name: getattr(np, name) for name in self.NAMES}
        orig = self._orig

        def mine():
            return _ACTIVE_GRAPH is not None and threading.get_ident() == _TRACING_THREAD

        def make_filled(name, fill_of):
            def ctor(shape, *args, **kwargs):
                g = _ACTIVE_GRAPH
                dtype = kwargs.get("dtype", args[1] if name == "full" and len(args) > 1 else
                                   (args[0] if name != "full" and args else None))
                plain = (g is None or not mine() or not _from_user_code() or set(kwargs) - {"dtype", "fill_value"}
                         or (dtype is not None and np.dtype(dtype) != np.float64))
                fill = fill_of(args, kwargs)
                if not plain and not isinstance(fill, Sym):
                    if isinstance(shape, (numbers.Integral, np.integer)):
                        shape = (int(shape),)
                    if isinstance(shape, (tuple, list)) and all(isinstance(v, (numbers.Integral, np.integer)) for v in shape):
                        if len(shape) == 1 and shape[0] > 0:
                            return Sym(g, g.cvec(np.full(int(shape[0]), float(fill))))
                        if len(shape) == 2 and shape[0] > 0 and shape[1] > 0:
                            return SymMat.filled(g, shape, fill)
                return orig[name](shape, *args, **kwargs)
            ctor.__name__ = name
            return ctor

        def array(obj, *args, **kwargs):
            if mine() and _has_traced(obj):
                dtype = kwargs.get("dtype", args[0] if args else None)
                if dtype is not None and np.dtype(dtype) != np.float64:
                    if isinstance(obj, Sym) and obj._concrete() is not None:
                        return orig["array"](obj._concrete(), *args, **kwargs)      # a constant-only buffer
                    raise TraceError("np.array(..., dtype=%s) of traced values: callbacks are traced in float64" % (dtype,))
                return _array_of(obj, copy=kwargs.get("copy", True) is not False)
            return orig["array"](obj, *args, **kwargs)

        def asarray(obj, *args, **kwargs):
            if mine() and _has_traced(obj):
                return _array_of(obj, copy=False)
            return orig["asarray"](obj, *args, **kwargs)

        def asanyarray(obj, *args, **kwargs):
            if mine() and _has_traced(obj):
                return _array_of(obj, copy=False)
            return orig["asanyarray"](obj, *args, **kwargs)

        np.zeros = make_filled("zeros", lambda a, k: 0.0)
        np.ones = make_filled("ones", lambda a, k: 1.0)
        np.empty = make_filled("empty", lambda a, k: 0.0)
        np.full = make_filled("full", lambda a, k: k["fill_value"] if "fill_value" in k else a[0])
        np.array, np.asarray, np.asanyarray = array, asarray, asanyarray
        return self

    def __exit__(self, *exc):
        global _ACTIVE_GRAPH, _TRACING_THREAD
        for name, fn in self._orig.items():
            setattr(np, name, fn)
        _ACTIVE_GRAPH, _TRACING_THREAD = self._saved_graph, self._saved_thread
        _TRACE_LOCK.release()
        return False


def _array_of(obj, copy=True):
    """``np.array(obj)`` where ``obj`` is, or contains, traced values."""
    if isinstance(obj, (Sym, SymMat)):
        return obj.copy() if copy else obj
    items = list(obj)
    if all((isinstance(it, Sym) and it.length is None) or isinstance(it, (numbers.Real, np.bool_))
           or (isinstance(it, np.ndarray) and it.ndim == 0) for it in items):
        return cat(items)                                      # a list of scalars: a 1-D vector
    return SymMat.vstack([_array_of(it) if isinstance(it, (list, tuple)) and _has_traced(it) else it for it in items],
                         rows_only=True)


class tracing:
    """Everything that is patched while callbacks run on the symbolic decision vector."""

    def __init__(self, graph):
        self._ctx = (intercept_interp1d(), tracing_numpy(graph))

    def __enter__(self):
        for c in self._ctx:
            c.__enter__()
        return self

    def __exit__(self, *exc):
        for c in reversed(self._ctx):
            c.__exit__(*exc)
        return False


def new_decision_vector(n):
    g = Graph()
    return Sym(g, g.add(("p", 0, int(n)), int(n)))
