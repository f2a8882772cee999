"""ctypes binding of ``libogsqp.so`` (C ABI: ``include/ogsqp.h``) - the QP subproblem and the
quasi-Newton update of the SQP driver.  No CPU fallback: loading or creating a handle without a
HIP device raises."""
from __future__ import annotations

import ctypes as C
import functools

import numpy as np

from . import build as _build

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)

OGSQP_ABI_VERSION = 1
QP_SOLVED, QP_TOO_MANY_EQ, QP_ITERATION_LIMIT, QP_INCOMPATIBLE, QP_SINGULAR_C = 1, 2, 3, 4, 6

# every symbol include/ogsqp.h declares: (restype, argtypes)
SIGNATURES = {
    "og_qp_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "og_qp_destroy": (None, [C.c_void_p]),
    "og_qp_reset": (C.c_int, [C.c_void_p]),
    "og_qp_get_factor": (C.c_int, [C.c_void_p, _dp]),
    "og_qp_set_factor": (C.c_int, [C.c_void_p, _dp]),
    "og_qp_solve_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, _dp, _dp, _dp, _dp, C.c_int32,
                                  C.c_double, _dp, _dp, _dp, _ip, _ip, C.c_void_p]),
    "og_qp_solve": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _dp, C.c_int32, C.c_double, _dp, _dp,
                              _dp, _ip, _ip]),
    "og_qp_get_active": (C.c_int, [C.c_void_p, _ip, C.c_int32, _ip]),
    "og_qp_set_active": (C.c_int, [C.c_void_p, _ip, C.c_int32]),
    "og_qp_recoveries": (C.c_int, [C.c_void_p, _ip]),
    "og_qp_resident_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "og_qp_bfgs": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _ip]),
    "og_jt_times": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, _dp, _dp, C.c_void_p]),
    "og_qp_last_error": (C.c_char_p, []),
}


class SqpNativeError(RuntimeError):
    pass


@functools.lru_cache(maxsize=None)
def lib():
    try:
        import torch  # noqa: F401  (share one HIP runtime with torch when it is installed)
    except Exception:
        pass
    handle = C.CDLL(_build.build_sqp(), mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)
        fn.restype = res
        fn.argtypes = args
    return handle


def check(rc, what):
    if rc != 0:
        msg = lib().og_qp_last_error()
        raise SqpNativeError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))


def _p(a):
    return a.ctypes.data_as(_dp)


def _vec(a, size=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if size is not None and a.shape != (size,):
        raise ValueError("expected a vector of %d entries, got shape %r" % (size, a.shape))
    return a


class QpCore:
    """One handle: the factor ``Z`` (``B^-1 = Z Z'``) and the QP work space on one device."""

    def __init__(self, n, m_eq, m_ineq, device=0):
        self.n, self.m_eq, self.m_ineq, self.m = int(n), int(m_eq), int(m_ineq), int(m_eq + m_ineq)
        self._lib = lib()
        self._handle = C.c_void_p()
        check(self._lib.og_qp_create(OGSQP_ABI_VERSION, int(device), self.n, self.m_eq, self.m_ineq,
                                     C.byref(self._handle)), "og_qp_create")

    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.og_qp_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        check(self._lib.og_qp_reset(self._handle), "og_qp_reset")

    def get_factor(self):
        Z = np.empty((self.n, self.n))
        check(self._lib.og_qp_get_factor(self._handle, _p(Z)), "og_qp_get_factor")
        return Z

    def set_factor(self, Z):
        Z = np.ascontiguousarray(Z, dtype=np.float64)
        if Z.shape != (self.n, self.n):
            raise ValueError("factor must be %d x %d" % (self.n, self.n))
        check(self._lib.og_qp_set_factor(self._handle, _p(Z)), "og_qp_set_factor")

    def _outputs(self, augmented):
        nq = self.n + (1 if augmented else 0)
        return np.zeros(nq), np.zeros(max(self.m, 1)), np.zeros(nq), C.c_int32(0), C.c_int32(0)

    def solve_dev(self, d_jt, ld, g, c, dl, du, augmented=False, rho=100.0, stream=0):
        """QP on the device-resident transposed Jacobian -> ``(d, mult, bound_mult, status, iterations)``."""
        nq = self.n + (1 if augmented else 0)
        g, c = _vec(g, self.n), _vec(c, self.m)
        dl, du = _vec(dl, nq), _vec(du, nq)
        d, mult, bm, status, iters = self._outputs(augmented)
        check(self._lib.og_qp_solve_dev(self._handle, d_jt, int(ld), _p(g), _p(c), _p(dl), _p(du),
                                        1 if augmented else 0, float(rho), _p(d), _p(mult), _p(bm),
                                        C.byref(status), C.byref(iters), stream), "og_qp_solve_dev")
        return d, mult[:self.m], bm, status.value, iters.value

    def solve(self, A, g, c, dl, du, augmented=False, rho=100.0):
        """Same with the Jacobian on the host (``m x n``, row-major)."""
        nq = self.n + (1 if augmented else 0)
        A = np.ascontiguousarray(A, dtype=np.float64).reshape(self.m, self.n)
        g, c = _vec(g, self.n), _vec(c, self.m)
        dl, du = _vec(dl, nq), _vec(du, nq)
        d, mult, bm, status, iters = self._outputs(augmented)
        check(self._lib.og_qp_solve(self._handle, _p(A), _p(g), _p(c), _p(dl), _p(du),
                                    1 if augmented else 0, float(rho), _p(d), _p(mult), _p(bm),
                                    C.byref(status), C.byref(iters)), "og_qp_solve")
        return d, mult[:self.m], bm, status.value, iters.value

    def get_active(self):
        """Rows active at the last solution (``og_qp_get_active`` numbering): where the next solve starts."""
        count = C.c_int32(0)
        check(self._lib.og_qp_get_active(self._handle, None, 0, C.byref(count)), "og_qp_get_active")
        ids = np.zeros(max(count.value, 1), dtype=np.int32)
        check(self._lib.og_qp_get_active(self._handle, ids.ctypes.data_as(_ip), count.value, C.byref(count)),
              "og_qp_get_active")
        return ids[:count.value].copy()

    def set_active(self, ids=()):
        """Replace the warm-start rows; ``set_active()`` makes the next solve start from the empty active set."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        check(self._lib.og_qp_set_active(self._handle, ids.ctypes.data_as(_ip), int(ids.size)), "og_qp_set_active")

    def recoveries(self):
        """Subproblems that were re-run with the separate-launch kernels after an inter-workgroup wait gave up."""
        count = C.c_int32(0)
        check(self._lib.og_qp_recoveries(self._handle, C.byref(count)), "og_qp_recoveries")
        return count.value

    def resident_stats(self):
        """(launches, active-set changes) of the one-launch active-set loop (``k_rows_resident``); (0, 0) when the
        two-launch form serves this handle."""
        launches, changes = C.c_int64(0), C.c_int64(0)
        check(self._lib.og_qp_resident_stats(self._handle, C.byref(launches), C.byref(changes)), "og_qp_resident_stats")
        return int(launches.value), int(changes.value)

    def bfgs(self, s, eta, Bs):
        """Damped BFGS on the factor; returns True when the caller has to reset instead."""
        s, eta, Bs = _vec(s, self.n), _vec(eta, self.n), _vec(Bs, self.n)
        flag = C.c_int32(0)
        check(self._lib.og_qp_bfgs(self._handle, _p(s), _p(eta), _p(Bs), C.byref(flag)), "og_qp_bfgs")
        return bool(flag.value)

    def jt_times(self, d_jt, ld, coef, stream=0):
        coef = _vec(coef, self.m + 1)
        out = np.empty(self.n)
        check(self._lib.og_jt_times(self._handle, d_jt, int(ld), _p(coef), _p(out), stream), "og_jt_times")
        return out
