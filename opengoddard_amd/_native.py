"""ctypes binding of ``libogpsx.so`` (C ABI: ``include/ogpsx.h``).

The library is the product: nothing here falls back to NumPy when it is missing.  It is built
in-tree by ``__graft_entry__.build()`` / :func:`opengoddard_amd.build.build_core`; if the
shared object is absent and ``hipcc`` exists it is compiled on first use, otherwise loading
raises.
"""
from __future__ import annotations

import ctypes as C
import functools
import os

import numpy as np

from . import build as _build

_c_double_p = C.POINTER(C.c_double)
_c_int32_p = C.POINTER(C.c_int32)


class OgDesc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("n", C.c_int32),
                ("m_eq", C.c_int32), ("m_ineq", C.c_int32), ("n_phase", C.c_int32),
                ("nodes", _c_int32_p), ("D", C.POINTER(_c_double_p)), ("cvec", _c_double_p),
                ("n_cvec", C.c_int32), ("module_path", C.c_char_p)]


OG_ABI_VERSION = 1

# every symbol include/ogpsx.h declares: (restype, argtypes)
SIGNATURES = {
    "og_lgl": (C.c_int, [C.c_int32, _c_double_p, _c_double_p, _c_double_p]),
    "og_lgl_dev": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "og_fd_step": (C.c_int, [C.c_int32, _c_double_p, _c_double_p, _c_double_p, _c_double_p]),
    "og_problem_create": (C.c_int, [C.POINTER(OgDesc), C.POINTER(C.c_void_p)]),
    "og_problem_destroy": (None, [C.c_void_p]),
    "og_problem_dims": (C.c_int, [C.c_void_p, _c_int32_p, _c_int32_p, _c_int32_p, _c_int32_p]),
    "og_sweep_mode": (C.c_int, [C.c_void_p]),
    "og_eval": (C.c_int, [C.c_void_p, _c_double_p, _c_double_p]),
    "og_fd_sweep": (C.c_int, [C.c_void_p, _c_double_p, _c_double_p, C.c_int32, C.c_int32,
                              _c_double_p, _c_double_p]),
    "og_eval_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "og_fd_sweep_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "og_fd_columns_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "og_jt_register_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "og_jt_unregister_dev": (C.c_int, [C.c_void_p, C.c_void_p]),
    "og_jt_register_host": (C.c_int, [C.c_void_p, _c_double_p, C.c_int32, C.c_int32]),
    "og_jt_host_path": (C.c_int, [C.c_void_p, _c_double_p, C.POINTER(C.c_int32)]),
    "og_pinned_alloc": (C.c_int, [C.c_int64, C.POINTER(C.c_void_p)]),
    "og_pinned_free": (C.c_int, [C.c_void_p]),
    "og_jt_unregister_host": (C.c_int, [C.c_void_p, _c_double_p]),
    "og_pattern": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                             _c_int32_p]),
    "og_pack_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "og_unpack_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "og_shard_plan": (C.c_int, [C.c_void_p, C.c_int32, _c_int32_p, C.POINTER(C.c_int64)]),
    "og_shard_comm_unique_id": (C.c_int, [C.c_void_p]),
    "og_shard_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "og_shard_comm_destroy": (None, [C.c_void_p]),
    "og_shard_comm_info": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "og_shard_all_gather_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "og_shard_sweep_dev": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "og_shard_pack_dev": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "og_shard_unpack_dev": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "og_comm_init": (C.c_int, [C.c_int32, _c_int32_p]),
    "og_comm_finalize": (None, []),
    "og_comm_size": (C.c_int, []),
    "og_comm_uses_rccl": (C.c_int, []),
    "og_multi_create": (C.c_int, [C.POINTER(OgDesc), C.POINTER(C.c_void_p)]),
    "og_multi_destroy": (None, [C.c_void_p]),
    "og_multi_devices": (C.c_int, [C.c_void_p]),
    "og_multi_eval": (C.c_int, [C.c_void_p, _c_double_p, _c_double_p]),
    "og_multi_fd_sweep": (C.c_int, [C.c_void_p, _c_double_p, _c_double_p, _c_double_p, _c_double_p]),
    "og_multi_fd_sweep_enqueue": (C.c_int, [C.c_void_p, _c_double_p, _c_double_p]),
    "og_multi_synchronize": (C.c_int, [C.c_void_p]),
    "og_multi_replica_dev": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_void_p)]),
    "og_multi_jt_register_host": (C.c_int, [C.c_void_p, _c_double_p]),
    "og_jacobian_exact": (C.c_int, [C.c_void_p, _c_double_p, C.c_int32, C.c_int32, _c_double_p, _c_double_p]),
    "og_jacobian_exact_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "og_device_read": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]),
    "og_trace_read": (C.c_int, [C.c_void_p, _c_double_p, C.c_int64]),
    "og_last_error": (C.c_char_p, []),
    "og_device_count": (C.c_int, []),
    "og_probe_launch": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
}


class NativeError(RuntimeError):
    pass


@functools.lru_cache(maxsize=None)
def lib():
    """Load (building if necessary) libogpsx.so.  Import torch first when it is installed so
    that both share one HIP runtime (same ``libamdhip64.so.7`` SONAME)."""
    try:
        import torch  # noqa: F401  (side effect: its bundled HIP runtime gets loaded first)
    except Exception:
        pass
    path = _build.build_core()          # no-op when the in-tree library matches its sources
    handle = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)
        fn.restype = res
        fn.argtypes = args
    return handle


def check(rc, what):
    if rc != 0:
        msg = lib().og_last_error()
        raise NativeError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))


def dptr(a):
    return a.ctypes.data_as(_c_double_p)


@functools.lru_cache(maxsize=64)
def _lgl_cached(n):
    tau, w, D = np.empty(n), np.empty(n), np.empty((n, n))
    check(lib().og_lgl(n, dptr(tau), dptr(w), dptr(D)), "og_lgl")
    for a in (tau, w, D):
        a.setflags(write=False)
    return tau, w, D


def lgl(n):
    """LGL nodes, weights, differentiation matrix (fresh writable copies)."""
    tau, w, D = _lgl_cached(int(n))
    return tau.copy(), w.copy(), D.copy()


def fd_step(x, lb, ub):
    """SciPy's forward-difference step with bound handling."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    lb = np.ascontiguousarray(lb, dtype=np.float64)
    ub = np.ascontiguousarray(ub, dtype=np.float64)
    h = np.empty_like(x)
    check(lib().og_fd_step(x.shape[0], dptr(x), dptr(lb), dptr(ub), dptr(h)), "og_fd_step")
    return h


def device_count():
    return int(lib().og_device_count())


def pinned_matrix(rows, cols):
    """A ``rows x cols`` float64 array in page-locked host memory from the HIP runtime (``og_pinned_alloc``), or None when
    the runtime refuses.  The memory is freed when the last view of the array is gone (never at interpreter exit: the
    runtime may be gone first).  What it is for: ``og_jt_register_host`` finds such a matrix mapped into the device's
    address space already, in the driver's large fragments - the sweep's writes over PCIe then need a handful of address
    translations instead of one per 4 KB page of a malloc'ed matrix."""
    import weakref
    count = int(rows) * int(cols)
    if count <= 0:
        return None
    ptr = C.c_void_p()
    try:
        if lib().og_pinned_alloc(C.c_int64(8 * count), C.byref(ptr)) != 0 or not ptr.value:
            return None
    except Exception:
        return None
    address = int(ptr.value)
    buf = (C.c_double * count).from_address(address)
    finalizer = weakref.finalize(buf, lib().og_pinned_free, C.c_void_p(address))
    finalizer.atexit = False
    return np.frombuffer(buf, dtype=np.float64).reshape(int(rows), int(cols))
