"""HIP engine: the GPU evaluation of one Problem's NLP callbacks behind ``Problem.solve``.

Construction traces the callbacks (``codegen.trace_problem``), emits the device header,
compiles the sweep kernels against it (``build.build_module``; cached by source hash) and
creates a handle in ``libogpsx.so`` (``og_problem_create``, include/ogpsx.h).  Afterwards

* :meth:`values` = one ``og_eval``: ``(cost, c_eq, c_ineq)`` at ``p``  - replaces one call each
  of the reference's ``cost_add`` / ``equality_add`` / ``inequality`` (``optimize.py:670-728``);
* :meth:`jacobians` = one ``og_fd_sweep`` over all n columns - replaces the 3n+2 Python
  callback evaluations SciPy's ``approx_derivative`` makes per SLSQP major iteration
  (``scipy/optimize/_slsqp_py.py:299-313``, ``_numdiff.py:584-625``).

There is deliberately no NumPy fallback: no GPU or no compiled extension => exception.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _native, build, codegen


class HipEngine:
    jacobian_mode = "fd"        # "fd": SciPy's forward differences (the reference); "exact": forward-mode AD

    def __init__(self, prob, obj, device=0, program=None, devices=None):
        """``devices=[d0, d1, ...]`` (more than one): every full sweep of :meth:`jacobians` is column-sharded
        over these GPUs from this one process (``og_comm_init`` + ``og_multi_fd_sweep``); ``device`` is then the
        first of them and keeps serving the single evaluations and the device-pointer entry points."""
        lib = _native.lib()
        if devices is not None:
            devices = [int(d) for d in devices]
            device = devices[0]
        if _native.device_count() < 1:
            raise RuntimeError("opengoddard_amd: no HIP device is visible; the MI355X engine has "
                               "no CPU fallback")
        self.program = program or codegen.trace_problem(prob, obj)
        self.header = codegen.emit_header(self.program)
        self.digest = codegen.program_hash(self.header)
        self.module_path = build.build_module(self.header, self.digest)
        P = self.program
        self.n, self.m, self.m_eq, self.m_ineq = P.n, P.m, P.m_eq, P.m_ineq
        self.device = int(device)

        self._nodes = (C.c_int32 * len(P.nodes))(*P.nodes)
        # a phase whose D is still the library's own LGL matrix is left NULL: the runtime then
        # builds it with the device LGL kernel; a user-modified D is uploaded as given
        self._D = [np.ascontiguousarray(D, dtype=np.float64) for D in prob.D]
        dp = C.POINTER(C.c_double)
        self._Dptr = (dp * len(self._D))(*[
            None if (d.shape == (n, n) and np.array_equal(d, _native.lgl(n)[2]))
            else d.ctypes.data_as(dp) for d, n in zip(self._D, P.nodes)])
        self._cvec = np.ascontiguousarray(P.cvec, dtype=np.float64)
        desc = _native.OgDesc(
            abi_version=_native.OG_ABI_VERSION, device=self.device, n=P.n, m_eq=P.m_eq,
            m_ineq=P.m_ineq, n_phase=len(P.nodes), nodes=self._nodes, D=self._Dptr,
            cvec=self._cvec.ctypes.data_as(dp) if self._cvec.size else None,
            n_cvec=int(self._cvec.size), module_path=self.module_path.encode())
        self._lib = lib
        self._handle = C.c_void_p()
        _native.check(lib.og_problem_create(C.byref(desc), C.byref(self._handle)),
                      "og_problem_create")
        self._val_key = self._val = None
        self._jac_key = self._jac = None
        self.n_values = self.n_sweeps = 0
        # the matrix SciPy's callbacks read: allocated once, registered as persistent-zero, so that a sweep
        # moves and writes the non-zeros only (og_jt_register_host)
        self._JT_host = None
        self.devices = devices if devices and len(devices) > 1 else None
        self._multi = C.c_void_p()
        if self.devices:
            arr = (C.c_int32 * len(self.devices))(*self.devices)
            _native.check(lib.og_comm_init(len(self.devices), arr), "og_comm_init")
            _native.check(lib.og_multi_create(C.byref(desc), C.byref(self._multi)), "og_multi_create")

    # ------------------------------------------------------------------ lifetime
    def close(self):
        cache = getattr(self, "_sqp_cache", None)
        if cache is not None:                       # QP work space of the SQP driver (sqp.py)
            cache[1].close()
            self._sqp_cache = None
        if getattr(self, "_multi", None) is not None and self._multi.value:
            self._lib.og_multi_destroy(self._multi)
            self._multi = C.c_void_p()
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.og_problem_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ raw calls
    def eval_stacked(self, x):
        """F(x) = [cost | c_eq | c_ineq] (host in, host out)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.shape == (self.n,)
        F = np.empty(self.m)
        _native.check(self._lib.og_eval(self._handle, _native.dptr(x), _native.dptr(F)), "og_eval")
        return F

    def sweep_stacked(self, x, h, col_lo=0, col_hi=None):
        """(F0, JT) with JT[(j - col_lo), r] = dF_r/dx_j for the columns in [col_lo, col_hi)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        h = np.ascontiguousarray(h, dtype=np.float64)
        col_hi = self.n if col_hi is None else int(col_hi)
        F0 = np.empty(self.m)
        JT = np.empty((col_hi - col_lo, self.m))
        _native.check(self._lib.og_fd_sweep(self._handle, _native.dptr(x), _native.dptr(h),
                                            int(col_lo), col_hi, _native.dptr(JT),
                                            _native.dptr(F0)), "og_fd_sweep")
        return F0, JT

    def exact_stacked(self, x, col_lo=0, col_hi=None):
        """(F0, JT) with JT[(j - col_lo), r] = dF_r/dx_j by forward-mode differentiation on the GPU."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        col_hi = self.n if col_hi is None else int(col_hi)
        F0 = np.empty(self.m)
        JT = np.empty((col_hi - col_lo, self.m))
        _native.check(self._lib.og_jacobian_exact(self._handle, _native.dptr(x), int(col_lo), col_hi,
                                                  _native.dptr(JT), _native.dptr(F0)), "og_jacobian_exact")
        return F0, JT

    def pattern(self, col_lo=0, col_hi=None):
        """Static pattern of J_T for the columns [col_lo, col_hi): ``(indptr, rows)`` (``og_pattern``)."""
        col_hi = self.n if col_hi is None else int(col_hi)
        nnz = C.c_int64()
        _native.check(self._lib.og_pattern(self._handle, int(col_lo), col_hi, C.byref(nnz), None, None), "og_pattern")
        indptr = np.empty(col_hi - col_lo + 1, dtype=np.int64)
        rows = np.empty(nnz.value, dtype=np.int32)
        _native.check(self._lib.og_pattern(self._handle, int(col_lo), col_hi, None,
                                           indptr.ctypes.data_as(C.POINTER(C.c_int64)),
                                           rows.ctypes.data_as(C.POINTER(C.c_int32))), "og_pattern")
        return indptr, rows

    def sweep_persistent(self, x, h, exact=False):
        """``(F0, JT)`` over all columns like :meth:`sweep_stacked`, but ``JT`` is the engine's ONE persistent
        host matrix (registered with ``og_jt_register_host``: only the packed non-zeros cross PCIe and are
        scattered into it).  The caller must be done with the previous result - SciPy's SLSQP is: it copies the
        Jacobians it is handed.  With ``devices=[...]`` the sweep is column-sharded over those GPUs."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        F0 = np.empty(self.m)
        if self._JT_host is None:
            # (page-locked memory of the runtime's own where it gives some: the sweep writes this matrix over PCIe)
            pinned = None if os.environ.get("OGPSX_HOST") == "staged" else _native.pinned_matrix(self.n, self.m)
            self._JT_host = pinned if pinned is not None else np.empty((self.n, self.m))
            if self._multi.value:
                _native.check(self._lib.og_multi_jt_register_host(self._multi, _native.dptr(self._JT_host)),
                              "og_multi_jt_register_host")
            _native.check(self._lib.og_jt_register_host(self._handle, _native.dptr(self._JT_host), 0, self.n),
                          "og_jt_register_host")
        JT = self._JT_host
        if exact:
            _native.check(self._lib.og_jacobian_exact(self._handle, _native.dptr(x), 0, self.n, _native.dptr(JT),
                                                      _native.dptr(F0)), "og_jacobian_exact")
        elif self._multi.value:
            h = np.ascontiguousarray(h, dtype=np.float64)
            _native.check(self._lib.og_multi_fd_sweep(self._multi, _native.dptr(x), _native.dptr(h), _native.dptr(JT),
                                                      _native.dptr(F0)), "og_multi_fd_sweep")
        else:
            h = np.ascontiguousarray(h, dtype=np.float64)
            _native.check(self._lib.og_fd_sweep(self._handle, _native.dptr(x), _native.dptr(h), 0, self.n,
                                                _native.dptr(JT), _native.dptr(F0)), "og_fd_sweep")
        return F0, JT

    @property
    def host_path(self):
        """How ``sweep_persistent`` reaches the persistent host matrix (``og_jt_host_path``): "mapped" (the launch writes
        it over PCIe), "staged" (packed copy + host scatter), "undecided" (the first six sweeps time both), None before
        the first call."""
        if self._JT_host is None:
            return None
        path = C.c_int32(-1)
        _native.check(self._lib.og_jt_host_path(self._handle, _native.dptr(self._JT_host), C.byref(path)), "og_jt_host_path")
        return {1: "mapped", 2: "staged", 0: "undecided"}.get(path.value)

    @property
    def sweep_mode(self):
        """How ``sweep_dev`` runs (``og_sweep_mode``): "fused" (one launch), "split" or "dense"."""
        return {5: "fused", 1: "split", 2: "dense"}[self._lib.og_sweep_mode(self._handle)]

    def exact_dev(self, d_x, col_lo, col_hi, d_JT, d_F0, stream=0):
        _native.check(self._lib.og_jacobian_exact_dev(self._handle, d_x, int(col_lo), int(col_hi), d_JT, d_F0,
                                                      stream), "og_jacobian_exact_dev")

    def eval_dev(self, d_x, d_F, stream=0):
        _native.check(self._lib.og_eval_dev(self._handle, d_x, d_F, stream), "og_eval_dev")

    def sweep_dev(self, d_x, d_h, col_lo, col_hi, d_JT, d_F0, stream=0):
        """Asynchronous device-pointer sweep (ints from ``torch.Tensor.data_ptr()``)."""
        _native.check(self._lib.og_fd_sweep_dev(self._handle, d_x, d_h, int(col_lo), int(col_hi),
                                                d_JT, d_F0, stream), "og_fd_sweep_dev")

    def register_jt_dev(self, d_JT, col_lo, col_hi, stream=0):
        """Declare ``d_JT`` a persistent-zero buffer for the columns [col_lo, col_hi): it is zero-filled now
        and later sweeps into it write the non-zeros only (``og_jt_register_dev``, include/ogpsx.h)."""
        _native.check(self._lib.og_jt_register_dev(self._handle, d_JT, int(col_lo), int(col_hi), stream),
                      "og_jt_register_dev")

    def unregister_jt_dev(self, d_JT):
        _native.check(self._lib.og_jt_unregister_dev(self._handle, d_JT), "og_jt_unregister_dev")

    def columns_dev(self, d_x, d_h, col_lo, col_hi, d_JT, d_F0, stream=0):
        """The sweep kernel alone (``d_F0`` must already hold F(x) from :meth:`eval_dev`)."""
        _native.check(self._lib.og_fd_columns_dev(self._handle, d_x, d_h, int(col_lo), int(col_hi),
                                                  d_JT, d_F0, stream), "og_fd_columns_dev")

    # ------------------------------------------------------------------ SciPy-facing
    def _split(self, F):
        return F[0], F[1:1 + self.m_eq], F[1 + self.m_eq:]

    def values(self, p):
        key = np.asarray(p, dtype=np.float64).tobytes()
        if key != self._val_key:
            self._val = self._split(self.eval_stacked(p))
            self._val_key = key
            self.n_values += 1
        return self._val

    def jacobians(self, p, lb, ub):
        """((grad, J_eq, J_ineq), h) at ``p``; one sweep serves all three SLSQP requests.  ``J_eq`` and ``J_ineq``
        are VIEWS of the engine's one persistent host matrix: they are valid until the next call with another
        ``p`` (SciPy's SLSQP copies what it is handed; any other consumer that keeps a result across calls must
        copy it).  ``grad`` is a copy."""
        key = np.asarray(p, dtype=np.float64).tobytes()
        if key != self._jac_key:
            h = _native.fd_step(p, lb, ub)           # (also in exact mode: quirk Q13 needs the last step)
            F0, JT = self.sweep_persistent(p, h, exact=self.jacobian_mode == "exact")
            J = JT.T                                 # views of the persistent matrix (SciPy copies them)
            self._jac = ((np.array(J[0], dtype=np.float64, copy=True), J[1:1 + self.m_eq], J[1 + self.m_eq:]), h)
            self._jac_key = key
            self._val, self._val_key = self._split(F0), key
            self.n_sweeps += 1
        return self._jac
