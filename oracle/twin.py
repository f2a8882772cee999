"""TEST INFRASTRUCTURE - build and call the CPU twin (``oracle/twin.cpp``) of a traced problem.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use this.
The shared objects go to ``oracle/_build/`` (git-ignored, shipped to the GPU box by gpurun).
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")
CSRC = os.path.join(os.path.dirname(HERE), "opengoddard_amd", "csrc")
CXX_FLAGS = ["-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-fPIC", "-shared",
             "-Wno-unused-value"]

_dp = C.POINTER(C.c_double)


def build_twin(header_source, digest=None, force=False):
    os.makedirs(BUILD, exist_ok=True)
    src = os.path.join(HERE, "twin.cpp")
    hsh = hashlib.sha256(header_source.encode())
    for path in (src, os.path.join(CSRC, "og_math.h"), os.path.join(CSRC, "og_dual.h")):
        with open(path, "rb") as fh:
            hsh.update(fh.read())
    hsh.update(" ".join(CXX_FLAGS).encode())
    digest = hsh.hexdigest()[:16]
    out = os.path.join(BUILD, "twin_%s.so" % digest)
    if not force and os.path.exists(out):
        return out
    header = os.path.join(BUILD, "og_gen_%s.h" % digest)
    with open(header + ".tmp%d" % os.getpid(), "w") as fh:
        fh.write(header_source)
    os.replace(fh.name, header)
    tmp = out + ".tmp%d" % os.getpid()
    cmd = ["g++"] + CXX_FLAGS + ["-I" + CSRC, "-DOG_GEN_HEADER=\"%s\"" % header, src, "-o", tmp]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("twin build failed:\n" + proc.stdout[-4000:])
    os.replace(tmp, out)
    return out


class Twin:
    """CPU twin of one traced problem: ``values(x)`` and ``sweep(x, h, cols)``."""

    def __init__(self, prob, obj, program=None, header=None):
        from opengoddard_amd import codegen
        self.program = program or codegen.trace_problem(prob, obj)
        self.header = header or codegen.emit_header(self.program)
        self.lib = C.CDLL(build_twin(self.header, codegen.program_hash(self.header)))
        dims = [C.c_int() for _ in range(4)]
        self.lib.twin_dims(*[C.byref(d) for d in dims])
        self.n, self.m, self.m_eq, self.m_ineq = (d.value for d in dims)
        self._D = [np.ascontiguousarray(D, dtype=np.float64) for D in prob.D]
        self._Dptr = (_dp * len(self._D))(*[d.ctypes.data_as(_dp) for d in self._D])
        self._cv = np.ascontiguousarray(self.program.cvec, dtype=np.float64)
        if self._cv.size == 0:
            self._cv = np.zeros(1)

    def values(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        F = np.empty(self.m)
        self.lib.twin_eval(x.ctypes.data_as(_dp), self._Dptr, self._cv.ctypes.data_as(_dp),
                           F.ctypes.data_as(_dp))
        return F

    def sweep(self, x, h, cols=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        h = np.ascontiguousarray(h, dtype=np.float64)
        cols = np.arange(self.n, dtype=np.int32) if cols is None else \
            np.ascontiguousarray(cols, dtype=np.int32)
        F0 = np.empty(self.m)
        JT = np.empty((cols.size, self.m))
        self.lib.twin_sweep(x.ctypes.data_as(_dp), h.ctypes.data_as(_dp), self._Dptr,
                            self._cv.ctypes.data_as(_dp), cols.ctypes.data_as(C.POINTER(C.c_int)),
                            int(cols.size), F0.ctypes.data_as(_dp), JT.ctypes.data_as(_dp))
        return F0, JT

    def exact(self, x, cols=None):
        """(F0, JT) with JT[r] = dF/dx_cols[r]: forward-mode derivatives of the generated code."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        cols = np.arange(self.n, dtype=np.int32) if cols is None else \
            np.ascontiguousarray(cols, dtype=np.int32)
        F0 = np.empty(self.m)
        JT = np.empty((cols.size, self.m))
        self.lib.twin_exact(x.ctypes.data_as(_dp), self._Dptr, self._cv.ctypes.data_as(_dp),
                            cols.ctypes.data_as(C.POINTER(C.c_int)), int(cols.size),
                            F0.ctypes.data_as(_dp), JT.ctypes.data_as(_dp))
        return F0, JT
