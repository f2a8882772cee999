"""csrc/og_math.h on the host (g++): accuracy against NumPy and special values.  The same source
is compiled for gfx950; bit-equality host<->device is checked on the GPU by tools/gpu_probe.hip
(tests/test_gpu_parity.py::test_hardware_probe)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

SRC = r"""
#include "og_math.h"
extern "C" {
void v_exp(const double* x, double* y, int n) { for (int i = 0; i < n; ++i) y[i] = ogm::exp_(x[i]); }
void v_log(const double* x, double* y, int n) { for (int i = 0; i < n; ++i) y[i] = ogm::log_(x[i]); }
void v_sin(const double* x, double* y, int n) { for (int i = 0; i < n; ++i) y[i] = ogm::sin_(x[i]); }
void v_cos(const double* x, double* y, int n) { for (int i = 0; i < n; ++i) y[i] = ogm::cos_(x[i]); }
void v_tan(const double* x, double* y, int n) { for (int i = 0; i < n; ++i) y[i] = ogm::tan_(x[i]); }
}
"""


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("ogmath")
    src = d / "m.cpp"
    src.write_text(SRC)
    so = d / "m.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-fPIC", "-shared",
                           "-I" + os.path.join(ROOT, "opengoddard_amd", "csrc"), str(src), "-o", str(so)])
    return C.CDLL(str(so))


def call(lib, name, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    dp = C.POINTER(C.c_double)
    getattr(lib, name)(x.ctypes.data_as(dp), y.ctypes.data_as(dp), x.size)
    return y


def ulps(a, b):
    return np.abs(a - b) / np.spacing(np.abs(b))


@pytest.mark.parametrize("name,ref,sample,tol", [
    ("v_exp", np.exp, lambda r: r.uniform(-700, 700, 200000), 1.0),
    ("v_exp", np.exp, lambda r: r.uniform(-1, 1, 200000) * 10.0 ** r.integers(-12, 1, 200000), 1.0),
    ("v_log", np.log, lambda r: np.exp(r.uniform(-700, 700, 200000)), 1.0),
    ("v_sin", np.sin, lambda r: r.uniform(-100, 100, 200000), 1.0),
    ("v_cos", np.cos, lambda r: r.uniform(-100, 100, 200000), 1.0),
    ("v_sin", np.sin, lambda r: r.uniform(-1e5, 1e5, 200000), 1.0),
    ("v_cos", np.cos, lambda r: r.uniform(-1e5, 1e5, 200000), 1.0),
    ("v_tan", np.tan, lambda r: r.uniform(-1.5, 1.5, 200000), 2.0),
])
def test_within_one_ulp_of_numpy(lib, name, ref, sample, tol):
    x = sample(np.random.default_rng(0))
    got, want = call(lib, name, x), ref(x)
    assert np.max(ulps(got, want)) <= tol


def test_special_values(lib):
    inf, nan = np.inf, np.nan
    e = call(lib, "v_exp", [0.0, -0.0, inf, -inf, nan, 710.0, -746.0, 1e-300, 709.78, -745.0])
    assert e[0] == 1.0 and e[1] == 1.0 and e[2] == inf and e[3] == 0.0 and np.isnan(e[4])
    assert e[5] == inf and e[6] == 0.0 and e[7] == 1.0
    assert ulps(e[8:], np.exp([709.78, -745.0])).max() <= 1.0          # near overflow / subnormal
    lg = call(lib, "v_log", [1.0, 0.0, -0.0, -1.0, inf, nan, 5e-324, 2.2250738585072014e-308])
    assert lg[0] == 0.0 and lg[1] == -inf and lg[2] == -inf and np.isnan(lg[3]) and lg[4] == inf
    assert np.isnan(lg[5])
    assert ulps(lg[6:], np.log([5e-324, 2.2250738585072014e-308])).max() <= 1.0
    for name in ("v_sin", "v_cos", "v_tan"):
        v = call(lib, name, [inf, -inf, nan])
        assert np.isnan(v).all()
    s = call(lib, "v_sin", [0.0, -0.0, 1e-300])
    assert s[0] == 0.0 and np.signbit(s[1]) and s[2] == 1e-300
    assert call(lib, "v_cos", [0.0, 1e-300]).tolist() == [1.0, 1.0]
