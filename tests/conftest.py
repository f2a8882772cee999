import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def has_gpu():
    try:
        from opengoddard_amd import _native
        return _native.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    gpu = has_gpu()
    for item in items:
        if "gpu" in item.keywords and not gpu:
            item.add_marker(pytest.mark.skip(reason="no HIP device"))
        if "reference" in item.keywords and not os.path.isdir(REFERENCE):
            item.add_marker(pytest.mark.skip(reason="/root/reference not present"))


@pytest.fixture(scope="session", autouse=True)
def _native_library():
    """Every test needs libogpsx.so (LGL lives there); build it once if it is not in-tree."""
    from opengoddard_amd import build
    build.build_core()


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"))
        return cache[name]
    return load


@pytest.fixture(scope="session")
def lgl_golden():
    return np.load(os.path.join(GOLDEN, "lgl.npz"))


def inject_reference_lgl(prob, lgl):
    """Replace a Problem's LGL data by the reference's own (tests/golden/lgl.npz) so that NumPy
    restatements can be compared with the goldens bit for bit."""
    for i, n in enumerate(prob.nodes):
        prob.tau[i] = lgl["tau_%d" % n].copy()
        prob.w[i] = lgl["w_%d" % n].copy()
        prob.D[i] = lgl["D_%d" % n].copy()


def fd_noise_bound(JT_ref, F_scale, h_cols, rel=1e-9, factor=64.0):
    """Attainable agreement of two forward-difference Jacobians whose residuals differ only in
    rounding: rel*|J| + factor*eps*scale_row/|h_col|."""
    eps = np.finfo(float).eps
    return rel * np.abs(JT_ref) + factor * eps * F_scale[None, :] / np.abs(h_cols)[:, None]
