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


def fd_noise_bound(JT_ref, F_scale, h_cols, rel=1e-9, factor=4.0):
    """Attainable agreement of two forward-difference Jacobians whose residuals differ only in
    rounding: rel*|J| + factor*eps*scale_row/|h_col|."""
    eps = np.finfo(float).eps
    return rel * np.abs(JT_ref) + factor * eps * F_scale[None, :] / np.abs(h_cols)[:, None]


def golden_full_columns(G, k):
    """The full-column capture of a large configuration at evaluation point ``k`` (tools/make_golden.py:
    ``Jfull_*``, CSR over the reference's exact non-zeros) as ``(cols, JT_dense)``, or ``None`` when this
    point has none (small configurations store every column densely in ``JT``)."""
    if "Jfull_points" not in G.files or k not in set(G["Jfull_points"].tolist()):
        return None
    cols = G["Jfull_cols"]
    indptr, indices, data = (G["Jfull_%s_%d" % (key, k)] for key in ("indptr", "indices", "data"))
    JT = np.zeros((cols.size, G["F"].shape[1]))
    rows = np.repeat(np.arange(cols.size), np.diff(indptr))
    JT[rows, indices] = data
    return cols, JT


def assert_zero_pattern(program, cols, JT, JT_ref, what=""):
    """Structural zeros must be exact zeros on both sides; inside the traced dependency pattern
    (``codegen.sparsity``) a forward difference may round to exactly 0 on one side only (the entry is then
    covered by the noise bound, which the caller asserts) - rare: a handful per 1e5 structural non-zeros."""
    from opengoddard_amd import codegen
    cache = getattr(program, "_pattern_cache", None)
    if cache is None:
        indptr, rows = codegen.sparsity(program)
        cache = np.zeros((program.n, program.m), dtype=bool)
        cache[np.repeat(np.arange(program.n), np.diff(indptr)), rows] = True
        program._pattern_cache = cache
    structural = cache[np.asarray(cols)]
    assert not JT[~structural].any(), "non-zero outside the traced dependency pattern %s" % what
    assert not JT_ref[~structural].any(), "the reference has a non-zero outside the traced pattern %s" % what
    differ = (JT != 0) != (JT_ref != 0)
    assert differ.sum() <= 2 + 2e-4 * structural.sum(), "zero pattern differs in %d entries %s" % (differ.sum(), what)


def record_measurement(test, **values):
    """Append what a GPU test measured to gpurun_out/test_measurements.jsonl (scratch that gpurun merges back): the
    numbers behind the bounds the tests assert - so that a bound can be set from a measurement, not from a guess."""
    import json
    folder = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(folder, exist_ok=True)
        with open(os.path.join(folder, "test_measurements.jsonl"), "a") as fh:
            fh.write(json.dumps(dict(values, test=test)) + "\n")
    except OSError:
        pass
