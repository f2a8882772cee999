"""GPU parity tests (``-m gpu``): the HIP sweep, called through the C ABI, against

1. the CPU twin (``oracle/twin.cpp``) - same arithmetic order, so the bar is **bit-exact**
   (values, Jacobian entries, zero pattern), at the reference's golden evaluation points;
2. the reference's own goldens (``tests/golden/cfg_*.npz``, made by the reference engine +
   SciPy 1.15.3) - residuals within 1e-9 relative to the row's term magnitude, Jacobian within
   the forward-difference noise bound of SURVEY.md section 8(c):
   ``1e-9*|J| + 4*eps*rowscale/|h|`` (``conftest.fd_noise_bound``, factor 4; the reference's own Jacobian
   moves by this much when its BLAS merely sums in a different order).
"""
import numpy as np
import pytest

from conftest import assert_zero_pattern, fd_noise_bound, golden_full_columns, inject_reference_lgl
from opengoddard_amd import _native, problems

pytestmark = pytest.mark.gpu

SMALL = ("brachistochrone", "goddard", "polar_tsto_shipped", "low_thrust_shipped")
ALL = problems.NAMES


def _engine_and_twin(name, lgl=None):
    from opengoddard_amd.engine import HipEngine
    from oracle import twin
    prob, obj = problems.build(name)
    if lgl is not None:
        inject_reference_lgl(prob, lgl)
    eng = HipEngine(prob, obj)
    tw = twin.Twin(prob, obj, program=eng.program, header=eng.header)
    return prob, obj, eng, tw


def row_scales(eng, prob, x, F):
    """Magnitude of the terms summed in each row (defect rows: sum_l |D_kl| |x~_l|)."""
    scale = np.maximum(1.0, np.abs(F))
    P = eng.program
    for g in P.groups:
        if g.kind != "defect":
            continue
        D = np.abs(prob.D[g.phase])
        for (row, _), slot in zip(g.outputs, g.mv_slots):
            leaf = P.mv[slot].leaf_base
            xs = np.abs(x[leaf:leaf + g.length])
            scale[row:row + g.length] = np.maximum(scale[row:row + g.length], D.dot(xs))
    return scale


@pytest.mark.parametrize("name", ALL)
def test_bit_exact_against_cpu_twin(name, golden):
    G = golden("cfg_" + name)
    prob, obj, eng, tw = _engine_and_twin(name)
    from oracle import np_path
    lb, ub = np_path.bounds_arrays(prob)
    for k in range(G["x"].shape[0]):
        x = G["x"][k]
        F_gpu = eng.eval_stacked(x)
        assert np.array_equal(F_gpu, tw.values(x)), "F differs at point %d" % k
        h = _native.fd_step(x, lb, ub)
        assert np.array_equal(h, G["h"][k]), "FD step rule differs from SciPy's at point %d" % k
        cols = np.arange(eng.n) if eng.n <= 800 else G["cols"]
        F0, JT = eng.sweep_stacked(x, h)
        assert np.array_equal(F0, F_gpu)
        F0c, JTc = tw.sweep(x, h, cols)
        assert np.array_equal(JT[cols], JTc), "Jacobian differs from the CPU twin at point %d" % k
        assert np.isfinite(JT).all()
    eng.close()


@pytest.mark.parametrize("name", ALL)
def test_against_reference_goldens(name, golden, lgl_golden):
    G = golden("cfg_" + name)
    prob, obj, eng, tw = _engine_and_twin(name, lgl_golden)
    cols = G["cols"]
    for k in range(G["x"].shape[0]):
        x, Fg, h, JTg = G["x"][k], G["F"][k], G["h"][k], G["JT"][k]
        F0, JT = eng.sweep_stacked(x, h)
        scale = row_scales(eng, prob, x, Fg)
        assert np.all(np.abs(F0 - Fg) <= 1e-9 * scale), \
            "residual off by %.3g" % np.max(np.abs(F0 - Fg) / scale)
        # all-elementwise rows without transcendentals round identically on GPU and in NumPy
        bound = fd_noise_bound(JTg, scale, h[cols])
        err = np.abs(JT[cols] - JTg)
        assert np.all(err <= bound), "Jacobian outside the FD noise bound: worst ratio %.3g" % \
            np.max(err / np.maximum(bound, 1e-300))
        # structural zeros are exact zeros on both sides (same code for base and perturbed values)
        assert np.array_equal(JT[cols] != 0, JTg != 0), "zero pattern differs from the reference's"
        full = golden_full_columns(G, k)
        if full is not None:            # every column (C3, C4) / 1024 columns (C5) the reference differenced
            fcols, JTf = full
            err, bound = np.abs(JT[fcols] - JTf), fd_noise_bound(JTf, scale, h[fcols])
            assert np.all(err <= bound), "full-column golden: worst ratio %.3g" % np.max(err / np.maximum(bound, 1e-300))
            assert_zero_pattern(eng.program, fcols, JT[fcols], JTf)
    eng.close()


@pytest.mark.parametrize("name", ALL)
def test_against_numpy_restatement_without_generated_code(name, golden):
    """An oracle link that shares NOTHING with the code under test: the GPU results against (a) the NumPy
    restatement of the reference path (oracle/np_path.py: the user's callbacks run by NumPy, SciPy's column
    loop - no tracer, no generated code, no og_math.h) for EVERY column of every configuration (C5: 6148 columns, 3.5 s of NumPy), and (b) the NumPy interpreter of the traced program (oracle/program_eval.py).  A tracer, lowering,
    code-emission or og_math bug cannot hide here the way it could behind the CPU twin, which compiles the
    same generated header.  Bounds: residual 1e-9 of the row's term magnitude; Jacobian within the
    forward-difference noise bound with factor 4; identical zero pattern."""
    from oracle import np_path, program_eval
    G = golden("cfg_" + name)
    prob, obj, eng, tw = _engine_and_twin(name)
    lb, ub = np_path.bounds_arrays(prob)
    k = G["x"].shape[0] - 1
    x = G["x"][k]
    cols = np.arange(eng.n)
    F_np, h, JT_np = np_path.sweep(prob, obj, x, cols)
    F0, JT = eng.sweep_stacked(x, h)
    scale = row_scales(eng, prob, x, F_np)
    assert np.all(np.abs(F0 - F_np) <= 1e-9 * scale)
    assert np.all(np.abs(program_eval.evaluate(eng.program, prob, x) - F0) <= 1e-9 * scale)
    err, bound = np.abs(JT[cols] - JT_np), fd_noise_bound(JT_np, scale, h[cols])
    assert np.all(err <= bound), "worst ratio %.3g" % np.max(err / np.maximum(bound, 1e-300))
    assert_zero_pattern(eng.program, cols, JT[cols], JT_np)
    if eng.n <= 320:
        # and SciPy's own approx_derivative (the third-party code the reference really runs: scipy/optimize/
        # _slsqp_py.py:299-313 -> _numdiff.approx_derivative) on the NumPy callbacks, here on the GPU box
        try:
            from scipy.optimize._numdiff import approx_derivative
        except Exception:
            approx_derivative = None
        if approx_derivative is not None:
            def stacked(p):
                return np_path.stacked_values(prob, obj, np.array(p, dtype=float))
            J_sp = approx_derivative(stacked, x, method="2-point", abs_step=np_path.ABS_STEP, bounds=(lb, ub))
            err, bound = np.abs(JT - J_sp.T), fd_noise_bound(J_sp.T, scale, h)
            assert np.all(err <= bound), "vs scipy approx_derivative: worst ratio %.3g" % np.max(err / np.maximum(bound, 1e-300))
            assert np.array_equal(J_sp.T, np_path.sweep(prob, obj, x)[2])      # the restatement IS SciPy's loop
    eng.close()


@pytest.mark.parametrize("name,splits", [(n, (2, 3, 8)) for n in SMALL] + [("low_thrust", (4,)), ("launch4", (8,))])
def test_column_sharding_is_bitwise_invariant(name, splits, golden):
    """sharding FD columns must not change a single bit - also on the two
    configurations BASELINE.json shards (C4 over 4, C5 over 8)."""
    G = golden("cfg_" + name)
    prob, obj, eng, tw = _engine_and_twin(name)
    x, h = G["x"][1], G["h"][1]
    _, full = eng.sweep_stacked(x, h)
    for parts in splits:
        edges = np.linspace(0, eng.n, parts + 1).astype(int)
        pieces = [eng.sweep_stacked(x, h, lo, hi)[1] for lo, hi in zip(edges[:-1], edges[1:])]
        assert np.array_equal(np.vstack(pieces), full)
    eng.close()


@pytest.mark.parametrize("name", ALL)
def test_structured_sweep_equals_dense_sweep(name, golden, monkeypatch):
    """The structured sweep skips (row, column) pairs without a data dependency.  It runs either as ONE
    launch together with the evaluation of F(x0) (ogk_fused, where every sweep workgroup recomputes the
    base values it needs; the default up to 100 MB of Jacobian) or as two launches (ogk_eval, ogk_sweep);
    the literal dense sweep (OGPSX_SWEEP=dense) evaluates everything.  They must agree entry for entry,
    and the structural zeros must be exact zeros in all of them."""
    G = golden("cfg_" + name)
    x, h = G["x"][-1], G["h"][-1]
    out = {}
    for layout in ("fused", "split", "dense"):
        monkeypatch.setenv("OGPSX_SWEEP", layout)
        prob, obj, eng, tw = _engine_and_twin(name)
        assert eng.sweep_mode == layout
        out[layout] = eng.sweep_stacked(x, h)
        if layout == "fused":                        # back to back on one handle: the ticket resets itself
            for _ in range(3):
                again = eng.sweep_stacked(x, h)
                assert np.array_equal(again[0], out[layout][0]) and np.array_equal(again[1], out[layout][1])
            lo, hi = eng.n // 4, eng.n // 2
            assert np.array_equal(eng.sweep_stacked(x, h, lo, hi)[1], out[layout][1][lo:hi])
        eng.close()
    F_s, JT_s = out["fused"]
    for layout in ("split", "dense"):
        assert np.array_equal(F_s, out[layout][0])
        assert np.array_equal(JT_s, out[layout][1])
    assert (JT_s != 0).mean() < 0.2


def test_launch_form(monkeypatch):
    """og_fd_sweep(_dev) runs evaluation + structured sweep as ONE launch at every size when the output is a
    registered persistent-zero buffer (the host-pointer entry points register their own); an unregistered
    buffer gets the two-launch form from the same handle, with identical results (include/ogpsx.h
    og_sweep_mode); OGPSX_SWEEP overrides."""
    import torch
    from opengoddard_amd.engine import HipEngine
    from oracle import np_path
    monkeypatch.delenv("OGPSX_SWEEP", raising=False)
    for name in ("goddard", "polar_tsto", "launch4"):
        prob, obj = problems.build(name)
        eng = HipEngine(prob, obj)
        assert eng.sweep_mode == "fused"
        if name != "launch4":
            lb, ub = np_path.bounds_arrays(prob)
            x = np.clip(prob.p, lb, ub)
            h = _native.fd_step(x, lb, ub)
            F0, JT = eng.sweep_stacked(x, h)                      # registered staging buffer: one launch
            dev = torch.device("cuda", 0)
            d_x, d_h = torch.from_numpy(x).to(dev), torch.from_numpy(h).to(dev)
            d_F = torch.empty(eng.m, dtype=torch.float64, device=dev)
            d_JT = torch.full((eng.n, eng.m), 3.0, dtype=torch.float64, device=dev)      # unregistered: two launches
            eng.sweep_dev(d_x.data_ptr(), d_h.data_ptr(), 0, eng.n, d_JT.data_ptr(), d_F.data_ptr(),
                          torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert np.array_equal(d_JT.cpu().numpy(), JT) and np.array_equal(d_F.cpu().numpy(), F0)
        eng.close()
    monkeypatch.setenv("OGPSX_SWEEP", "split")
    prob, obj = problems.build("goddard")
    eng = HipEngine(prob, obj)
    assert eng.sweep_mode == "split"
    eng.close()


@pytest.mark.parametrize("name", ["polar_tsto", "low_thrust", "launch4"])
def test_full_size_jacobian_has_the_pseudospectral_structure(name, golden):
    """Size-independent properties of the equality Jacobian at BASELINE.json's full sizes
    (SURVEY.md Appendix C): for phase i, state s the block d(defect_s)/d(state_s) is the LGL
    differentiation matrix D_i off the diagonal (the dynamics term is node-local), blocks
    between different states of a phase are diagonal, blocks between different phases vanish
    except through the phase times, and every row of D_i sums to zero."""
    G = golden("cfg_" + name)
    prob, obj, eng, tw = _engine_and_twin(name)
    x, h = G["x"][0], G["h"][0]
    F0, JT = eng.sweep_stacked(x, h)
    P = eng.program
    nphase = len(P.nodes)
    tf_cols = set(range(eng.n - nphase, eng.n))
    slots = []
    for g in P.groups:
        if g.kind == "defect":
            for (row, _), sl in zip(g.outputs, g.mv_slots):
                slots.append((g.phase, row, P.mv[sl].leaf_base, g.length))
    for phase, row, leaf, N in slots:
        D = prob.D[phase]
        block = JT[leaf:leaf + N, row:row + N].T             # [k, l] = d defect(k) / d x(l)
        off = ~np.eye(N, dtype=bool)
        scale = np.abs(D).max()
        assert np.max(np.abs(block[off] - D[off])) <= 2e-6 * scale      # FD of a linear map
        assert abs(D.sum(axis=1)).max() <= 1e-9 * scale
        for phase2, row2, leaf2, N2 in slots:
            other = JT[leaf2:leaf2 + N2, row:row + N].T      # d defect_s(k) / d state_s'(l)
            if phase2 != phase:
                assert not other.any()                        # exact structural zeros
            elif leaf2 != leaf:
                assert not other[~np.eye(N, dtype=bool)].any()          # node-local coupling
    # defect rows depend on no variable of another phase except the phase times
    for phase, row, leaf, N in slots:
        cols = np.nonzero(JT[:, row:row + N].any(axis=1))[0]
        lo = 0 if phase == 0 else prob.div[phase - 1][-1]
        hi = prob.div[phase][-1]
        assert all((lo <= c < hi) or (c in tf_cols) for c in cols)
    eng.close()


@pytest.mark.parametrize("layout", ["fused", "split"])
@pytest.mark.parametrize("name,state", [("goddard", 2), ("polar_tsto", 4)])
def test_non_finite_rows_propagate_like_dense_fd(name, state, layout, monkeypatch):
    """A row that is NaN/inf at x0 makes its whole Jacobian row NaN in SciPy's dense FD
    ((NaN - NaN)/dx); the structured sweep must reproduce that, not write zeros.  (In the fused launch
    this is the one thing the sweep workgroups take from the evaluation workgroups of the same kernel:
    the evaluation workgroup that draws the last ticket sees the count of non-finite rows and fills every row
    from z around the positions the sweep workgroups write; unregistered buffers take the two-launch form.)"""
    from opengoddard_amd.engine import HipEngine
    from oracle import np_path, twin
    monkeypatch.setenv("OGPSX_SWEEP", layout)
    prob, obj = problems.build(name)
    lb, ub = np_path.bounds_arrays(prob)
    x = np.clip(prob.p, lb, ub)
    x[prob.index_states(state, 0, 7)] = 0.0      # mass = 0 at one node -> division by zero
    eng = HipEngine(prob, obj)
    assert eng.sweep_mode == layout
    tw = twin.Twin(prob, obj, program=eng.program, header=eng.header)
    h = _native.fd_step(x, lb, ub)
    F0c, JTc = tw.sweep(x, h)
    assert not np.isfinite(F0c).all()
    for _ in range(2):
        F0, JT = eng.sweep_stacked(x, h)
        assert np.array_equal(F0, F0c, equal_nan=True)
        assert np.array_equal(np.isnan(JT), np.isnan(JTc))
        assert np.array_equal(JT, JTc, equal_nan=True)
    # and back to a finite point on the same handle: no NaN may linger in the fill
    x2 = np.clip(prob.p, lb, ub)
    h2 = _native.fd_step(x2, lb, ub)
    F0, JT = eng.sweep_stacked(x2, h2)
    F0c, JTc = tw.sweep(x2, h2)
    assert np.array_equal(F0, F0c) and np.array_equal(JT, JTc)
    eng.close()


@pytest.mark.parametrize("layout", ["fused", "split", "dense"])
@pytest.mark.parametrize("name,state", [("goddard", 2), ("polar_tsto", 4)])
def test_registered_buffer_keeps_its_structural_zeros(name, state, layout, monkeypatch):
    """og_jt_register_dev: a persistent-zero buffer is written only where J_T can be non-zero, and must
    hold exactly what an unregistered (fully rewritten) buffer holds after every sweep - through a
    finite -> non-finite -> non-finite -> finite -> finite sequence on one handle (the NaN fill of a
    non-finite sweep has to be cleaned up by the next one), with the exact-Jacobian mode in between, and
    for a block of columns."""
    import torch
    from opengoddard_amd.engine import HipEngine
    from oracle import np_path, twin
    monkeypatch.setenv("OGPSX_SWEEP", layout)
    prob, obj = problems.build(name)
    lb, ub = np_path.bounds_arrays(prob)
    eng = HipEngine(prob, obj)
    tw = twin.Twin(prob, obj, program=eng.program, header=eng.header)
    n, m = eng.n, eng.m
    x_ok = np.clip(prob.p, lb, ub)
    x_bad = x_ok.copy()
    x_bad[prob.index_states(state, 0, 7)] = 0.0            # mass = 0 at one node -> division by zero
    rng = np.random.default_rng(5)
    x_other = np.clip(x_ok + 1e-3 * rng.standard_normal(n), lb, ub)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    d_F = torch.empty(m, dtype=torch.float64, device=dev)
    lo, hi = n // 3, n // 3 + max(5, n // 2)
    reg_full = torch.full((n, m), 7.0, dtype=torch.float64, device=dev)       # garbage before registration
    reg_part = torch.full((hi - lo, m), -3.0, dtype=torch.float64, device=dev)
    eng.register_jt_dev(reg_full.data_ptr(), 0, n, stream)
    eng.register_jt_dev(reg_part.data_ptr(), lo, hi, stream)
    plain = torch.empty((n, m), dtype=torch.float64, device=dev)

    def sweep(x, into, c0, c1):
        h = _native.fd_step(x, lb, ub)
        d_x, d_h = torch.from_numpy(x).to(dev), torch.from_numpy(h).to(dev)
        eng.sweep_dev(d_x.data_ptr(), d_h.data_ptr(), c0, c1, into.data_ptr(), d_F.data_ptr(), stream)
        torch.cuda.synchronize()
        return into.cpu().numpy().copy(), h

    for step, x in enumerate((x_ok, x_bad, x_bad, x_ok, x_other, x_bad, x_other)):
        plain.fill_(float(step) + 0.5)
        want, h = sweep(x, plain, 0, n)
        got, _ = sweep(x, reg_full, 0, n)
        part, _ = sweep(x, reg_part, lo, hi)
        assert np.array_equal(got, want, equal_nan=True), "registered buffer differs at step %d" % step
        assert np.array_equal(part, want[lo:hi], equal_nan=True), "registered block differs at step %d" % step
        assert np.array_equal(want, tw.sweep(x, h)[1], equal_nan=True)
        assert np.isnan(want).any() == (x is x_bad)
        if step == 2:            # exact-Jacobian mode into the buffer a NaN fill was left in: cleans it too
            d_x = torch.from_numpy(x_ok).to(dev)
            eng.exact_dev(d_x.data_ptr(), 0, n, plain.data_ptr(), d_F.data_ptr(), stream)
            eng.exact_dev(d_x.data_ptr(), 0, n, reg_full.data_ptr(), d_F.data_ptr(), stream)
            torch.cuda.synchronize()
            assert np.array_equal(reg_full.cpu().numpy(), plain.cpu().numpy())
            assert np.isfinite(reg_full.cpu().numpy()).all()
    # the sweep kernel alone (og_fd_columns_dev) honours the registration as well
    h = _native.fd_step(x_ok, lb, ub)
    d_x, d_h = torch.from_numpy(x_ok).to(dev), torch.from_numpy(h).to(dev)
    eng.eval_dev(d_x.data_ptr(), d_F.data_ptr(), stream)
    eng.columns_dev(d_x.data_ptr(), d_h.data_ptr(), 0, n, reg_full.data_ptr(), d_F.data_ptr(), stream)
    torch.cuda.synchronize()
    assert np.array_equal(reg_full.cpu().numpy(), tw.sweep(x_ok, h)[1])
    # after unregistering, the buffer is an ordinary one again (everything rewritten)
    eng.unregister_jt_dev(reg_full.data_ptr())
    reg_full.fill_(11.0)
    got, h = sweep(x_other, reg_full, 0, n)
    assert np.array_equal(got, tw.sweep(x_other, h)[1])
    lib = _native.lib()
    assert lib.og_jt_unregister_dev(eng._handle, reg_full.data_ptr()) != 0
    assert b"not registered" in lib.og_last_error()
    eng.close()


@pytest.mark.parametrize("name", ["goddard", "polar_tsto", "launch4"])
def test_pattern_pack_and_unpack(name, golden):
    """og_pattern is the tracer's static pattern (codegen.sparsity); og_pack_dev gathers exactly those entries
    of a sweep's result and og_unpack_dev puts them back: everything outside the pattern is an exact zero."""
    import ctypes as C
    import torch
    from opengoddard_amd import codegen
    G = golden("cfg_" + name)
    prob, obj, eng, tw = _engine_and_twin(name)
    n, m = eng.n, eng.m
    indptr, rows = eng.pattern()
    want_ptr, want_rows = codegen.sparsity(eng.program)
    assert np.array_equal(indptr, want_ptr) and np.array_equal(rows, want_rows)
    lo, hi = n // 5, n // 5 + n // 2
    ip, rw = eng.pattern(lo, hi)
    assert np.array_equal(ip, indptr[lo:hi + 1] - indptr[lo]) and np.array_equal(rw, rows[indptr[lo]:indptr[hi]])
    x, h = G["x"][-1], G["h"][-1]
    F0, JT = eng.sweep_stacked(x, h)
    dense = np.zeros((n, m), dtype=bool)
    dense[np.repeat(np.arange(n), np.diff(indptr)), rows] = True
    assert not JT[~dense].any()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    lib = _native.lib()
    d_JT = torch.from_numpy(JT[lo:hi].copy()).to(dev)
    d_vals = torch.empty(int(ip[-1]), dtype=torch.float64, device=dev)
    _native.check(lib.og_pack_dev(eng._handle, d_JT.data_ptr(), lo, hi, d_vals.data_ptr(), stream), "og_pack_dev")
    torch.cuda.synchronize()
    vals = d_vals.cpu().numpy()
    assert np.array_equal(vals, JT[lo:hi][np.repeat(np.arange(hi - lo), np.diff(ip)), rw])
    d_out = torch.full((hi - lo, m), 5.0, dtype=torch.float64, device=dev)
    _native.check(lib.og_unpack_dev(eng._handle, d_vals.data_ptr(), lo, hi, d_out.data_ptr(), stream), "og_unpack_dev")
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    assert np.array_equal(out[dense[lo:hi]], JT[lo:hi][dense[lo:hi]]) and np.all(out[~dense[lo:hi]] == 5.0)
    eng.close()


@pytest.mark.parametrize("name,state", [("goddard", 2), ("polar_tsto", 4)])
def test_registered_host_matrix_receives_the_packed_nonzeros(name, state):
    """og_jt_register_host (what Problem.solve's callbacks use): one persistent host matrix, packed non-zeros
    over PCIe, scatter on the host - the same matrix as the dense transfer into a fresh array, through
    finite -> non-finite -> finite points, FD and exact mode."""
    from opengoddard_amd.engine import HipEngine
    from oracle import np_path
    prob, obj = problems.build(name)
    lb, ub = np_path.bounds_arrays(prob)
    eng = HipEngine(prob, obj)
    x_ok = np.clip(prob.p, lb, ub)
    x_bad = x_ok.copy()
    x_bad[prob.index_states(state, 0, 7)] = 0.0
    rng = np.random.default_rng(9)
    x_other = np.clip(x_ok + 1e-3 * rng.standard_normal(eng.n), lb, ub)
    for step, x in enumerate((x_ok, x_bad, x_other, x_bad, x_bad, x_ok)):
        h = _native.fd_step(x, lb, ub)
        F_want, JT_want = eng.sweep_stacked(x, h)
        F_got, JT_got = eng.sweep_persistent(x, h)
        assert JT_got is eng._JT_host
        assert np.array_equal(F_got, F_want, equal_nan=True), "F at step %d" % step
        assert np.array_equal(JT_got, JT_want, equal_nan=True), "J_T at step %d" % step
        if x is not x_bad:
            Fe, JTe = eng.exact_stacked(x)
            Fp, JTp = eng.sweep_persistent(x, h, exact=True)
            assert np.array_equal(JTp, JTe) and np.array_equal(Fp, Fe)
    (g, Jeq, Jineq), _ = eng.jacobians(x_other, lb, ub)
    F_want, JT_want = eng.sweep_stacked(x_other, _native.fd_step(x_other, lb, ub))
    assert np.array_equal(np.vstack([g[None, :], Jeq, Jineq]), JT_want.T)
    eng.close()


@pytest.mark.parametrize("gather,devices", [("peer", [0, 0]), ("peer", [0, 0, 0]), ("peer", [0] * 8), ("rccl", [0])])
@pytest.mark.parametrize("name,state", [("goddard", 2), ("polar_tsto", 4)])
def test_multi_device_handle_reassembles_the_jacobian(name, state, gather, devices, monkeypatch):
    """og_comm_init + og_multi_fd_sweep (include/ogpsx.h): the columns split over G sub-handles of one process,
    the packed non-zeros exchanged, every replica complete.  The 1-GPU box runs G = 2, 3, 8 as G sub-handles on
    device 0 with peer copies (OGPSX_GATHER=peer) and G = 1 with RCCL (ncclCommInitAll + ncclAllGather), and the
    result must be the single-device matrix bit for bit, through non-finite points as well."""
    import ctypes as C
    from opengoddard_amd.engine import HipEngine
    from oracle import np_path
    monkeypatch.setenv("OGPSX_GATHER", gather)
    prob, obj = problems.build(name)
    lb, ub = np_path.bounds_arrays(prob)
    single = HipEngine(prob, obj)
    lib = _native.lib()
    arr = (C.c_int32 * len(devices))(*devices)
    _native.check(lib.og_comm_init(len(devices), arr), "og_comm_init")
    assert lib.og_comm_size() == len(devices) and lib.og_comm_uses_rccl() == (1 if gather == "rccl" else 0)
    desc = _native.OgDesc(abi_version=_native.OG_ABI_VERSION, device=0, n=single.n, m_eq=single.m_eq,
                          m_ineq=single.m_ineq, n_phase=len(single.program.nodes), nodes=single._nodes, D=single._Dptr,
                          cvec=single._cvec.ctypes.data_as(C.POINTER(C.c_double)) if single._cvec.size else None,
                          n_cvec=int(single._cvec.size), module_path=single.module_path.encode())
    mh = C.c_void_p()
    _native.check(lib.og_multi_create(C.byref(desc), C.byref(mh)), "og_multi_create")
    assert lib.og_multi_devices(mh) == len(devices)
    x_ok = np.clip(prob.p, lb, ub)
    x_bad = x_ok.copy()
    x_bad[prob.index_states(state, 0, 7)] = 0.0
    rng = np.random.default_rng(4)
    x_other = np.clip(x_ok + 1e-3 * rng.standard_normal(single.n), lb, ub)
    JT = np.empty((single.n, single.m))
    _native.check(lib.og_multi_jt_register_host(mh, _native.dptr(JT)), "og_multi_jt_register_host")
    F = np.empty(single.m)
    for step, x in enumerate((x_ok, x_bad, x_bad, x_other, x_ok)):
        h = _native.fd_step(x, lb, ub)
        F_want, JT_want = single.sweep_stacked(x, h)
        _native.check(lib.og_multi_fd_sweep(mh, _native.dptr(x), _native.dptr(h), _native.dptr(JT), _native.dptr(F)),
                      "og_multi_fd_sweep")
        assert np.array_equal(F, F_want, equal_nan=True)
        assert np.array_equal(JT, JT_want, equal_nan=True), "step %d" % step
        # every device's replica holds the whole matrix
        for g in range(len(devices)):
            d_full, d_F0, st = C.c_void_p(), C.c_void_p(), C.c_void_p()
            _native.check(lib.og_multi_replica_dev(mh, g, C.byref(d_full), C.byref(d_F0), C.byref(st)),
                          "og_multi_replica_dev")
            host = np.empty_like(JT)
            _native.check(lib.og_device_read(devices[g], d_full, host.ctypes.data_as(C.c_void_p), host.nbytes),
                          "og_device_read")
            assert np.array_equal(host, JT_want, equal_nan=True), "replica %d at step %d" % (g, step)
    lib.og_multi_destroy(mh)
    lib.og_comm_finalize()
    single.close()


def test_engine_with_devices_runs_problem_solve(monkeypatch):
    """HipEngine(devices=[...]) / Problem.solve(devices=...): SciPy's callbacks served by the sharded sweep."""
    from opengoddard_amd.engine import HipEngine
    from oracle import np_path
    monkeypatch.setenv("OGPSX_GATHER", "peer")
    prob, obj = problems.build("goddard")
    lb, ub = np_path.bounds_arrays(prob)
    x = np.clip(prob.p, lb, ub)
    one = HipEngine(prob, obj)
    many = HipEngine(prob, obj, devices=[0, 0, 0])
    (g1, e1, i1), h1 = one.jacobians(x, lb, ub)
    (g2, e2, i2), h2 = many.jacobians(x, lb, ub)
    assert np.array_equal(g1, g2) and np.array_equal(e1, e2) and np.array_equal(i1, i2) and np.array_equal(h1, h2)
    one.close()
    many.close()
    prob2, obj2 = problems.build("brachistochrone")
    prob2.maxIterator = 1
    prob2.solve(obj2, maxiter=100, devices=[0, 0])
    assert prob2.last_result.status == 0 and abs(prob2.last_result.fun - 1.7724562) < 2e-5


def test_sharded_sweep_class_on_the_device(golden):
    """sharding.ShardedSweep with the HIP backend (what bench.py --gpus N runs per rank): the library's shard
    plan equals sharding.plan, and rank r of W leaves its block and nothing else in its replica before the
    exchange (the exchange itself needs W processes: tests/test_sharding_gloo.py drives the same class)."""
    import torch
    from opengoddard_amd import sharding
    G = golden("cfg_polar_tsto")
    prob, obj, eng, tw = _engine_and_twin("polar_tsto")
    x, h = G["x"][0], G["h"][0]
    _, full = eng.sweep_stacked(x, h)
    dev = torch.device("cuda", 0)
    be = sharding.HipBackend(eng, dev)
    for world, rank in ((1, 0), (4, 2), (8, 7)):
        sh = sharding.ShardedSweep(be, eng.n, eng.m, rank, world)
        sh.step(be.upload(x), be.upload(h), gather=False)
        torch.cuda.synchronize()
        got = sh.replica.cpu().numpy()
        assert np.array_equal(got[sh.lo:sh.hi], full[sh.lo:sh.hi])
        assert not got[:sh.lo].any() and not got[sh.hi:].any()
        # the rank's message: its packed non-zeros in pattern order
        be.pack(rank, sh.lo, sh.hi, sh.replica, sh.send)
        torch.cuda.synchronize()
        indptr, rows = eng.pattern()
        want = full[np.repeat(np.arange(sh.lo, sh.hi), np.diff(indptr[sh.lo:sh.hi + 1])), rows[indptr[sh.lo]:indptr[sh.hi]]]
        assert np.array_equal(sh.send.cpu().numpy()[:want.size], want)
        # the one-launch form of sweep + pack (og_shard_sweep_dev) writes the same block and the same message
        sh.send.fill_(-1.0)
        be.sweep_and_pack(rank, be.upload(x), be.upload(h), sh.lo, sh.hi, sh.replica, sh.F0, sh.send)
        torch.cuda.synchronize()
        assert np.array_equal(sh.send.cpu().numpy()[:want.size], want)
        assert np.array_equal(sh.replica.cpu().numpy()[sh.lo:sh.hi], full[sh.lo:sh.hi])
        eng.unregister_jt_dev(sh.replica[sh.lo:sh.hi].data_ptr())
    eng.close()


def test_rank_without_columns_keeps_its_replica_clean_through_nan_points():
    """More ranks than column blocks (goddard: n = 201, 64 ranks of 4 columns, ranks 51.. own nothing): such a rank
    evaluates F(x0) only, and og_shard_unpack_dev keeps the NaN history of its replica itself (it
    used to compare against stand-in state words and left stale NaN rows behind after a non-finite point).  The
    messages of the other ranks are made here from the single-device matrix in the plan's layout."""
    import torch
    from opengoddard_amd import sharding
    from opengoddard_amd.engine import HipEngine
    from oracle import np_path
    prob, obj = problems.build("goddard")
    lb, ub = np_path.bounds_arrays(prob)
    eng = HipEngine(prob, obj)
    dev = torch.device("cuda", 0)
    be = sharding.HipBackend(eng, dev)
    world, rank = 64, 63
    sh = sharding.ShardedSweep(be, eng.n, eng.m, rank, world)
    assert sh.hi <= sh.lo
    indptr, rows = eng.pattern()
    cols = np.repeat(np.arange(eng.n), np.diff(indptr))
    where = np.repeat(sh.offsets, np.diff(indptr)) + (np.arange(rows.size) - np.repeat(indptr[:-1], np.diff(indptr)))
    x_ok = np.clip(prob.p, lb, ub)
    x_bad = x_ok.copy()
    x_bad[prob.index_states(2, 0, 7)] = 0.0
    rng = np.random.default_rng(11)
    x_other = np.clip(x_ok + 1e-3 * rng.standard_normal(eng.n), lb, ub)
    single = HipEngine(prob, obj)
    for step, x in enumerate((x_ok, x_bad, x_other, x_bad, x_bad, x_ok, x_other)):
        h = _native.fd_step(x, lb, ub)
        F_want, JT_want = single.sweep_stacked(x, h)
        message = np.zeros(sh.recv.numel())
        message[where] = JT_want[cols, rows]
        sh.recv.copy_(torch.from_numpy(message))
        be.sweep_and_pack(rank, be.upload(x), be.upload(h), sh.lo, sh.hi, sh.replica, sh.F0, sh.send)
        be.unpack(rank, sh.recv, sh.replica)
        torch.cuda.synchronize()
        assert np.array_equal(sh.F0.cpu().numpy(), F_want, equal_nan=True)
        assert np.array_equal(sh.replica.cpu().numpy(), JT_want, equal_nan=True), "step %d" % step
    single.close()
    eng.close()


@pytest.mark.parametrize("name,state", [("goddard", 2), ("polar_tsto", 4)])
def test_one_launch_sweep_replays_from_a_captured_graph(name, state):
    """The launch arguments of a sweep into a registered buffer are pointers only - the count of non-finite rows,
    the ticket and the buffer's launch number live on the device - so the launch can be captured once into a
    hipGraph and replayed at new points (written into the same device vectors), through non-finite points too."""
    import torch
    from opengoddard_amd.engine import HipEngine
    from oracle import np_path, twin
    prob, obj = problems.build(name)
    lb, ub = np_path.bounds_arrays(prob)
    eng = HipEngine(prob, obj)
    assert eng.sweep_mode == "fused"
    tw = twin.Twin(prob, obj, program=eng.program, header=eng.header)
    n, m = eng.n, eng.m
    dev = torch.device("cuda", 0)
    d_x = torch.zeros(n, dtype=torch.float64, device=dev)
    d_h = torch.zeros(n, dtype=torch.float64, device=dev)
    d_F = torch.empty(m, dtype=torch.float64, device=dev)
    d_JT = torch.empty((n, m), dtype=torch.float64, device=dev)
    x_ok = np.clip(prob.p, lb, ub)
    x_bad = x_ok.copy()
    x_bad[prob.index_states(state, 0, 7)] = 0.0
    rng = np.random.default_rng(2)
    x_other = np.clip(x_ok + 1e-3 * rng.standard_normal(n), lb, ub)

    def load(x):
        d_x.copy_(torch.from_numpy(x))
        d_h.copy_(torch.from_numpy(_native.fd_step(x, lb, ub)))

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        eng.register_jt_dev(d_JT.data_ptr(), 0, n, side.cuda_stream)
        load(x_ok)
        eng.sweep_dev(d_x.data_ptr(), d_h.data_ptr(), 0, n, d_JT.data_ptr(), d_F.data_ptr(), side.cuda_stream)   # warm
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        eng.sweep_dev(d_x.data_ptr(), d_h.data_ptr(), 0, n, d_JT.data_ptr(), d_F.data_ptr(),
                      torch.cuda.current_stream().cuda_stream)
    for step, x in enumerate((x_other, x_bad, x_bad, x_ok, x_other, x_bad, x_ok)):
        load(x)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        F_want, JT_want = tw.sweep(x, _native.fd_step(x, lb, ub))
        assert np.array_equal(d_F.cpu().numpy(), F_want, equal_nan=True), "F at replay %d" % step
        assert np.array_equal(d_JT.cpu().numpy(), JT_want, equal_nan=True), "J_T at replay %d" % step
    eng.close()


def test_sweep_is_deterministic(golden):
    G = golden("cfg_polar_tsto")
    prob, obj, eng, tw = _engine_and_twin("polar_tsto")
    x, h = G["x"][0], G["h"][0]
    a = eng.sweep_stacked(x, h)[1]
    b = eng.sweep_stacked(x, h)[1]
    assert np.array_equal(a, b)
    eng.close()


def test_device_pointer_entry_points_match_host_entry_points(golden):
    import torch
    G = golden("cfg_goddard")
    prob, obj, eng, tw = _engine_and_twin("goddard")
    x, h = G["x"][0], G["h"][0]
    F0, JT = eng.sweep_stacked(x, h)
    dev = torch.device("cuda", 0)
    d_x = torch.from_numpy(x).to(dev)
    d_h = torch.from_numpy(h).to(dev)
    d_F = torch.empty(eng.m, dtype=torch.float64, device=dev)
    d_JT = torch.empty((eng.n, eng.m), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    eng.sweep_dev(d_x.data_ptr(), d_h.data_ptr(), 0, eng.n, d_JT.data_ptr(), d_F.data_ptr(), stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_JT.cpu().numpy(), JT)
    assert np.array_equal(d_F.cpu().numpy(), F0)
    eng.close()


def test_lgl_kernel_matches_host_bitwise():
    import ctypes as C
    import torch
    lib = _native.lib()
    dev = torch.device("cuda", 0)
    for n in (3, 4, 5, 20, 50, 80, 128, 200):
        tau, w, D = _native.lgl(n)
        d_tau = torch.empty(n, dtype=torch.float64, device=dev)
        d_w = torch.empty(n, dtype=torch.float64, device=dev)
        d_D = torch.empty((n, n), dtype=torch.float64, device=dev)
        _native.check(lib.og_lgl_dev(n, C.c_void_p(d_tau.data_ptr()), C.c_void_p(d_w.data_ptr()),
                                     C.c_void_p(d_D.data_ptr()), None), "og_lgl_dev")
        assert np.array_equal(d_tau.cpu().numpy(), tau)
        assert np.array_equal(d_w.cpu().numpy(), w)
        assert np.array_equal(d_D.cpu().numpy(), D)


@pytest.mark.parametrize("n", (3, 4, 5, 10, 20, 25, 30, 40, 50, 80, 100, 128, 200))
def test_lgl_kernel_against_the_reference_golden(n):
    """The DEVICE LGL kernels (``og_lgl_dev``: what builds D for every handle) beside the reference's own
    ``_nodes_LGL`` / ``_weight_LGL`` / ``_differentiation_matrix_LGL`` (``optimize.py:183-213``) as captured in
    tests/golden/lgl.npz, with north_star's tolerances: index order identical, tau within 1e-15, w and D within 1e-12
    relative, the same structural zeros (VERDICT r4 #8a: the golden comparison used to live in the CPU suite only)."""
    import ctypes as C
    import os
    import torch
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lgl.npz"))
    lib = _native.lib()
    dev = torch.device("cuda", 0)
    d_tau = torch.empty(n, dtype=torch.float64, device=dev)
    d_w = torch.empty(n, dtype=torch.float64, device=dev)
    d_D = torch.empty((n, n), dtype=torch.float64, device=dev)
    _native.check(lib.og_lgl_dev(n, C.c_void_p(d_tau.data_ptr()), C.c_void_p(d_w.data_ptr()),
                                 C.c_void_p(d_D.data_ptr()), None), "og_lgl_dev")
    tau, w, D = d_tau.cpu().numpy(), d_w.cpu().numpy(), d_D.cpu().numpy()
    rt, rw, rD = G["tau_%d" % n], G["w_%d" % n], G["D_%d" % n]
    assert np.all(np.diff(tau) > 0) and tau[0] == -1.0 and tau[-1] == 1.0
    assert np.max(np.abs(tau - rt)) <= 1e-15
    assert np.array_equal(tau, -tau[::-1])
    assert np.max(np.abs(w / rw - 1.0)) <= 1e-12
    nz = rD != 0
    assert np.array_equal(D == 0, ~nz)
    assert np.max(np.abs(D[nz] / rD[nz] - 1.0)) <= 1e-12
    assert D[0, 0] == -n * (n - 1) * 0.25 and D[-1, -1] == n * (n - 1) * 0.25


def test_hardware_probe():
    """MFMA f64 = k-ordered fma chain; f64 div/sqrt/og_math bit-identical host vs device."""
    import os
    import subprocess
    probe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                         "tools", "_build", "gpu_probe")
    if not os.path.exists(probe):
        pytest.skip("probe binary not built (run __graft_entry__.build())")
    out = subprocess.run([probe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout


def test_cabi_error_paths_of_the_round2_entry_points():
    """Registration, pattern, pack / unpack, shard and multi-device entry points report misuse through return
    codes + og_last_error(), never by crashing."""
    import ctypes as C
    import torch
    from opengoddard_amd.engine import HipEngine
    lib = _native.lib()
    prob, obj = problems.build("brachistochrone")
    eng = HipEngine(prob, obj)
    n, m, hnd = eng.n, eng.m, eng._handle
    dev = torch.device("cuda", 0)
    buf = torch.zeros((n, m), dtype=torch.float64, device=dev)
    vals = torch.zeros(n * m, dtype=torch.float64, device=dev)
    err = lambda: lib.og_last_error()
    assert lib.og_jt_register_dev(hnd, None, 0, n, None) != 0 and b"null" in err()
    assert lib.og_jt_register_dev(hnd, buf.data_ptr(), 3, 3, None) != 0 and b"bad column range" in err()
    assert lib.og_jt_register_dev(hnd, buf.data_ptr(), 0, n + 1, None) != 0
    assert lib.og_jt_unregister_dev(hnd, buf.data_ptr()) != 0 and b"not registered" in err()
    assert lib.og_jt_register_dev(hnd, buf.data_ptr(), 0, n, None) == 0
    assert lib.og_jt_register_dev(hnd, buf.data_ptr(), 0, n // 2, None) == 0          # re-registration: new range
    assert lib.og_jt_unregister_dev(hnd, buf.data_ptr()) == 0
    host = np.empty((n, m))
    assert lib.og_jt_register_host(hnd, None, 0, n) != 0
    assert lib.og_jt_register_host(hnd, _native.dptr(host), 0, 0) != 0 and b"bad column range" in err()
    assert lib.og_jt_unregister_host(hnd, _native.dptr(host)) != 0 and b"not registered" in err()
    nnz = C.c_int64(-1)
    assert lib.og_pattern(hnd, 5, 2, C.byref(nnz), None, None) != 0 and b"bad column range" in err()
    assert lib.og_pattern(hnd, 4, 4, C.byref(nnz), None, None) == 0 and nnz.value == 0
    assert lib.og_pack_dev(hnd, None, 0, n, vals.data_ptr(), None) != 0
    assert lib.og_pack_dev(hnd, buf.data_ptr(), -1, n, vals.data_ptr(), None) != 0
    assert lib.og_unpack_dev(hnd, vals.data_ptr(), 0, n + 1, buf.data_ptr(), None) != 0
    assert lib.og_shard_pack_dev(hnd, 0, buf.data_ptr(), vals.data_ptr(), None) != 0 and b"no shard plan" in err()
    assert lib.og_shard_plan(hnd, 0, None, None) != 0
    B, bv = C.c_int32(), C.c_int64()
    assert lib.og_shard_plan(hnd, 3, C.byref(B), C.byref(bv)) == 0 and B.value == -(-n // 3) and bv.value > 0
    assert lib.og_shard_pack_dev(hnd, 3, buf.data_ptr(), vals.data_ptr(), None) != 0 and b"rank out of range" in err()
    # unpack into a replica whose own block was never registered
    assert lib.og_shard_unpack_dev(hnd, 1, vals.data_ptr(), buf.data_ptr(), None) != 0 and b"not a registered" in err()
    # more ranks than columns: trailing ranks own nothing and still plan / pack / unpack
    assert lib.og_shard_plan(hnd, n + 5, C.byref(B), C.byref(bv)) == 0 and B.value == 1
    assert lib.og_shard_pack_dev(hnd, n + 4, buf.data_ptr(), vals.data_ptr(), None) == 0
    lib.og_comm_finalize()
    mh = C.c_void_p()
    assert lib.og_multi_create(None, C.byref(mh)) != 0
    desc = _native.OgDesc(abi_version=_native.OG_ABI_VERSION, device=0, n=n, m_eq=eng.m_eq, m_ineq=eng.m_ineq,
                          n_phase=1, nodes=eng._nodes, D=eng._Dptr, cvec=None, n_cvec=int(eng._cvec.size),
                          module_path=eng.module_path.encode())
    assert lib.og_multi_create(C.byref(desc), C.byref(mh)) != 0 and b"og_comm_init first" in err()
    bad = (C.c_int32 * 2)(0, 99)
    assert lib.og_comm_init(2, bad) != 0 and b"out of range" in err()
    twice = (C.c_int32 * 2)(0, 0)
    os_env = __import__("os").environ
    saved = os_env.pop("OGPSX_GATHER", None)
    assert lib.og_comm_init(2, twice) != 0 and b"listed twice" in err()
    if saved is not None:
        os_env["OGPSX_GATHER"] = saved
    assert lib.og_comm_size() == 0 or lib.og_comm_size() == 2
    lib.og_comm_finalize()
    assert lib.og_multi_fd_sweep(None, None, None, None, None) != 0
    assert lib.og_trace_read(hnd, _native.dptr(host), 8) != 0 and b"OGPSX_TRACE" in err()
    torch.cuda.synchronize()
    eng.close()


def test_cabi_error_paths_on_device(golden):
    """The C ABI reports misuse through return codes + og_last_error(), never by crashing."""
    import ctypes as C
    from opengoddard_amd.engine import HipEngine
    lib = _native.lib()
    prob, obj = problems.build("brachistochrone")
    eng = HipEngine(prob, obj)
    n, m = eng.n, eng.m
    dims = [C.c_int32() for _ in range(4)]
    assert lib.og_problem_dims(eng._handle, *[C.byref(d) for d in dims]) == 0
    assert [d.value for d in dims] == [n, m, eng.m_eq, eng.m_ineq]
    x = np.zeros(n)
    JT = np.zeros((n, m))
    assert lib.og_fd_sweep(eng._handle, _native.dptr(x), _native.dptr(x), 3, 2, _native.dptr(JT), None) != 0
    assert b"bad column range" in lib.og_last_error()
    assert lib.og_fd_sweep(eng._handle, _native.dptr(x), _native.dptr(x), 0, n + 1, _native.dptr(JT), None) != 0
    assert lib.og_eval(eng._handle, None, _native.dptr(JT)) != 0
    assert b"null" in lib.og_last_error()
    # an empty column range is legal and touches nothing
    sentinel = np.full((1, m), 7.0)
    F0 = np.zeros(m)
    h = np.full(n, 1e-8)
    assert lib.og_fd_sweep(eng._handle, _native.dptr(np.clip(prob.p, 0, None)), _native.dptr(h), 5, 5,
                           _native.dptr(sentinel), _native.dptr(F0)) == 0
    assert np.all(sentinel == 7.0) and np.isfinite(F0).any()
    # descriptor / module mismatch
    desc = _native.OgDesc(abi_version=_native.OG_ABI_VERSION, device=0, n=n + 1, m_eq=eng.m_eq,
                          m_ineq=eng.m_ineq, n_phase=1, nodes=eng._nodes, D=eng._Dptr, cvec=None,
                          n_cvec=int(eng._cvec.size), module_path=eng.module_path.encode())
    handle = C.c_void_p()
    assert lib.og_problem_create(C.byref(desc), C.byref(handle)) != 0
    assert b"does not match" in lib.og_last_error() and not handle.value
    desc.n = n
    desc.module_path = b"/nonexistent/libogk.so"
    assert lib.og_problem_create(C.byref(desc), C.byref(handle)) != 0
    assert b"dlopen" in lib.og_last_error()
    desc.module_path = eng.module_path.encode()
    desc.device = 99
    assert lib.og_problem_create(C.byref(desc), C.byref(handle)) != 0
    assert b"device" in lib.og_last_error()
    eng.close()
    eng.close()                                   # idempotent
