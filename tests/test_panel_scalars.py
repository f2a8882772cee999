"""The scalars of a reflector in the LQ panel (``csrc/ogsqp_lq16.h``: ``fast_sqrt``, ``fast_inverse``) replace ``sqrt()`` and
a division by the hardware's reciprocal-square-root / reciprocal estimates and a few multiply-adds.  The same operation
sequences, restated here in NumPy (fused multiply-add emulated in extended precision), started from estimates as coarse
as a single-precision result (relative error up to 2^-22, coarser than ``v_rsq_f64`` / ``v_rcp_f64`` deliver): the results
are within 2 ulp of the correctly rounded values, and the reflector built from them reflects its row onto the pivot axis
to rounding - what the sweep needs of them (the GPU tests compare whole subproblems with the restatement)."""
import numpy as np

LD = np.longdouble


def fma(a, b, c):
    return np.float64(LD(a) * LD(b) + LD(c))


def fast_sqrt(d, estimate_error):
    y = np.float64((1.0 / np.sqrt(LD(d))) * (1.0 + estimate_error))
    g, h = np.float64(d * y), np.float64(0.5 * y)
    r = fma(-h, g, 0.5)
    g = fma(g, r, g)
    h = fma(h, r, h)
    e = fma(-g, g, d)
    return fma(e, h, g)


def fast_inverse(s, estimate_error):
    y = np.float64((1.0 / LD(s)) * (1.0 + estimate_error))
    e = fma(-s, y, 1.0)
    y = fma(y, e, y)
    e = fma(-s, y, 1.0)
    return fma(y, e, y)


def ulps(value, exact):
    exact64 = np.float64(exact)
    return abs(LD(value) - exact) / LD(np.spacing(abs(exact64)))


def test_fast_sqrt_and_inverse_are_within_two_ulp_from_single_precision_estimates():
    rng = np.random.default_rng(7)
    worst_sqrt = worst_inv = 0.0
    for _ in range(4000):
        d = np.float64(10.0 ** rng.uniform(-20, 20) * rng.uniform(1, 10))
        err = rng.uniform(-1, 1) * 2.0 ** -22
        worst_sqrt = max(worst_sqrt, float(ulps(fast_sqrt(d, err), np.sqrt(LD(d)))))
        worst_inv = max(worst_inv, float(ulps(fast_inverse(d, err), 1.0 / LD(d))))
    assert worst_sqrt <= 2.0 and worst_inv <= 2.0, (worst_sqrt, worst_inv)


def test_the_reflector_from_the_fast_scalars_maps_its_row_onto_the_pivot_axis():
    """sigma = |row|, s = sigma + |x0|, v = row with v0 = sign(x0) s, beta = 1 / (|row|^2 + sigma |x0|): H = I - beta v v'
    is orthogonal to rounding and H row = -sign(x0) sigma e0."""
    rng = np.random.default_rng(11)
    for n in (5, 64, 1500):
        row = rng.normal(size=n) * 10.0 ** rng.uniform(-3, 3)
        Db = np.float64(np.sum(LD(row) ** 2))
        sigma = fast_sqrt(Db, 2.0 ** -23)
        x0 = row[0]
        s = sigma + abs(x0)
        v = row.copy()
        v[0] = s if x0 >= 0 else -s
        beta = fast_inverse(fma(sigma, abs(x0), Db), -2.0 ** -23)
        assert abs(beta * np.dot(v, v) - 2.0) <= 1e-14                 # beta = 2 / v'v: H is orthogonal
        image = row - beta * np.dot(v, row) * v
        alpha = -sigma if x0 >= 0 else sigma
        assert abs(image[0] - alpha) <= 1e-13 * sigma and np.max(np.abs(image[1:])) <= 1e-13 * sigma
