"""End-to-end on the GPU: Problem.solve driving SciPy SLSQP with HIP-evaluated callbacks and
Jacobians (G5 of SURVEY.md section 8(c): converged objectives, not iteration counts)."""
import time

import numpy as np
import pytest

from opengoddard_amd import problems

pytestmark = pytest.mark.gpu


def test_brachistochrone_converges_to_sqrt_pi(capsys):
    prob, obj = problems.build("brachistochrone")
    prob.solve(obj)
    out = capsys.readouterr().out
    assert "Optimization terminated successfully" in out
    # default ftol is 1e-6, and the iteration path depends on last-ulp rounding
    assert abs(prob.time_final(-1) - np.sqrt(np.pi)) < 2e-5      # reference run: 1.77245410898455
    eng = prob._engine
    assert eng.n_sweeps > 0 and eng.n_values > 0
    eng.close()


def test_goddard_converges_to_literature_altitude(capsys):
    prob, obj = problems.build("goddard")
    t0 = time.perf_counter()
    prob.solve(obj, ftol=1e-10)
    wall = time.perf_counter() - t0
    out = capsys.readouterr().out
    assert "Optimization terminated successfully" in out
    h_end = prob.states_all_section(0)[-1]
    assert abs(h_end - 1.0128342) < 1e-6                        # reference: 1.0128341906902811
    eng = prob._engine
    print("goddard: %.2f s wall, %d sweeps, %d evaluations" % (wall, eng.n_sweeps, eng.n_values))
    eng.close()


def test_solve_records_q13_like_the_reference():
    """After a gradient request the reference leaves prob.p at the last FD column's input:
    p == x except p[-1] = x[-1] + h[-1] (quirk Q13)."""
    prob, obj = problems.build("goddard")
    prob.maxIterator = 1
    prob.solve(obj, maxiter=2)
    eng = prob._engine
    (_, _, _), h = eng._jac
    x = np.frombuffer(eng._jac_key, dtype=np.float64)
    assert np.array_equal(prob.p[:-1], x[:-1]) and prob.p[-1] == x[-1] + h[-1]
    eng.close()
