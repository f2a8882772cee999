"""usage: python tools/exact_paths.py <problem> <maxiter> [ftol]   (GPU box)
With exact Jacobians, SciPy's Fortran SLSQP core and the HIP core free-running on the same problem:
exit mode, iteration / evaluation counts, objective and the largest difference of the final iterates."""
import contextlib, io, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from opengoddard_amd import problems
name, maxiter = sys.argv[1], int(sys.argv[2])
ftol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-6
res = {}
for core in ("scipy", "hip"):
    prob, obj = problems.build(name)
    prob.maxIterator = 1
    t = time.time()
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        prob.solve(obj, maxiter=maxiter, ftol=ftol, sqp_core=core, jacobian="exact")
    res[core] = (prob.last_result, time.time() - t)
a, b = res["scipy"][0], res["hip"][0]
print(name, "scipy: mode %d nit %d nfev %d fun %.12g (%.2fs) | hip: mode %d nit %d nfev %d fun %.12g (%.2fs) | max|dx| %.2e"
      % (a.status, a.nit, a.nfev, a.fun, res["scipy"][1], b.status, b.nit, b.nfev, b.fun, res["hip"][1],
         np.max(np.abs(a.x - b.x))))
