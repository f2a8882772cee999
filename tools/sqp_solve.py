"""usage: python tools/sqp_solve.py <problem> <maxiter> <ftol> <core[,core]>   (cores: scipy, hip)
One Problem.solve per core with maxIterator = 1; prints status, counts, objective, wall time and the
SQP core's own timing split.  Used by tools/sqp_kstats.sh."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from opengoddard_amd import problems
name = sys.argv[1]; maxiter = int(sys.argv[2]); ftol = float(sys.argv[3]); cores = sys.argv[4].split(',')
for core in cores:
    prob, obj = problems.build(name)
    prob.maxIterator = 1
    t = time.time()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        prob.solve(obj, maxiter=maxiter, ftol=ftol, sqp_core=core)
    dt = time.time() - t
    r = prob.last_result
    print(name, core, 'status', r.status, 'nit', r.nit, 'nfev', r.nfev, 'njev', r.njev, 'fun %.12g' % r.fun, '%.2fs' % dt,
          getattr(prob, 'sqp_timing', None), flush=True)
