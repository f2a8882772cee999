import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opengoddard_amd import problems
name = sys.argv[1]; maxiter = int(sys.argv[2]); ftol = float(sys.argv[3])
for core in sys.argv[4].split(','):
    for jac in ("fd", "exact"):
        prob, obj = problems.build(name)
        prob.maxIterator = 1
        t = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            prob.solve(obj, maxiter=maxiter, ftol=ftol, sqp_core=core, jacobian=jac)
        r = prob.last_result
        print(name, core, jac, 'status', r.status, 'nit', r.nit, 'nfev', r.nfev, 'fun %.10g' % r.fun, '%.2fs' % (time.time() - t), flush=True)
