"""usage: python tools/sqp_converge.py <problem> [maxiter]   (GPU box)
Problem.solve(sqp_core="hip") with the reference's restart loop until it converges; prints the timing split."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opengoddard_amd import problems
name = sys.argv[1]; maxiter = int(sys.argv[2]) if len(sys.argv) > 2 else 400
prob, obj = problems.build(name)
buf = io.StringIO()
t = time.time()
with contextlib.redirect_stdout(buf):
    prob.solve(obj, maxiter=maxiter, sqp_core="hip")
dt = time.time() - t
tm = prob.sqp_timings
print(name, "status", prob.last_result.status, "restarts", len(tm), "wall %.2f s" % dt,
      {k: round(sum(x[k] for x in tm), 3) for k in ("callbacks", "qp", "bfgs", "qp_solves", "qp_iterations")}, flush=True)
