import contextlib, io, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opengoddard_amd import problems
scale = float(sys.argv[1]); maxiter = int(sys.argv[2]); ftol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-6
prob, obj = problems.build("launch4", effort_scale=scale)
prob.maxIterator = 1
buf = io.StringIO(); t = time.time()
with contextlib.redirect_stdout(buf):
    prob.solve(obj, maxiter=maxiter, ftol=ftol)
r = prob.last_result
print(json.dumps({"effort_scale": scale, "ftol": ftol, "wall_s": time.time() - t, "nit": int(r.nit), "status": int(r.status), "cost": float(r.fun),
                  "cost_unscaled": float(r.fun) / scale, "qp_solves": prob.sqp_timing["qp_solves"]}), flush=True)
