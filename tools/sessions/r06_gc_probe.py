"""Does Python's cyclic GC (2e5 module objects of numpy + scipy + torch: 50 ms per full collection) cost a solve anything?"""
import sys, os, gc, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from opengoddard_amd import problems
for mode in ("default", "freeze", "default", "freeze"):
    prob, obj = problems.build("polar_tsto")
    g0 = [s["collections"] for s in gc.get_stats()]
    if mode == "freeze":
        gc.collect(); gc.freeze()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        prob.solve(obj, sqp_core="hip", maxiter=400)
    wall = time.perf_counter() - t0
    if mode == "freeze":
        gc.unfreeze()
    g1 = [s["collections"] for s in gc.get_stats()]
    print(mode, "wall %.2f s" % wall, "collections gen0/1/2:", [b - a for a, b in zip(g0, g1)], "cost", prob.last_result.fun, flush=True)
    prob._engine.close()
