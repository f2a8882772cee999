"""Why does eng.sweep_persistent cost 0.5 ms per call inside bench.py when og_fd_sweep itself takes 0.05 ms?"""
import sys, time, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from opengoddard_amd import _native, problems, sharding
from opengoddard_amd.engine import HipEngine
prob, obj = problems.build("polar_tsto")
eng = HipEngine(prob, obj, device=0)
lb = np.array([-np.inf if b[0] is None else b[0] for b in prob.bounds]); ub = np.array([np.inf if b[1] is None else b[1] for b in prob.bounds])
x0 = np.clip(prob.p, lb, ub); h = _native.fd_step(x0, lb, ub)
def timeit(tag, reps=100):
    for _ in range(12): eng.sweep_persistent(x0, h)
    t0 = time.perf_counter()
    for _ in range(reps): eng.sweep_persistent(x0, h)
    print(tag, "%.4f ms per call" % ((time.perf_counter() - t0) / reps * 1e3), eng.host_path, flush=True)
timeit("fresh engine")
dev = torch.device("cuda", 0)
backend = sharding.HipBackend(eng, dev)
d_x, d_h = backend.upload(x0), backend.upload(h)
timeit("after HipBackend + uploads")
sweeps = [sharding.ShardedSweep(backend, eng.n, eng.m, 0, 1) for _ in range(17)]
timeit("after 17 registered replicas")
for sh in sweeps: sh.step(d_x, d_h, gather=False)
torch.cuda.synchronize()
timeit("after sweeping into them")
pr = cProfile.Profile(); pr.enable()
for _ in range(100): eng.sweep_persistent(x0, h)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(8)
