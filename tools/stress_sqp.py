#!/usr/bin/env python3
"""Dev tool (GPU box): the same QP subproblems solved over and over on one handle must give the same bits every time -
the LQ sweep's look-ahead launch (head workgroups, panel workgroup) and the chained triangular solves hand data
between workgroups inside a launch, and a race there would show up as a difference between repetitions.
    python tools/stress_sqp.py [workload] [repetitions]
Solves the first subproblem of the workload (cold, then warm-started from its own active set) `repetitions` times."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opengoddard_amd import _native, _sqp_native, problems
from opengoddard_amd.engine import HipEngine
from oracle import np_path

name = sys.argv[1] if len(sys.argv) > 1 else "polar_tsto"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
prob, obj = problems.build(name)
eng = HipEngine(prob, obj)
lb, ub = np_path.bounds_arrays(prob)
x = np.clip(prob.p, lb, ub)
F, JT = eng.sweep_stacked(x, _native.fd_step(x, lb, ub))
n, meq, mg = eng.n, eng.m_eq, eng.m_ineq
A = JT[:, 1:].T.copy()
g = JT[:, 0].copy()
c = F[1:].copy()
core = _sqp_native.QpCore(n, meq, mg)
rho = 100.0
extra = np.concatenate([-c[:meq], np.maximum(-c[meq:], 0.0)])
lo, hi = np.append(lb - x, 0.0), np.append(ub - x, 1.0)
first = None
t0 = time.time()
bad = 0
for rep in range(reps):
    for mode in ("cold", "warm"):
        core.reset()
        if mode == "cold":
            core.set_active()
        out = core.solve(A, g, c, lb - x, ub - x)
        if out[3] != 1:                                     # inconsistent linearisation: the relaxed subproblem
            out = core.solve(A, g, c, lo, hi, True, rho)
        key = (mode,)
        sig = (out[0].tobytes(), out[1].tobytes(), out[3], out[4])
        if first is None:
            first = {}
        if mode not in first:
            first[mode] = sig
            print(name, mode, "status", out[3], "changes", out[4], "|d|", float(np.abs(out[0]).max()), flush=True)
        elif sig != first[mode]:
            bad += 1
            print("DIFFERENT bits in repetition", rep, mode, "status", out[3], "changes", out[4],
                  "max |d - d0|", float(np.abs(out[0] - np.frombuffer(first[mode][0])).max()), flush=True)
print("%s: %d repetitions x (cold, warm) in %.1f s, %d differed" % (name, reps, time.time() - t0, bad))
sys.exit(1 if bad else 0)
