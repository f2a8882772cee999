#!/usr/bin/env python3
"""Dev tool (GPU box): many sweeps at the same point into rotating registered buffers must be bitwise identical
(the one-launch form meets its wavefronts through LDS flags and a device-side ticket: a race would show here).
    python tools/stress_determinism.py [workload] [sweeps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opengoddard_amd import _native, problems
from opengoddard_amd.engine import HipEngine
name = sys.argv[1] if len(sys.argv) > 1 else "polar_tsto"
count = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
prob, obj = problems.build(name)
eng = HipEngine(prob, obj)
n, m = eng.n, eng.m
lb = np.array([-np.inf if b[0] is None else b[0] for b in prob.bounds])
ub = np.array([np.inf if b[1] is None else b[1] for b in prob.bounds])
x = np.clip(prob.p, lb, ub); h = _native.fd_step(x, lb, ub)
dev = torch.device("cuda", 0)
d_x, d_h = torch.from_numpy(x).to(dev), torch.from_numpy(h).to(dev)
d_F = torch.empty(m, dtype=torch.float64, device=dev)
bufs = [torch.empty((n, m), dtype=torch.float64, device=dev) for _ in range(4)]
stream = torch.cuda.current_stream().cuda_stream
for b in bufs:
    eng.register_jt_dev(b.data_ptr(), 0, n, stream)
eng.sweep_dev(d_x.data_ptr(), d_h.data_ptr(), 0, n, bufs[0].data_ptr(), d_F.data_ptr(), stream)
torch.cuda.synchronize()
ref, Fref = bufs[0].clone(), d_F.clone()
bad = 0
for i in range(count):
    b = bufs[i % 4]
    eng.sweep_dev(d_x.data_ptr(), d_h.data_ptr(), 0, n, b.data_ptr(), d_F.data_ptr(), stream)
    if i % 50 == 49:
        torch.cuda.synchronize()
        for bb in bufs:
            if not torch.equal(bb, ref):
                bad += 1
        if not torch.equal(d_F, Fref):
            bad += 1
torch.cuda.synchronize()
print("%s: %d sweeps, %d mismatching buffers" % (name, count, bad))
sys.exit(1 if bad else 0)
