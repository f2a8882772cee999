#!/usr/bin/env python3
"""Dev tool (GPU box): phase timing of the light workgroups of ogk_fused from in-kernel stamps
(s_memrealtime, 100 MHz: 10 ns per tick, one clock for the whole chip).
    OG_EXTRA_HIPFLAGS=-DOGK_TRACE=1 OGPSX_SWEEP=fused python tools/trace_fused.py [workload]
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
assert "OGK_TRACE" in os.environ.get("OG_EXTRA_HIPFLAGS", ""), "set OG_EXTRA_HIPFLAGS=-DOGK_TRACE=1"
from opengoddard_amd import _native, problems
from opengoddard_amd.engine import HipEngine
name = sys.argv[1] if len(sys.argv) > 1 else "polar_tsto"
prob, obj = problems.build(name)
eng = HipEngine(prob, obj)
lb = np.array([-np.inf if b[0] is None else b[0] for b in prob.bounds])
ub = np.array([np.inf if b[1] is None else b[1] for b in prob.bounds])
x = np.clip(prob.p, lb, ub); h = _native.fd_step(x, lb, ub)
for _ in range(5):
    F0, JT = eng.sweep_stacked(x, h)
flat = JT.ravel()
def records(tag_lo, tag_hi, width):
    """rows: the stamps of one wavefront, with the J_T row (= first column of its workgroup) appended"""
    idx = np.nonzero((flat >= tag_lo) & (flat <= tag_hi))[0]
    idx = idx[idx + width < flat.size]
    recs = np.array([np.r_[flat[i:i + width], i // JT.shape[1]] for i in idx])
    return recs[recs[:, 1] > 1e6] if len(recs) else recs
item = records(1.0e6, 1.0e6 + 1, 8)
serv = records(3.0e6, 3.0e6, 8)
t0 = min(item[:, 1].min(), serv[:, 1].min())
print("%s: %d item / %d service wavefront records; times in us after the first light wavefront started" % (
    name, len(item), len(serv)))
def line(label, v):
    v = (v - t0) * 0.01
    print("   %-38s p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (label, np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max()))
line("start", item[:, 1]); line("after barrier 2", item[:, 2]); line("fill issued", item[:, 3])
w = item[item[:, 0] == 1.0e6 + 1]
line("base products flag seen", w[:, 4]); line("items done", w[:, 5]); line("end (verdict seen)", item[:, 6])
line("service: chain done", serv[:, 3]); line("service: verdict", serv[:, 6])
# which workgroups are the late ones?  second barrier by position in the grid (first column of the workgroup)
order = np.argsort(serv[:, -1])
b2 = (serv[order, 2] - t0) * 0.01
st = (serv[order, 1] - t0) * 0.01
print("   second barrier / start by tenth of the grid (first to last light workgroup):")
for part, (x, y) in enumerate(zip(np.array_split(b2, 10), np.array_split(st, 10))):
    print("      tenth %d: start p50 %5.2f  barrier 2 p50 %5.2f  max %5.2f" % (part, np.median(y), np.median(x), x.max()))
