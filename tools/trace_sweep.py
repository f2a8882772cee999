#!/usr/bin/env python3
"""Dev tool (GPU box): phase timing of the structured sweep from in-kernel shader-clock stamps.
    OG_EXTRA_HIPFLAGS=-DOGK_TRACE=1 python tools/trace_sweep.py [workload]
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
for tag, label, names in ((1.0e6, "light", ["fill", "setup", "items"]), (2.0e6, "tile", ["stage", "mfma", "epilogue"])):
    idx = np.nonzero((flat >= tag) & (flat < tag + 4))[0]
    idx = idx[idx + 5 < flat.size]
    idx = idx[idx + 8 < flat.size]
    recs = np.array([flat[i:i + 8] for i in idx])
    recs = recs[(recs[:, 1] > 1e9) & (recs[:, 4] >= recs[:, 1])] if len(recs) else recs
    if not len(recs):
        print(label, "no records"); continue
    t0 = recs[:, 1].min()
    d = np.diff(recs[:, 1:5], axis=1)
    print("%s: %d records; start spread %.0f..%.0f ticks; end max %.0f ticks after first start" % (
        label, len(recs), (recs[:, 1] - t0).min(), (recs[:, 1] - t0).max(), (recs[:, 4] - t0).max()))
    for i, nm in enumerate(names):
        print("   %-9s mean %8.0f  p50 %8.0f  max %8.0f ticks" % (nm, d[:, i].mean(), np.median(d[:, i]), d[:, i].max()))
    if label == "light":
        dt_core = recs[:, 4] - recs[:, 1]
        dt_real = recs[:, 7] - recs[:, 6]
        ok = dt_real > 0
        print("   shader clock during the kernel: %.0f MHz (s_memtime ticks per 100 MHz s_memrealtime tick)" % (
            100.0 * np.median(dt_core[ok] / dt_real[ok])))
        for w in range(4):
            sel = recs[:, 0] == tag + w
            if sel.any():
                print("   wave %d: n=%d items-phase mean %.0f max %.0f ; items/col mean %.1f" % (w, sel.sum(), d[sel, 2].mean(), d[sel, 2].max(), recs[sel, 5].mean()))
print("(ticks = s_memtime; 100 MHz => 10 ns per tick)")
