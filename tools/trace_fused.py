#!/usr/bin/env python3
"""Dev tool (GPU box): phase timing of every workgroup of ogk_fused from in-kernel stamps
(s_memrealtime, 100 MHz: 10 ns per tick, one clock for the whole chip), read back through og_trace_read.
    OG_MODULE_HIPFLAGS=-DOGK_TRACE=1 OGPSX_TRACE=1 OGPSX_SWEEP=fused python tools/trace_fused.py [workload] [--json] [--nodes a,b]
Record per wavefront: kind (0 evaluation, 1 light, 2 heavy part, 3 MFMA tile), start, four phase stamps, end.
``--json`` (bench.py's measured_chain): one JSON line with the dependent chain of a launch - per kind of workgroup the
smallest (last end - first start) over its workgroups, the largest of those over the kinds - and the kernel span.
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
assert "OGK_TRACE" in os.environ.get("OG_MODULE_HIPFLAGS", "") + os.environ.get("OG_EXTRA_HIPFLAGS", ""), \
    "set OG_MODULE_HIPFLAGS=-DOGK_TRACE=1"
os.environ["OGPSX_TRACE"] = "1"
import torch
from opengoddard_amd import _native, problems
from opengoddard_amd.engine import HipEngine
argv = [v for v in sys.argv[1:] if not v.startswith("--")]
as_json = "--json" in sys.argv
name = argv[0] if argv else "polar_tsto"
nodes = sys.argv[sys.argv.index("--nodes") + 1] if "--nodes" in sys.argv else None
if nodes:
    argv = [v for v in argv if v != nodes]
    name = argv[0] if argv else "polar_tsto"
prob, obj = problems.build(name, **({"nodes": [int(v) for v in nodes.split(",")]} if nodes else {}))
eng = HipEngine(prob, obj)
n, m = eng.n, eng.m
lb = np.array([-np.inf if b[0] is None else b[0] for b in prob.bounds])
ub = np.array([np.inf if b[1] is None else b[1] for b in prob.bounds])
x = np.clip(prob.p, lb, ub); h = _native.fd_step(x, lb, ub)
dev = torch.device("cuda", 0)
d_x, d_h = torch.from_numpy(x).to(dev), torch.from_numpy(h).to(dev)
d_F = torch.empty(m, dtype=torch.float64, device=dev)
d_JT = torch.empty((n, m), dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
eng.register_jt_dev(d_JT.data_ptr(), 0, n, stream)
count = 1 << 20
buf = np.empty(count)
for rep in range(6):
    for _ in range(5):
        eng.sweep_dev(d_x.data_ptr(), d_h.data_ptr(), 0, n, d_JT.data_ptr(), d_F.data_ptr(), stream)
    torch.cuda.synchronize()
    _native.check(_native.lib().og_trace_read(eng._handle, _native.dptr(buf), count), "og_trace_read")
rec = buf.reshape(-1, 8)
wg = np.repeat(np.arange(rec.shape[0] // 8), 8)
live = rec[:, 1] > 0
rec, wg = rec[live], wg[live]
t0 = rec[:, 1].min()
us = lambda v: (v - t0) * 0.01
if as_json:
    import json
    # the records hold the last launches stamped; one launch's workgroups share a start within a few us: take the records
    # of the LAST launch (starts within 100 us of the latest start)
    last = rec[:, 1] > rec[:, 1].max() - 10000
    r1, w1 = rec[last], wg[last]
    by_kind = {}
    for kind, label in ((0, "evaluation"), (1, "light"), (2, "heavy_part"), (3, "mfma_tile")):
        sel = r1[:, 0] == kind
        if not sel.any():
            continue
        spans = []
        for g in np.unique(w1[sel]):
            rows = r1[sel & (w1 == g)]
            spans.append((rows[:, 7].max() - rows[:, 1].min()) * 0.01)
        by_kind[label] = float(np.min(spans))
    print(json.dumps({"chain_us": max(by_kind.values()), "by_kind_us": by_kind,
                      "kernel_span_us": float((r1[:, 7].max() - r1[:, 1].min()) * 0.01), "workload": name,
                      "wavefront_records": int(len(r1))}))
    sys.exit(0)
print("%s (%s): %d wavefront records, kernel span %.2f us (first start to last end)" % (
    name, eng.sweep_mode, len(rec), us(rec[:, 7].max())))
names = {0: "evaluation", 1: "light", 2: "heavy part", 3: "MFMA tile"}
def line(label, v):
    v = us(v)
    print("      %-28s p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (label, np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max()))
for kind in (0, 1, 2, 3):
    r = rec[rec[:, 0] == kind]
    if not len(r):
        continue
    print("   %s: %d wavefronts in %d workgroups" % (names[kind], len(r), len(set(wg[rec[:, 0] == kind]))))
    line("start", r[:, 1])
    if kind == 0:
        line("results out", r[r[:, 2] > 0][:, 2])
    if kind == 1:
        line("after the barrier", r[r[:, 2] > 0][:, 2])
        s = r[r[:, 3] > 0]
        if len(s): line("service: chain done", s[:, 3])
        it = r[r[:, 4] > 0]
        if len(it): line("items: flag seen", it[:, 4])
    if kind == 2:
        line("operands staged", r[r[:, 2] > 0][:, 2]); line("base products done", r[r[:, 3] > 0][:, 3])
    if kind == 3:
        line("operands staged", r[r[:, 2] > 0][:, 2]); line("MFMA chain done", r[r[:, 3] > 0][:, 3])
    line("end", r[:, 7])
# who ends last?
order = np.argsort(-rec[:, 7])[:12]
print("   last wavefronts to end: " + ", ".join("%s wg %d @%.2f" % (names[int(rec[i, 0])][:5], wg[i], us(rec[i, 7])) for i in order))
light = rec[rec[:, 0] == 1]
lw = wg[rec[:, 0] == 1]
if len(light):
    print("   light workgroups by tenth of their grid range: start p50 / end p50 / end max")
    o = np.argsort(lw)
    for part, idx in enumerate(np.array_split(o, 10)):
        print("      tenth %d: %5.2f  %5.2f  %5.2f" % (part, np.median(us(light[idx, 1])), np.median(us(light[idx, 7])), us(light[idx, 7]).max()))
    # the slowest item wavefronts of the light workgroups, stamp by stamp (service wavefronts have no 'operands' stamp... they
    # return before the items' stamps: column 6 is 0 for them)
    it = np.where((rec[:, 0] == 1) & (rec[:, 6] > 0))[0]
    it = it[np.argsort(-rec[it, 7])[:10]]
    print("   slowest item wavefronts (light): wg / start / barrier / operands of the column / tails of the first rounds / flag seen / items done / end")
    for i in it:
        print("      wg %4d  %s" % (wg[i], "  ".join("%6.2f" % us(rec[i, c]) for c in (1, 2, 3, 6, 4, 5, 7))))
