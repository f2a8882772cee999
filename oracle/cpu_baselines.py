"""TEST INFRASTRUCTURE - the CPU baselines ``bench.py`` reports next to the GPU number.

Only ``bench.py``'s ``cpu_baseline`` leg (and tests) run this; it is never part of the product path.  Three ways
to spend host cores on the reference's algorithm, all timed on a bounded sample of the same workload:

``serial``      what the reference does: SciPy's column loop (``scipy/optimize/_numdiff.py:584-625``) calling the
                NumPy restatement of ``equality_add`` / ``cost_add`` / the user inequality
                (``OpenGoddard/optimize.py:670-709``) once per decision variable, one core (``oracle/np_path.py``).
``all_cores``   the same column loop with the columns dealt to ``os.cpu_count()`` worker processes (fork; BLAS
                pinned to one thread per worker).  The reference cannot do this itself - SciPy's loop is serial -
                but it is the fair "whole host" number.
``batch_last``  one evaluation of the unmodified callbacks on an ``(n, B)`` array, FD column last
                (``oracle/batch_last.py``): the strongest honest single-core number.

    python -m oracle.cpu_baselines --workload polar_tsto --mode all_cores --seconds 10     (prints one JSON line)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_STATE = {}


def _setup(name, nodes):
    import numpy as np
    from opengoddard_amd import problems
    from oracle import np_path
    kw = {"nodes": nodes} if nodes else {}
    prob, obj = problems.build(name, **kw)
    lb, ub = np_path.bounds_arrays(prob)
    x0 = np.clip(prob.p, lb, ub)
    return prob, obj, x0


def serial(name, seconds, nodes=None):
    from oracle import np_path
    prob, obj, x0 = _setup(name, nodes)
    n = x0.size
    np_path.stacked_values(prob, obj, x0)                       # warm
    evals, sweeps, t0 = 0, 0, time.perf_counter()
    while True:
        np_path.sweep(prob, obj, x0)                            # n+1 stacked evaluations
        sweeps += 1
        evals += 3 * (n + 1)
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    return {"value": evals / dt, "unit": "callback evals/s", "cores": 1, "kind": "port",
            "sample": "%d full FD sweeps of %s (n=%d, 3(n+1) callback evaluations each) with the NumPy restatement "
                      "of the reference path, serial column loop, %.1f s" % (sweeps, name, n, dt)}


def _worker_init(name, nodes):
    prob, obj, x0 = _setup(name, nodes)
    from oracle import np_path
    f0 = np_path.stacked_values(prob, obj, x0)
    lb, ub = np_path.bounds_arrays(prob)
    _STATE.update(prob=prob, obj=obj, x0=x0, f0=f0, h=np_path.fd_step(x0, lb, ub))


def _worker_columns(cols):
    from oracle import np_path
    s = _STATE
    saved = s["prob"].p
    try:
        np_path.dense_difference(lambda p: np_path.stacked_values(s["prob"], s["obj"], p), s["x0"], s["f0"], s["h"],
                                 list(cols))
    finally:
        s["prob"].p = saved
    return len(cols)


def all_cores(name, seconds, nodes=None, workers=None):
    import multiprocessing as mp
    import numpy as np
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[var] = "1"                                   # one BLAS thread per worker process
    workers = workers or os.cpu_count() or 1
    prob, obj, x0 = _setup(name, nodes)
    n = x0.size
    ctx = mp.get_context("fork")
    with ctx.Pool(workers, initializer=_worker_init, initargs=(name, nodes)) as pool:
        chunks = [c for c in np.array_split(np.arange(n), min(n, workers * 2)) if c.size]
        pool.map(_worker_columns, chunks)                       # warm every worker
        evals, sweeps, t0 = 0, 0, time.perf_counter()
        while True:
            done = sum(pool.map(_worker_columns, chunks))       # one full sweep, columns dealt to the workers
            sweeps += 1
            evals += 3 * (done + 1)
            dt = time.perf_counter() - t0
            if dt >= seconds:
                break
    return {"value": evals / dt, "unit": "callback evals/s", "cores": workers, "kind": "port",
            "sample": "%d full FD sweeps of %s (n=%d), the reference's column loop with the columns dealt to %d "
                      "worker processes (fork, 1 BLAS thread each), %.1f s" % (sweeps, name, n, workers, dt)}


def batch_last(name, seconds, nodes=None):
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(var, "1")                         # one core: the GEMMs of D.X stay single-threaded
    import numpy as np
    from opengoddard_amd import problems
    from oracle import batch_last as bl, np_path
    kw = {"nodes": nodes} if nodes else {}
    prob, obj = problems.build(name, api=bl.api, **kw)
    lb, ub = np_path.bounds_arrays(prob)
    x0 = np.clip(prob.p, lb, ub)
    n = x0.size
    # all n+1 columns at once need (n+1) x n doubles per intermediate: chunks of at most 2048 columns
    width = min(n, 2048)
    bl.sweep(prob, obj, x0, np.arange(min(n, 64)))              # warm
    evals, sweeps, t0 = 0, 0, time.perf_counter()
    while True:
        for c0 in range(0, n, width):
            bl.sweep(prob, obj, x0, np.arange(c0, min(n, c0 + width)))
        sweeps += 1
        evals += 3 * (n + 1)
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    return {"value": evals / dt, "unit": "callback evals/s", "cores": int(os.environ.get("OPENBLAS_NUM_THREADS", "1")),
            "kind": "port",
            "sample": "%d full FD sweeps of %s (n=%d) as batch-last NumPy evaluations of the unmodified callbacks "
                      "(%d columns per call), %.1f s" % (sweeps, name, n, width, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="polar_tsto")
    ap.add_argument("--mode", choices=("serial", "all_cores", "batch_last"), default="serial")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--nodes", default=None)
    ap.add_argument("--workers", type=int, default=None)
    a = ap.parse_args()
    nodes = [int(v) for v in a.nodes.split(",")] if a.nodes else None
    if a.mode == "serial":
        out = serial(a.workload, a.seconds, nodes)
    elif a.mode == "all_cores":
        out = all_cores(a.workload, a.seconds, nodes, a.workers)
    else:
        out = batch_last(a.workload, a.seconds, nodes)
    out["host_cpus"] = os.cpu_count()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
