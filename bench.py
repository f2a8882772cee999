#!/usr/bin/env python3
"""Benchmark of the hot path: full forward-difference Jacobian sweeps of the NLP callbacks.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One *step* = one SLSQP major iteration's worth of callback work at a fixed point x0 that is
already resident in HBM: F(x0) plus the n forward-difference columns of
[cost | c_eq | c_ineq], written as the transposed Jacobian (n x m, float64) into HBM.  The
reference spends 3n+2 Python callback evaluations on this, so
``value`` = (3n+2) * K / elapsed  [NLP-callback evals/s], BASELINE.json's metric.

Workload: BASELINE.json's target configuration, the 2-phase / 6-state / 3-control /
80-node-per-phase polar ascent (``polar_tsto``, C3, n = 1442).  With N > 1 ranks the columns are
split in contiguous blocks (strong scaling - total work is fixed) and reassembled by one
RCCL all-gather per step; timing is barrier + synchronize on both
sides, max over ranks.

Extra objects on the JSON line: ``roofline`` (dominant kernel - ``ogk_fused``, evaluation + sweep in one
launch, or ``ogk_sweep`` where the runtime uses two launches - against the
HBM roofline, algorithmic bytes 8*[(n+1)n + m n + sum N_i^2] per launch, duration from HIP
events on the launch stream) and ``cpu_baseline`` (the NumPy restatement of the reference path,
``oracle/np_path.py``, timed on this host for ~10 s; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s


def cpu_baseline(name, seconds=10.0):
    """Reference-style serial NumPy sweep (oracle as the *checker-side* baseline, never the
    measured product)."""
    import numpy as np
    from opengoddard_amd import problems
    from oracle import np_path
    prob, obj = problems.build(name)
    lb, ub = np_path.bounds_arrays(prob)
    x0 = np.clip(prob.p, lb, ub)
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
    out = {"value": evals / dt, "unit": "callback evals/s", "cores": 1, "kind": "port",
           "sample": "%d full FD sweeps of %s (n=%d, 3(n+1) callback evaluations each) with the "
                     "NumPy restatement of the reference path, %.1f s" % (sweeps, name, n, dt),
           "host_cpus": os.cpu_count()}
    # context only: the same dense column loop as compiled C++ (oracle/twin.cpp, one core) - how
    # much of the GPU/NumPy ratio is Python interpreter overhead rather than arithmetic
    try:
        from opengoddard_amd import _native
        from oracle import twin
        tw = twin.Twin(prob, obj)
        h = _native.fd_step(x0, lb, ub)
        cols = np.arange(min(n, 256))
        tw.sweep(x0, h, cols[:8])
        t0 = time.perf_counter()
        tw.sweep(x0, h, cols)
        dt2 = time.perf_counter() - t0
        out["compiled_cpp_dense_loop_evals_per_s"] = 3 * (cols.size + 1) / dt2
    except Exception as exc:                                   # never let context break the line
        out["compiled_cpp_dense_loop_evals_per_s"] = None
        out["compiled_cpp_note"] = repr(exc)
    return out


def measured_traffic(workload, kernel="ogk_sweep"):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/rNN_traffic.json, made by tools/summarize_profiles.py; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  None when this workload was not profiled."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json"))):
        try:
            with open(path) as fh:
                entry = json.load(fh).get(workload, {}).get(kernel)
        except (OSError, ValueError):
            continue
        if entry and entry.get("hbm_bytes_per_launch"):
            best = {"bytes": entry["hbm_bytes_per_launch"], "source": os.path.basename(path)}
    return best


def sqp_leg(eng, prob, iterations, reference_iterations=0):
    """Second BASELINE metric, "wall-clock to SLSQP convergence", on a bounded sample: the first
    ``iterations`` major iterations of ``Problem.solve`` with the QP subproblems on the GPU
    (``sqp_core="hip"``), split into callbacks / QP / BFGS.  The same iterations with SciPy's Fortran
    core take 8 s each at n = 1442 (profiles/r01_solve_timing.jsonl), which is why they are not
    re-timed in every bench run."""
    import numpy as np
    from opengoddard_amd import sqp
    lb = np.array([-np.inf if b[0] is None else b[0] for b in prob.bounds], dtype=float)
    ub = np.array([np.inf if b[1] is None else b[1] for b in prob.bounds], dtype=float)
    t0 = time.perf_counter()
    res = sqp.minimize_slsqp_hip(eng, prob.p.copy(), lb, ub, ftol=1e-6, maxiter=iterations + 1)
    wall = time.perf_counter() - t0
    t = res.timing
    out = _sqp_result(res, wall, t, iterations)
    if reference_iterations > 0:
        out["scipy_core"] = scipy_core_sample(eng, prob, lb, ub, reference_iterations)
        out["speedup_per_major_iteration"] = (out["scipy_core"]["ms_per_major_iteration"] /
                                              out["ms_per_major_iteration"])
    return out


def scipy_core_sample(eng, prob, lb, ub, iterations):
    """The same solve with SciPy's Fortran SLSQP core (GPU callbacks and Jacobians, as
    Problem.solve(sqp_core="scipy") runs it), for ``iterations`` major iterations: the measured
    baseline of the SQP leg on this machine."""
    import numpy as np
    from scipy import optimize
    import warnings

    def fun(which):
        return lambda p: eng.values(p)[which]

    def jac(which):
        return lambda p: eng.jacobians(p, lb, ub)[0][which]

    cons = ({"type": "eq", "fun": fun(1), "jac": jac(1)}, {"type": "ineq", "fun": fun(2), "jac": jac(2)})
    t0 = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = optimize.minimize(fun(0), prob.p.copy(), jac=jac(0), bounds=list(zip(lb, ub)), constraints=cons,
                                method="SLSQP", options={"maxiter": iterations + 1, "ftol": 1e-6})
    wall = time.perf_counter() - t0
    done = max(1, int(res.nit) - 1)
    return {"core": "scipy %s _slsqp (Fortran)" % __import__("scipy").__version__, "major_iterations": done,
            "wall_s": wall, "ms_per_major_iteration": 1e3 * wall / done}


def _sqp_result(res, wall, t, iterations):
    return {"core": "hip (include/ogsqp.h)", "major_iterations": int(res.nit - 1 if res.status == 9 else res.nit),
            "exit_mode": int(res.status), "wall_s": wall, "callbacks_s": t["callbacks"], "qp_s": t["qp"],
            "bfgs_s": t["bfgs"], "qp_solves": t["qp_solves"], "active_set_iterations": t["qp_iterations"],
            "ms_per_major_iteration": 1e3 * wall / max(1, res.nit - 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="polar_tsto")
    ap.add_argument("--nodes", default=None,
                    help="comma-separated LGL node counts per phase (size studies; default: the "
                         "workload's own)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sqp-iterations", type=int, default=10,
                    help="major iterations of the SQP leg (0 = skip); skipped above n = 3000")
    ap.add_argument("--sqp-reference-iterations", type=int, default=2,
                    help="major iterations of the same solve with SciPy's Fortran core, timed next to the SQP "
                         "leg (0 = skip; about 8 s each at n = 1442); skipped above n = 1600")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--force-collective", action="store_true",
                    help="initialise RCCL and run the all-gather even with one rank (plumbing test)")
    a = ap.parse_args()

    import numpy as np
    import torch
    from opengoddard_amd import _native, problems, sharding
    from opengoddard_amd.engine import HipEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with "
                         "torch.distributed.run)" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible - the engine has no CPU path to measure")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    collective = world > 1 or (a.force_collective and "MASTER_ADDR" in os.environ)
    if collective:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    build_kw = {"nodes": [int(v) for v in a.nodes.split(",")]} if a.nodes else {}
    prob, obj = problems.build(a.workload, **build_kw)
    eng = HipEngine(prob, obj, device=local_rank)
    n, m = eng.n, eng.m
    lb = np.array([-np.inf if b[0] is None else b[0] for b in prob.bounds])
    ub = np.array([np.inf if b[1] is None else b[1] for b in prob.bounds])
    x0 = np.clip(prob.p, lb, ub)
    h = _native.fd_step(x0, lb, ub)
    d_x = torch.from_numpy(x0).to(dev)
    d_h = torch.from_numpy(h).to(dev)
    d_F0 = torch.empty(m, dtype=torch.float64, device=dev)
    lo, hi = sharding.column_range(n, rank, world)
    rows = sharding.block_rows(n, world)
    d_local = torch.zeros((rows, m), dtype=torch.float64, device=dev)
    d_full = torch.empty(sharding.gathered_shape(n, m, world), dtype=torch.float64, device=dev) \
        if collective else d_local
    stream = torch.cuda.current_stream().cuda_stream
    if hi > lo and not os.environ.get("OG_BENCH_UNREGISTERED"):
        # persistent-zero output: the sweep writes the non-zeros only (og_jt_register_dev)
        eng.register_jt_dev(d_local.data_ptr(), lo, hi, stream)

    def step(gather=True):
        # one pass of the hot path: F(x0) and this rank's FD columns (og_fd_sweep_dev: one launch, or two above
        # 100 MB of Jacobian - og_sweep_mode)
        eng.sweep_dev(d_x.data_ptr(), d_h.data_ptr(), lo, hi, d_local.data_ptr(), d_F0.data_ptr(), stream)
        if collective and gather:
            dist.all_gather_into_tensor(d_full, d_local)

    def fence():
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if collective:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # the same K steps without the all-gather: how the column-sharded kernels alone scale
    # (the collective costs more than the sweep it reassembles)
    elapsed_local = elapsed
    if collective:
        fence()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step(gather=False)
        fence()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_local = float(t.item())

    # duration of the dominant kernel from HIP events on the launch stream.  A single
    # event-to-event interval around one launch carries ~2 us of event overhead (an empty kernel
    # reads 6 us that way, tools/gpu_probe.hip), so the kernel is also timed in back-to-back
    # batches of 10 between two events; the batch figure is the one used for the roofline and is
    # the one that agrees with rocprofv3 --kernel-trace (profiles/).
    def timed(launch, reps, batch):
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                 for _ in range(reps)]
        for e0, e1 in pairs:
            e0.record()
            for _ in range(batch):
                launch()
            e1.record()
        torch.cuda.synchronize()
        return np.array([e0.elapsed_time(e1) / batch for e0, e1 in pairs])

    def launch_sweep():
        eng.sweep_dev(d_x.data_ptr(), d_h.data_ptr(), lo, hi, d_local.data_ptr(), d_F0.data_ptr(), stream)

    def launch_columns():
        eng.columns_dev(d_x.data_ptr(), d_h.data_ptr(), lo, hi, d_local.data_ptr(),
                        d_F0.data_ptr(), stream)

    def launch_eval():
        eng.eval_dev(d_x.data_ptr(), d_F0.data_ptr(), stream)

    fused = eng.sweep_mode == "fused"
    launch_eval()
    single = timed(launch_sweep if fused else launch_columns, a.steps, 1)
    batched = timed(launch_sweep if fused else launch_columns, max(a.steps // 10, 5), 10)
    # the two kernels the fused launch replaces, for reference
    eval_ms_mean = float(np.mean(timed(launch_eval, max(a.steps // 10, 5), 10)))
    columns_ms_mean = float(np.mean(timed(launch_columns, max(a.steps // 10, 5), 10)))
    kern_ms = float(np.median(batched))
    kern_ms_mean = float(np.mean(batched))
    kern_ms_single = float(np.mean(single))

    ncols = hi - lo
    sumN2 = sum(int(v) ** 2 for v in prob.nodes)
    alg_bytes = 8.0 * ((ncols + 1) * n + m * ncols + sumN2)
    achieved = alg_bytes / (kern_ms_mean * 1e-3) / 1e9

    traffic = measured_traffic(a.workload, "ogk_fused" if fused else "ogk_sweep") if world == 1 else None
    result = {
        "metric": "NLP-callback evals/sec (cost+constr+FD-Jacobian)",
        "value": (3 * n + 2) * a.steps / elapsed,
        "unit": "callback evals/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "value_without_collective": (3 * n + 2) * a.steps / elapsed_local,
        "ms_per_step_without_collective": elapsed_local / a.steps * 1e3,
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "%s: %d phases, %s states, %s controls, %s LGL nodes" % (
            a.workload, len(prob.nodes), prob.number_of_states, prob.number_of_controls, prob.nodes),
            "n": n, "m_eq": eng.m_eq, "m_ineq": eng.m_ineq,
            "evals_per_step": 3 * n + 2,
            "parallelism": "fd-columns x%d%s" % (world, " + RCCL all-gather" if collective else "")},
        "roofline": {"bound": "hbm", "kernel": "ogk_fused" if fused else "ogk_sweep", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic["bytes"] if traffic else None,
                     "traffic_source": traffic["source"] if traffic else None,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "bytes_actually_written_per_launch": 8.0 * m * ncols,
                     "kernel_ms_mean": kern_ms_mean, "kernel_ms_median": kern_ms,
                     "kernel_ms_single_launch_events": kern_ms_single,
                     # the two-launch form of the same step (og_fd_sweep above 100 MB of Jacobian, OGPSX_SWEEP=split),
                     # timed in this run: the FD sweep kernel on its own, and evaluation + sweep as a step
                     "split_eval_kernel_ms_mean": eval_ms_mean,
                     "split_sweep_kernel_ms_mean": columns_ms_mean,
                     "split_sweep_kernel_frac": alg_bytes / (columns_ms_mean * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "split_step_frac": alg_bytes / ((columns_ms_mean + eval_ms_mean) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "frac_of_whole_step": alg_bytes / (elapsed / a.steps) / 1e9 / HBM_PEAK_GBS,
                     "note": ("ogk_fused = evaluation of F(x0) + the FD sweep in ONE launch: its duration contains the "
                              "evaluation's latency chain, so its fraction is a whole-step figure; the FD sweep kernel on "
                              "its own (ogk_sweep, two-launch form, timed in this run) is split_sweep_kernel_frac, and the "
                              "two launches as a step split_step_frac") if fused else
                             "two launches per step (ogk_eval, ogk_sweep): frac is the sweep kernel's, frac_of_whole_step "
                             "includes the evaluation"},
    }
    if world == 1 and rank == 0 and not a.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(a.workload, a.cpu_seconds)
        result["speedup_vs_cpu_baseline"] = result["value"] / result["cpu_baseline"]["value"]
    if world == 1 and rank == 0:
        # exact-Jacobian mode (og_jacobian_exact_dev: evaluation + ogk_exact), same resident x0, same output buffer
        d_jt = torch.empty((n, m), dtype=torch.float64, device=dev)
        for _ in range(3):
            eng.exact_dev(d_x.data_ptr(), 0, n, d_jt.data_ptr(), d_F0.data_ptr(), stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            eng.exact_dev(d_x.data_ptr(), 0, n, d_jt.data_ptr(), d_F0.data_ptr(), stream)
        e1.record()
        torch.cuda.synchronize()
        result["exact_jacobian"] = {"ms_per_jacobian": e0.elapsed_time(e1) / reps, "kernel": "ogk_eval + ogk_exact_struct",
                                    "note": "forward-mode derivatives (opt-in mode, jacobian='exact')"}
        del d_jt
    if world == 1 and rank == 0 and a.sqp_iterations > 0 and n <= 3000:
        result["sqp"] = sqp_leg(eng, prob, a.sqp_iterations,
                                a.sqp_reference_iterations if n <= 1600 else 0)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if collective and rank == 0:
        # the gathered matrix must equal this rank's own slab where they overlap
        assert torch.equal(d_full[lo:hi], d_local[:hi - lo])
    eng.close()
    if collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
