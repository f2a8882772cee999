#!/bin/bash
# round 6, call 21: the KKT tests with C4 at 400 iterations per restart, then the line as the driver runs it
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_solve.py -q -m gpu -k "converged_optimum" 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version\|^$" | tail -3
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_driver_style.json 2> gpurun_out/r06/bench_driver_style.err
tail -c 600 gpurun_out/r06/bench_driver_style.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06/bench_driver_style.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","host_api_ms_per_sweep")})
print("roofline", {k:d["roofline"].get(k) for k in ("frac","achieved","traffic")})
print("sqp", {k:d["sqp"].get(k) for k in ("ms_per_major_iteration","ms_per_major_iteration_without_setup","qp_s","wall_s","qp_solves","active_set_iterations","parity_checked")})
s=d["solve"]; print("solve", {k:s.get(k) for k in ("wall_s","qp_s","qp_solves","exit_mode","cost")}); print("starts", s.get("starts"))
for a in s.get("also",[]): print("also", {k:a.get(k) for k in ("workload","wall_s","qp_s","qp_solves","exit_mode","cost","options")}, a.get("starts"))
print("cold", d.get("cold_start_s"))
PY
