#!/bin/bash
# round 6, call 33: the line as the driver runs it, on the final tree
mkdir -p gpurun_out/r06
t0=$(date +%s)
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_driver_style.json 2> gpurun_out/r06/bench_driver_style.err
echo "rc $? in $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06/bench_driver_style.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","host_api_ms_per_sweep","n_gpus","dtype")})
print("roofline", {k:d["roofline"].get(k) for k in ("frac","achieved","traffic","bound")}, "cpu_baseline", {k:d["cpu_baseline"].get(k) for k in ("value","cores","kind")})
print("sqp", {k:d["sqp"].get(k) for k in ("ms_per_major_iteration","ms_per_major_iteration_without_setup","parity_checked","first_qp_same_active_set")})
s=d["solve"]; print("solve", {k:s.get(k) for k in ("wall_s","qp_s","qp_solves","exit_mode","cost")}); st=s.get("starts") or {}; print("starts", {k:st.get(k) for k in ("exit_mode_0","wall_s_median","wall_s_min","wall_s_max")})
for a in s.get("also",[]): print("also", {k:a.get(k) for k in ("workload","wall_s","qp_solves","exit_mode","options")}, {k:(a.get("starts") or {}).get(k) for k in ("exit_mode_0","wall_s_median")})
print("self_check", d.get("self_check"))
PY
