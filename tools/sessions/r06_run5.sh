#!/bin/bash
# round 6, call 5: the whole GPU suite, smoke, then the bench line as the driver runs it
mkdir -p gpurun_out/r06
export PYTHONUNBUFFERED=1
rm -f gpurun_out/test_measurements.jsonl
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06/run5_gputests.log 2>&1; echo "pytest rc=$?"
grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version\|^$" gpurun_out/r06/run5_gputests.log | tail -6
cat gpurun_out/test_measurements.jsonl 2>/dev/null
python -c "import __graft_entry__ as e; e.smoke()" 2>&1 | tail -2
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r06/run5_bench.log 2> gpurun_out/r06/run5_bench.err; tail -3 gpurun_out/r06/run5_bench.err
tail -1 gpurun_out/r06/run5_bench.log > gpurun_out/r06/run5_bench.json
python - <<'PY'
import json
r=json.load(open("gpurun_out/r06/run5_bench.json"))
print({k:r[k] for k in ("value","ms_per_step","host_api_ms_per_sweep")})
print("roofline", {k:r["roofline"].get(k) for k in ("achieved","frac","rocprofv3_average_us","frac_from_rocprofv3_average","frac_of_latency_floor")})
print("sqp", {k:r["sqp"].get(k) for k in ("ms_per_major_iteration","qp_s","callbacks_s","active_set_iterations","parity_checked")})
s=r["solve"]
print("solve", {k:s.get(k) for k in ("wall_s","qp_s","qp_solves","exit_mode","cost","resident_active_set")})
print("starts", {k:v for k,v in s.get("starts",{}).items() if k!="per_start" and k!="note"})
for a in s.get("also",[]):
    print("also", a.get("workload"), {k:a.get(k) for k in ("wall_s","qp_s","exit_mode","cost","bounded")}, {k:v for k,v in a.get("starts",{}).items() if k not in("per_start","note")}, (a.get("kkt") or {}).get("stationarity"))
print("cold", r.get("cold_start_s"))
PY
