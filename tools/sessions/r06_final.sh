#!/bin/bash
# round 6, final call: the suite, smoke, the driver's line, the per-workload lines, the resident launch's profiles
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r06f
export PYTHONUNBUFFERED=1
rm -f gpurun_out/test_measurements.jsonl
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r06f/gputests.log 2>&1; echo "pytest rc=$?"
grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version\|^$" gpurun_out/r06f/gputests.log | tail -5
python -c "import __graft_entry__ as e; e.smoke()" 2>&1 | tail -1
( time python bench.py --steps 20 --warmup 5 ) 2> gpurun_out/r06f/bench_driver_style.err | tail -1 > gpurun_out/r06f/bench_driver_style.json; tail -3 gpurun_out/r06f/bench_driver_style.err
for w in polar_tsto goddard low_thrust launch4; do
    timeout 1200 python bench.py --workload $w --cpu-seconds 6 --solve-starts 1 $([ $w = launch4 ] && echo --no-solve) 2>/dev/null | tail -1 > gpurun_out/r06f/bench_$w.json
done
for form in 1 0; do
  OGSQP_RESIDENT=$form timeout 900 python tests/perf/solve_timing.py polar_tsto --sqp-core hip --maxiter 400 2>/dev/null | tail -1 > gpurun_out/r06f/solve_c3_resident$form.json
  OGSQP_RESIDENT=$form timeout 600 python tests/perf/solve_timing.py low_thrust --sqp-core hip 2>/dev/null | tail -1 > /dev/null
  OGSQP_RESIDENT=$form timeout 600 python tests/perf/solve_timing.py low_thrust --sqp-core hip 2>/dev/null | tail -1 > gpurun_out/r06f/solve_c4_resident$form.json
done
rm -f tools/_build/libogsqp_trace.so
for w in polar_tsto low_thrust; do
  bash tools/sqp_trace.sh $w 10 > /dev/null
  python - $w <<'PY'
import re,collections,sys
w=sys.argv[1]
acc=collections.OrderedDict()
for line in open("gpurun_out/sqp_trace_%s.log"%w):
    m=re.match(r"\[ogsqp trace\]\s+resident: (.*?)\s+([\d.]+) us per change \((\d+) changes, (\d+) partial", line)
    if m:
        n=int(m.group(3)); acc.setdefault(m.group(1),[0.0,0,0]); acc[m.group(1)][0]+=float(m.group(2))*n; acc[m.group(1)][1]+=n; acc[m.group(1)][2]+=int(m.group(4))
out=open("gpurun_out/r06f/resident_trace_%s.txt"%w,"w")
out.write("# in-kernel s_memrealtime sections of k_rows_resident (workgroup 0, wavefront 0; -DOGSQP_TRACE, tools/sqp_trace.sh %s 10)\n"%w)
for k,(v,n,pp) in acc.items(): out.write("%-28s %7.2f us per change (%d changes, %d partial steps)\n"%(k, v/max(n,1), n, pp))
out.write("sum of wavefront 0's sections %.2f us per change\n"%sum(v/max(n,1) for k,(v,n,pp) in acc.items() if not k.startswith("(")))
out.close()
PY
done
bash tools/sqp_kstats.sh polar_tsto 10 r06f_sqp_polar_tsto > /dev/null 2>&1
bash tools/sqp_kstats.sh low_thrust 10 r06f_sqp_low_thrust > /dev/null 2>&1
python tools/hostapi_timing.py polar_tsto low_thrust launch4 goddard 2>/dev/null | grep host_api > gpurun_out/r06f/hostapi.txt
python - <<'PY'
import json
r=json.load(open("gpurun_out/r06f/bench_driver_style.json"))
print({k:r.get(k) for k in ("value","ms_per_step","host_api_ms_per_sweep","host_api_ms_per_sweep_mean","host_api_path")})
print("roofline", {k:r["roofline"].get(k) for k in ("achieved","frac","frac_source","frac_hip_events_batch_mean","frac_of_latency_floor")})
print("sqp", {k:r["sqp"].get(k) for k in ("ms_per_major_iteration","ms_per_major_iteration_without_setup","qp_s","active_set_iterations","parity_checked")})
s=r["solve"]
print("solve", {k:s.get(k) for k in ("wall_s","qp_s","qp_solves","exit_mode","cost")})
print("starts", {k:v for k,v in s.get("starts",{}).items() if k not in ("per_start","note")})
for a in s.get("also",[]):
    print("also", a.get("workload"), {k:a.get(k) for k in ("wall_s","qp_s","exit_mode","cost")}, {k:v for k,v in a.get("starts",{}).items() if k in("wall_s_median","wall_s_min","wall_s_max","exit_mode_0")})
print("cold", r.get("cold_start_s",{}).get("total_s"))
for c in ("c3","c4"):
    for f in (1,0):
        q=json.load(open("gpurun_out/r06f/solve_%s_resident%d.json"%(c,f))); print(c,"resident",f,{k:q[k] for k in ("wall_s","t_qp_s","qp_solves","cost")})
print(open("gpurun_out/r06f/hostapi.txt").read())
print(open("gpurun_out/r06f/resident_trace_polar_tsto.txt").read())
PY
