#!/bin/bash
# round 6, call 3: resident launch v2 (speculative rows with the prices, DPP argmin, presence with (4), y on its own wavefront)
mkdir -p gpurun_out/r06
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_slsqp_core.py -x -q -m gpu -k "resident or recovers" 2>&1 | tail -8 > gpurun_out/r06/run3_tests.log
cat gpurun_out/r06/run3_tests.log
rm -f tools/_build/libogsqp_trace.so
bash tools/sqp_trace.sh polar_tsto 10
python - <<'PY'
import re,collections
acc=collections.OrderedDict()
for line in open("gpurun_out/sqp_trace_polar_tsto.log"):
    m=re.match(r"\[ogsqp trace\]\s+resident: (.*?)\s+([\d.]+) us per change \((\d+) changes, (\d+) partial", line)
    if m:
        n=int(m.group(3)); acc.setdefault(m.group(1),[0.0,0]); acc[m.group(1)][0]+=float(m.group(2))*n; acc[m.group(1)][1]+=n
for k,(v,n) in acc.items(): print("%-26s %7.2f us per change (%d changes)"%(k, v/max(n,1), n))
print("sum of wave 0's sections %.2f"%sum(v/max(n,1) for k,(v,n) in acc.items() if not k.startswith("(")))
PY
cp gpurun_out/sqp_trace_polar_tsto.log gpurun_out/r06/run3_trace_polar_tsto.log
bash tools/sqp_kstats.sh polar_tsto 10 r06_run3_sqp_polar_tsto
for form in 1 0; do
  OGSQP_RESIDENT=$form timeout 600 python bench.py --no-cpu-baseline --no-cold-start --no-solve --reps 3 --sqp-reference-iterations 0 2>gpurun_out/r06/run3_bench_res$form.err | tail -1 > gpurun_out/r06/run3_bench_res$form.json
  python - <<PY
import json
r=json.load(open("gpurun_out/r06/run3_bench_res$form.json"))
s=r.get("sqp",{})
print("RESIDENT=$form", {k:s.get(k) for k in ("ms_per_major_iteration","active_set_iterations","parity_checked","recoveries","wall_s")})
PY
done
