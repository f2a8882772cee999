#!/bin/bash
# round 6, call 1: the resident active-set launch - parity with the two-launch form, then A/B timings
mkdir -p gpurun_out/r06
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_slsqp_core.py -x -q -m gpu -k "resident or recovers or first_subproblem_of_the_baseline or warm_started or bit_reproducible" 2>&1 | tail -15 > gpurun_out/r06/run1_tests.log
cat gpurun_out/r06/run1_tests.log
for form in 1 0; do
  OGSQP_RESIDENT=$form timeout 600 python bench.py --no-cpu-baseline --no-cold-start --no-solve --reps 3 --sqp-reference-iterations 0 2>gpurun_out/r06/run1_bench_res$form.err | tail -1 > gpurun_out/r06/run1_bench_res$form.json
  python - <<PY
import json
r=json.load(open("gpurun_out/r06/run1_bench_res$form.json"))
s=r.get("sqp",{})
print("RESIDENT=$form", {k:s.get(k) for k in ("ms_per_major_iteration","qp_ms","active_set_iterations","parity_checked","recoveries","wall_s")})
PY
done
for form in 1 0; do
  OGSQP_RESIDENT=$form timeout 900 python tests/perf/solve_timing.py polar_tsto --sqp-core hip --maxiter 400 > gpurun_out/r06/run1_solve_c3_res$form.json 2>gpurun_out/r06/run1_solve_c3_res$form.err
  cat gpurun_out/r06/run1_solve_c3_res$form.json
  OGSQP_RESIDENT=$form timeout 600 python tests/perf/solve_timing.py low_thrust --sqp-core hip > gpurun_out/r06/run1_solve_c4_res$form.json 2>gpurun_out/r06/run1_solve_c4_res$form.err
  cat gpurun_out/r06/run1_solve_c4_res$form.json
done
