#!/bin/bash
# round 6, call 4: resident launch - whole solves, A/B (C3 to exit mode 0, C4 defaults), SQP leg of the bench
mkdir -p gpurun_out/r06
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_slsqp_core.py -x -q -m gpu -k "resident or recovers or warm" 2>&1 | tail -5
for form in 1 0; do
  OGSQP_RESIDENT=$form timeout 900 python tests/perf/solve_timing.py polar_tsto --sqp-core hip --maxiter 400 > gpurun_out/r06/run4_solve_c3_res$form.json 2>gpurun_out/r06/run4_solve_c3_res$form.err
  python -c "
import json; r=json.load(open('gpurun_out/r06/run4_solve_c3_res$form.json')); print('C3 RESIDENT=$form', {k:r[k] for k in ('wall_s','t_qp_s','qp_solves','active_set_iterations','exit_mode','cost')})"
  for i in 1 2; do
  OGSQP_RESIDENT=$form timeout 600 python tests/perf/solve_timing.py low_thrust --sqp-core hip > gpurun_out/r06/run4_solve_c4_res$form.json 2>gpurun_out/r06/run4_solve_c4_res$form.err
  python -c "
import json; r=json.load(open('gpurun_out/r06/run4_solve_c4_res$form.json')); print('C4 RESIDENT=$form', {k:r[k] for k in ('wall_s','t_qp_s','qp_solves','active_set_iterations','exit_mode','cost')})"
  done
done
for form in 1 0; do
  OGSQP_RESIDENT=$form timeout 600 python bench.py --workload low_thrust --no-cpu-baseline --no-cold-start --no-solve --reps 3 --sqp-reference-iterations 0 2>gpurun_out/r06/run4_bench_c4_res$form.err | tail -1 > gpurun_out/r06/run4_bench_c4_res$form.json
  python -c "
import json; s=json.load(open('gpurun_out/r06/run4_bench_c4_res$form.json')).get('sqp',{}); print('C4 leg RESIDENT=$form', {k:s.get(k) for k in ('ms_per_major_iteration','qp_s','callbacks_s','active_set_iterations','parity_checked','recoveries')})"
done
