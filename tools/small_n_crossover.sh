#!/bin/bash
# usage (GPU box): tools/small_n_crossover.sh <round> - where does the HIP SQP core overtake SciPy's Fortran core?
# wall-clock of the same solves with both cores for the small configurations -> gpurun_out/<round>_small_n.jsonl
R=${GRAFT_REPO_ROOT:-/root/repo}
rnd=${1:-r04}
out=$R/gpurun_out/${rnd}_small_n.jsonl
mkdir -p $R/gpurun_out; : > $out
for core in scipy hip; do
  for w in "brachistochrone" "goddard" "table_ascent --max-restarts 4" "polar_tsto_shipped --max-restarts 4" "low_thrust_shipped --max-restarts 2"; do
    timeout 600 python $R/tests/perf/solve_timing.py $w --sqp-core $core 2>/dev/null | tail -1 >> $out
  done
done
python - $out <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print("%-20s n %4d core %-5s wall %6.2f s  callbacks %.3f  converged %s cost %.7g"%(d["workload"],d["n"],d.get("sqp_core","scipy"),d["wall_s"],d["t_callbacks_s"],d["converged"],d["cost"]))
PY
