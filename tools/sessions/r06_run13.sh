#!/bin/bash
# round 6, call 15: whole solves with the new LQ panel (C3 to exit mode 0, C4), QP time per subproblem
mkdir -p gpurun_out/r06
for w in polar_tsto low_thrust; do
  timeout 900 python tests/perf/solve_timing.py $w --sqp-core hip --maxiter 400 2>/dev/null | tail -1 > gpurun_out/r06/solve_${w}_panel_waves.json
  python - $w <<'PY'
import json,sys
d=json.load(open("gpurun_out/r06/solve_%s_panel_waves.json"%sys.argv[1]))
print(sys.argv[1], {k:d[k] for k in d if k in ("wall_s","status","nit","cost","qp_s","n_qp","qp_count","sqp_core_s","callbacks_s","restarts")})
print({k:(v if not isinstance(v,(list,dict)) else "...") for k,v in d.items()})
PY
done
