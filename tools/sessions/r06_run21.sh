#!/bin/bash
# round 6, call 25: the callback modules without the backend's memory-clause pass (-mllvm -amdgpu-max-memory-clause=1, 13 % of
# the module's compile time): same step time?  (bit parity: the sweep's self-check inside bench.py compares with the dense sweep)
mkdir -p gpurun_out/r06
for w in polar_tsto low_thrust launch4; do
  for f in "" "-mllvm -amdgpu-max-memory-clause=1"; do
    t0=$(date +%s%N)
    OG_MODULE_HIPFLAGS="$f" timeout 600 python bench.py --workload $w --quick --steps 200 --warmup 20 2>/dev/null | tail -1 > /tmp/line.json
    t1=$(date +%s%N)
    python - "$w" "$f" $(( (t1 - t0)/1000000 )) <<'PY'
import json,sys
d=json.load(open("/tmp/line.json"))
print("%-11s flags [%s]: %.3f us per step, self_check %s, whole run %s ms"%(sys.argv[1], sys.argv[2], 1e3*d["ms_per_step"], (d.get("self_check") or {}).get("equals_dense_sweep_bitwise"), sys.argv[3]))
PY
  done
done
