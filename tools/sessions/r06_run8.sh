#!/bin/bash
# round 6, call 8: resident launch with the next election's prices published before the pass over the rows
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_slsqp_core.py -x -q -m gpu -k "resident or recovers or warm or bit_reproducible or first_subproblem_of_the_baseline" 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version\|^$" | tail -3
rm -f tools/_build/libogsqp_trace.so
bash tools/sqp_trace.sh polar_tsto 10 > /dev/null
python - <<'PY'
import re,collections
acc=collections.OrderedDict()
for line in open("gpurun_out/sqp_trace_polar_tsto.log"):
    m=re.match(r"\[ogsqp trace\]\s+resident: (.*?)\s+([\d.]+) us per change \((\d+) changes, (\d+) partial", line)
    if m:
        n=int(m.group(3)); acc.setdefault(m.group(1),[0.0,0]); acc[m.group(1)][0]+=float(m.group(2))*n; acc[m.group(1)][1]+=n
for k,(v,n) in acc.items(): print("%-28s %7.2f us per change (%d changes)"%(k, v/max(n,1), n))
print("sum of wave 0's sections %.2f"%sum(v/max(n,1) for k,(v,n) in acc.items() if not k.startswith("(")))
PY
bash tools/sqp_kstats.sh polar_tsto 10 r06_run8_sqp_polar_tsto | head -3
bash tools/sqp_kstats.sh low_thrust 10 r06_run8_sqp_low_thrust | head -3
