#!/bin/bash
# round 6, call 2: in-kernel section times of the resident launch + rocprofv3 kernel stats of the SQP leg
mkdir -p gpurun_out/r06
rm -f tools/_build/libogsqp_trace.so
bash tools/sqp_trace.sh polar_tsto 10
python - <<'PY'
import re,collections
acc=collections.OrderedDict(); tot=0
for line in open("gpurun_out/sqp_trace_polar_tsto.log"):
    m=re.match(r"\[ogsqp trace\]\s+resident: (.*?)\s+([\d.]+) us per change \((\d+) changes, (\d+) partial", line)
    if m:
        n=int(m.group(3)); acc.setdefault(m.group(1),[0.0,0]); acc[m.group(1)][0]+=float(m.group(2))*n; acc[m.group(1)][1]+=n
for k,(v,n) in acc.items(): print("%-24s %7.2f us per change (%d changes)"%(k, v/max(n,1), n))
print("sum of wave 0's sections %.2f"%sum(v/max(n,1) for k,(v,n) in acc.items() if not k.startswith("(")))
PY
cp gpurun_out/sqp_trace_polar_tsto.log gpurun_out/r06/run2_trace_polar_tsto.log
bash tools/sqp_kstats.sh polar_tsto 10 r06_run2_sqp_polar_tsto
