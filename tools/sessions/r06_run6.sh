#!/bin/bash
# round 6, call 6: the round's profiles - resident launch (bits, in-kernel sections, kernel statistics, PMC), the exchange
# probe, whole solves with the resident launch on / off, the sweep kernels' rocprofv3 statistics and PMC passes
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r06
export PYTHONUNBUFFERED=1
cd $R
timeout 900 python -m pytest tests/test_slsqp_core.py -x -q -m gpu -k "resident or recovers" 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version\|^$" | tail -3
timeout 120 tools/_build/res_probe > gpurun_out/r06/res_probe.txt 2>&1; tail -3 gpurun_out/r06/res_probe.txt
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
out=open("gpurun_out/r06/resident_trace_%s.txt"%w,"w")
out.write("# in-kernel s_memrealtime sections of k_rows_resident (workgroup 0, wavefront 0; -DOGSQP_TRACE, tools/sqp_trace.sh %s 10)\n"%w)
for k,(v,n,pp) in acc.items(): out.write("%-28s %7.2f us per change (%d changes, %d partial steps)\n"%(k, v/max(n,1), n, pp))
out.write("sum of wavefront 0's sections %.2f us per change\n"%sum(v/max(n,1) for k,(v,n,pp) in acc.items() if not k.startswith("(")))
out.close(); print(open(out.name).read())
PY
done
bash tools/sqp_kstats.sh polar_tsto 10 r06_sqp_polar_tsto | tail -8
bash tools/sqp_kstats.sh low_thrust 10 r06_sqp_low_thrust | tail -8
OGSQP_RESIDENT=0 bash tools/sqp_kstats.sh polar_tsto 10 r06_sqp_polar_tsto_two_launch | tail -6
bash tools/sqp_pmc.sh polar_tsto 10 r06_sqp_polar_tsto > /dev/null 2>&1; head -12 gpurun_out/r06_sqp_polar_tsto_pmc.txt
for form in 1 0; do
  OGSQP_RESIDENT=$form timeout 900 python tests/perf/solve_timing.py polar_tsto --sqp-core hip --maxiter 400 2>/dev/null | tail -1 > gpurun_out/r06/solve_c3_resident$form.json
  OGSQP_RESIDENT=$form timeout 600 python tests/perf/solve_timing.py low_thrust --sqp-core hip 2>/dev/null | tail -1 > gpurun_out/r06/solve_c4_resident$form.json
  OGSQP_RESIDENT=$form timeout 600 python tests/perf/solve_timing.py low_thrust --sqp-core hip 2>/dev/null | tail -1 > gpurun_out/r06/solve_c4_resident$form.json
  python -c "
import json
for c in ('c3','c4'):
    r=json.load(open('gpurun_out/r06/solve_%s_resident$form.json'%c)); print(c,'RESIDENT=$form', {k:r[k] for k in ('wall_s','t_qp_s','qp_solves','active_set_iterations','exit_mode','cost')})"
done
bash tools/capture_profiles.sh r06 2>&1 | tail -3
