#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02e; mkdir -p $out
cd $R
(timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > $out/gputest.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['config']['n'], 'us/step %.2f' % (1e3*d['ms_per_step']), r['kernel'], 'kernel us %.2f' % (1e3*r['kernel_ms_mean']), 'frac %.3f' % r['frac'])"; }
for w in polar_tsto low_thrust launch4 goddard low_thrust_shipped; do
  timeout 300 python bench.py --workload $w --quick 2>/dev/null | tail -1 | line "default $w"
done > $out/ab.log 2>&1
for occ in 6 8; do
for w in polar_tsto low_thrust launch4; do
  OG_EXTRA_HIPFLAGS="-DOGK_FUSED_ATTR=__attribute__((amdgpu_waves_per_eu($occ,$occ)))" timeout 400 python bench.py --workload $w --quick 2>/dev/null | tail -1 | line "waves_per_eu=$occ $w"
done
done >> $out/ab.log 2>&1
for w in polar_tsto low_thrust_shipped; do
  (OG_EXTRA_HIPFLAGS=-DOGK_TRACE=1 timeout 600 python tools/trace_fused.py $w) > $out/trace_$w.log 2>&1
done
cat $out/gputest.log $out/ab.log $out/trace_polar_tsto.log $out/trace_low_thrust_shipped.log
