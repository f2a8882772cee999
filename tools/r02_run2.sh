#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02b; mkdir -p $out
cd $R
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $out/gputest.log
tools/r02_ab.sh polar_tsto low_thrust launch4 goddard > $out/ab.log 2>&1
(OGPSX_SWEEP=fused timeout 300 python bench.py --workload launch4 --no-cpu-baseline --sqp-iterations 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('forced fused launch4 us/step %.2f kernel %.2f' % (1e3*d['ms_per_step'], 1e3*r['kernel_ms_mean']))") >> $out/ab.log 2>&1
for w in polar_tsto low_thrust; do
  (OG_EXTRA_HIPFLAGS=-DOGK_TRACE=1 OGPSX_SWEEP=fused timeout 600 python tools/trace_fused.py $w) > $out/trace_$w.log 2>&1
done
cat $out/gputest.log $out/ab.log $out/trace_polar_tsto.log $out/trace_low_thrust.log
