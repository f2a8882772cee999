#!/bin/bash
# usage (GPU box): tools/r02_ab.sh [workload ...] - bench step with a registered (persistent-zero) output buffer
# and with an ordinary one, fused and split launch forms
R=${GRAFT_REPO_ROOT:-/root/repo}
for reg in registered unregistered; do
  if [ $reg = unregistered ]; then export OG_BENCH_UNREGISTERED=1; else unset OG_BENCH_UNREGISTERED; fi
  for m in default split; do
    if [ $m = split ]; then export OGPSX_SWEEP=split; else unset OGPSX_SWEEP; fi
    for w in ${@:-polar_tsto low_thrust launch4 goddard}; do
        timeout 300 python $R/bench.py --workload $w --no-cpu-baseline --sqp-iterations 0 2>/dev/null | tail -1 | \
            python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$reg $m', '$w', d['config']['n'], 'us/step %.2f' % (1e3*d['ms_per_step']), 'evals/s %.3g' % d['value'], r['kernel'], 'kernel us %.2f' % (1e3*r['kernel_ms_mean']), 'frac %.3f' % r['frac'], 'split eval %.2f sweep %.2f' % (1e3*r['split_eval_kernel_ms_mean'], 1e3*r['split_sweep_kernel_ms_mean']))"
    done
  done
done
