#!/bin/bash
# usage (GPU box): tools/fused_ab.sh [workload ...] - the bench step with the fused launch and with two launches
R=${GRAFT_REPO_ROOT:-/root/repo}
for m in fused split; do
    if [ $m = split ]; then export OGPSX_SWEEP=split; fi
    for w in ${@:-polar_tsto launch4}; do
        timeout 300 python $R/bench.py --workload $w --no-cpu-baseline --sqp-iterations 0 2>/dev/null | tail -1 | \
            python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$m', d['config']['n'], 'us/step %.2f' % (1e3*d['ms_per_step']), 'evals/s %.3g' % d['value'], r['kernel'], 'kernel us %.2f' % (1e3*r['kernel_ms_mean']), 'frac %.3f' % r['frac'], 'split eval %.2f sweep %.2f' % (1e3*r['split_eval_kernel_ms_mean'], 1e3*r['split_sweep_kernel_ms_mean']))"
    done
done
