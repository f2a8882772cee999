#!/bin/bash
# usage (GPU box): tools/fz_exp.sh mask [mask ...] - ogk_fused with pieces removed (timing only, results wrong)
R=${GRAFT_REPO_ROOT:-/root/repo}
export OGPSX_SWEEP=fused
for mask in "$@"; do
    export OG_EXTRA_HIPFLAGS=-DOGK_FZ=$mask
    for w in ${FZ_WORKLOADS:-polar_tsto}; do
        timeout 300 python $R/bench.py --workload $w --no-cpu-baseline --sqp-iterations 0 2>/dev/null | tail -1 | \
            python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('mask $mask', d['config']['n'], 'us/step %.2f' % (1e3*d['ms_per_step']))"
    done
done
