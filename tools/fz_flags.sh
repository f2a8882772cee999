#!/bin/bash
# usage (GPU box): tools/fz_flags.sh "<hipcc flags>" ["<flags>" ...] - ogk_fused built with extra flags, bench step time
R=${GRAFT_REPO_ROOT:-/root/repo}
export OGPSX_SWEEP=fused
for f in "$@"; do
    export OG_EXTRA_HIPFLAGS="$f"
    for w in ${FZ_WORKLOADS:-polar_tsto}; do
        timeout 300 python $R/bench.py --workload $w --no-cpu-baseline --sqp-iterations 0 2>/dev/null | tail -1 | \
            python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$f]', d['config']['n'], 'us/step %.2f' % (1e3*d['ms_per_step']))"
    done
done
