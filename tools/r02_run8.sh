#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['config']['n'], 'us/step %.2f' % (1e3*d['ms_per_step']), 'kernel us %.2f' % (1e3*r['kernel_ms_mean']))"; }
for o in 0 1 2; do
for w in polar_tsto low_thrust launch4 goddard; do
  OG_EXTRA_HIPFLAGS="-DOGK_ORDER=$o" timeout 400 python bench.py --workload $w --quick 2>/dev/null | tail -1 | line "order=$o $w"
done
done
