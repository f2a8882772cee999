#!/bin/bash
# round 5, GPU session 16: the timed loop with its steps bound once against the per-step Python path, same lease, alternating
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
: > gpurun_out/r05_host_ab.txt
for rep in 1 2 3; do
  for v in bound unbound; do
    if [ $v = unbound ]; then export OG_BENCH_UNBOUND=1; else unset OG_BENCH_UNBOUND; fi
    for st in 200 20; do
      python bench.py --quick --steps $st --warmup 20 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); t = d['timed_region']
print('$v steps $st: us/step median %.3f min %.3f max %.3f | kernel mean %.3f' % (1e3*t['ms_per_step_median'], 1e3*t['ms_per_step_min'], 1e3*t['ms_per_step_max'], 1e3*d['roofline']['kernel_ms_mean']))" >> gpurun_out/r05_host_ab.txt
    done
  done
done
cat gpurun_out/r05_host_ab.txt
