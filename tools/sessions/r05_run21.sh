#!/bin/bash
# round 5, GPU session 21: compute units reserved for the sweep's chain (side streams masked) - C5 wall-clock over 121 subproblems
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
out=gpurun_out/r05_lane_reserve.jsonl
: > $out
for g in 0 8 4 2 0 8 4 2; do
  OGSQP_LANE_RESERVE=$g timeout -s KILL 600 python tests/perf/solve_timing.py launch4 --sqp-core hip 2>/dev/null | tail -1 | sed "s/^{/{\"lane_reserve_every\": $g, /" >> $out
done
cut -c1-270 $out
OGSQP_LANE_RESERVE=4 tools/sqp_kstats.sh launch4 60 r05_reserve4 2>&1 | grep "^k_" | head -9 | cut -c1-160
