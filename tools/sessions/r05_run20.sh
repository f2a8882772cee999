#!/bin/bash
# round 5, GPU session 20: more column slices for k_wy_w (fuller SIMDs on the side streams) - C5 wall-clock over 121 subproblems
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
out=gpurun_out/r05_wyw_grid.jsonl
: > $out
for g in 1024 2048 4096 1024 2048 4096; do
  OGSQP_WYW_GRID=$g timeout -s KILL 600 python tests/perf/solve_timing.py launch4 --sqp-core hip 2>/dev/null | tail -1 | sed "s/^{/{\"wyw_grid\": $g, /" >> $out
done
cut -c1-260 $out
OGSQP_WYW_GRID=4096 tools/sqp_kstats.sh launch4 60 r05_wyw4096 2>&1 | grep "^k_" | head -8 | cut -c1-160
