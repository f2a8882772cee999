#!/bin/bash
# round 5, GPU session 4: wall-clock A/B of the C5 subproblem (first 25 major iterations, 121 subproblems)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
out=gpurun_out/r05_ab_launch4.jsonl
: > $out
run() { tag=$1; shift; env "$@" timeout 600 python tests/perf/solve_timing.py launch4 --sqp-core hip 2>/dev/null | tail -1 | sed "s/^{/{\"variant\": \"$tag\", /" >> $out; }
run default X=1
run inblock_off OGSQP_WIDE_INBLOCK=0
run rows_reg OGSQP_ROWS=reg
run both_off OGSQP_WIDE_INBLOCK=0 OGSQP_ROWS=reg
run default_again X=1
cut -c1-420 $out
