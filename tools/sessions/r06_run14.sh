#!/bin/bash
# round 6, call 16: where a C5 subproblem's wall-clock goes now: kernel statistics + the core's own timing split, 6 iterations
mkdir -p gpurun_out/r06
python tools/sqp_solve.py launch4 6 1e-6 hip 2>&1 | tail -2
bash tools/sqp_kstats.sh launch4 6 r06_run14_sqp_launch4 2>&1 | grep -v "Opened result" | head -18
