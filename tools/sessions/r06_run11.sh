#!/bin/bash
# round 6, call 12: the flag-synchronised LQ panel (ring of 4 in aligned LDS): tests, panel trace, kernel statistics C3 / C4
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_slsqp_core.py -x -q -m gpu 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version\|^$" | tail -5
rm -f tools/_build/libogsqp_trace.so
bash tools/sqp_trace.sh polar_tsto 2 > /dev/null
grep "panel16 at" gpurun_out/sqp_trace_polar_tsto.log | head -3
bash tools/sqp_kstats.sh polar_tsto 10 r06_run11_sqp_polar_tsto 2>&1 | grep -v "Opened result" | head -14
bash tools/sqp_kstats.sh low_thrust 10 r06_run11_sqp_low_thrust 2>&1 | grep -v "Opened result" | head -10
