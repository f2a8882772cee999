#!/bin/bash
# round 6, call 9-10: LQ panel of 16 with a row pair per wavefront (9: one barrier per reflector; 10: flags, no barrier)
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_slsqp_core.py -x -q -m gpu 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version\|^$" | tail -5
rm -f tools/_build/libogsqp_trace.so
bash tools/sqp_trace.sh polar_tsto 3 > /dev/null
grep "panel16\|its panel" gpurun_out/sqp_trace_polar_tsto.log | head -8
bash tools/sqp_kstats.sh polar_tsto 10 r06_run9_sqp_polar_tsto | head -14
bash tools/sqp_kstats.sh low_thrust 10 r06_run9_sqp_low_thrust | head -8
