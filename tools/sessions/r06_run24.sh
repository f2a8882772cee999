#!/bin/bash
# round 6, call 28: the QP core's tests after the lane-group panel got its own LDS size back (E = 7, 8)
timeout 1500 python -m pytest tests/test_slsqp_core.py -q -m gpu 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version\|^$" | tail -3
bash tools/sqp_kstats.sh low_thrust 10 r06_run24_sqp_low_thrust 2>&1 | grep "k_lq_step16<16\|k_lq_panel16" | head -4
