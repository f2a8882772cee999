#!/bin/bash
# round 5, GPU session 11: the short-row pass, this round's form against rounds 3-4's (C3 and C4)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for v in default r4 default r4; do
  OGSQP_ROWS=$v tools/sqp_kstats.sh polar_tsto 10 r05_rows8_$v 2>&1 | grep "k_rows\|^polar" | cut -c1-200
done
for v in default r4; do
  OGSQP_ROWS=$v tools/sqp_kstats.sh low_thrust 25 r05_rows16_$v 2>&1 | grep "k_rows\|^low" | cut -c1-200
done
