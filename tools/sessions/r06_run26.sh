#!/bin/bash
# round 6, call 31: the same subproblems over and over on one handle: the same bits every time (the flag-synchronised panel
# hands data between wavefronts, the look-ahead launch between workgroups)
for w in "polar_tsto 2000" "low_thrust 1000" "goddard 1000"; do
  timeout 900 python tools/stress_sqp.py $w 2>&1 | tail -1 | cut -c1-250
done
