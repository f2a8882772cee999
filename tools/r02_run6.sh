#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02f; mkdir -p $out
cd $R
for w in low_thrust launch4; do
  (OG_EXTRA_HIPFLAGS=-DOGK_TRACE=1 timeout 600 python tools/trace_fused.py $w) > $out/trace_$w.log 2>&1
done
cat $out/trace_low_thrust.log $out/trace_launch4.log
