#!/bin/bash
# round 6, call 11: what makes the flag-synchronised LQ panel slow - variants of the panel kernel (trace builds), first
# panel of each subproblem
mkdir -p gpurun_out/r06 tools/_build
VARIANTS=${VARIANTS:-"flags: sleep8:-DP16_SLEEP=8 nodone:-DP16_NODONE barrier:-DP16_BARRIER sleep0:-DP16_SLEEP=0"}
NAMES=$(for v in $VARIANTS; do echo -n "${v%%:*} "; done)
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in $VARIANTS; do
  name=${v%%:*}; flag=${v#*:}; flag=${flag//,/ }
  lib=$R/tools/_build/libogsqp_trace_$name.so
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -DOGSQP_TRACE $flag \
      $R/opengoddard_amd/csrc/ogsqp.hip -o $lib 2>/dev/null &
done
wait
for name in $NAMES; do
  lib=$R/tools/_build/libogsqp_trace_$name.so
  OG_SQP_LIB=$lib timeout 300 python $R/tests/perf/solve_timing.py polar_tsto --sqp-core hip --max-restarts 1 --maxiter 2 > $R/gpurun_out/r06/trace_$name.log 2>&1
  echo "== $name"; grep "panel16 at\|its panel" $R/gpurun_out/r06/trace_$name.log | sed 's/.*its panel/   head: its panel/' | cut -c1-230 | head -4
done
