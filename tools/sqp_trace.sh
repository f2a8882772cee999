#!/bin/bash
# usage (GPU box): tools/sqp_trace.sh [workload] [major iterations] - in-kernel section times of the SQP core's active-set
# kernels (libogsqp.so rebuilt with -DOGSQP_TRACE into tools/_build/, loaded through OG_SQP_LIB) over the first
# major iterations of a solve; output gpurun_out/sqp_trace_<workload>.log
R=${GRAFT_REPO_ROOT:-/root/repo}
w=${1:-polar_tsto}; its=${2:-10}
mkdir -p $R/gpurun_out $R/tools/_build
lib=$R/tools/_build/libogsqp_trace.so
[ -f $lib ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -DOGSQP_TRACE \
    $R/opengoddard_amd/csrc/ogsqp.hip -o $lib || exit 2
OG_SQP_LIB=$lib timeout 600 python $R/tests/perf/solve_timing.py $w --sqp-core hip --max-restarts 1 --maxiter $its \
    > $R/gpurun_out/sqp_trace_$w.log 2>&1
grep -c "ogsqp trace" $R/gpurun_out/sqp_trace_$w.log
