#!/bin/bash
# usage (GPU box): tools/fused_check.sh - parity of the sweep forms, then rocprofv3 launch averages of ogk_fused at C5/C4/C3
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_edge_problems.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for w in launch4 low_thrust polar_tsto; do bash tools/kstats.sh $w "$@" 2>&1 | grep fused; done
