#!/bin/bash
# Step time of the one-launch sweep as a function of problem size: the 4-phase / 8-state / 4-control problem with N
# nodes per phase (run on the GPU box; the kernels of the extra sizes are compiled there).
# usage: tools/size_sweep.sh > gpurun_out/size_sweep.jsonl
R=${GRAFT_REPO_ROOT:-/root/repo}
for N in 16 32 64 96 128 160 200; do
  timeout 900 python $R/bench.py --workload launch4 --nodes $N,$N,$N,$N --steps 100 --warmup 10 --quick 2>/dev/null | grep '^{"metric' | tail -1
done
