#!/bin/bash
# Roofline fraction of ogk_sweep as a function of problem size (run on the GPU box).
# usage: tools/size_sweep.sh > gpurun_out/size_sweep.jsonl
R=${GRAFT_REPO_ROOT:-/root/repo}
for N in 16 32 48 64 96 128 160 200; do
  timeout 600 python $R/bench.py --workload launch4 --nodes $N,$N,$N,$N --steps 100 --warmup 10 --no-cpu-baseline --sqp-iterations 0 2>/dev/null | tail -1
done
