#!/bin/bash
# round 5, GPU session 13: the bench line as the driver runs it, timed
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
s=$(date +%s)
timeout -s KILL 900 python bench.py > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err
echo "bench.py wall $(( $(date +%s) - s )) s"
cut -c1-300 gpurun_out/r05_bench_final.json
