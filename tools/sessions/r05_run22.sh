#!/bin/bash
# round 5, GPU session 22: the final tree - whole GPU suite, smoke(), the bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -s KILL 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=6 2>&1 | tail -12 > gpurun_out/r05_gputests.log
cat gpurun_out/r05_gputests.log | cut -c1-200
timeout -s KILL 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-300
s=$(date +%s)
timeout -s KILL 900 python bench.py > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err
echo "bench.py wall $(( $(date +%s) - s )) s"
cut -c1-260 gpurun_out/r05_bench_final.json
