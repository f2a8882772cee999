#!/bin/bash
# round 5, GPU session 10: bench.py's N = 4 and N = 8 paths as dry runs on the one device (gloo, host-staged exchange), then the suite
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for n in 4 8; do
  OG_BENCH_SAME_DEVICE=1 timeout -s KILL 900 python bench.py --gpus $n --steps 20 --warmup 5 --reps 5 > gpurun_out/r05_bench_${n}ranks_same_device.json 2> gpurun_out/r05_bench_${n}ranks_same_device.err
  echo "N=$n rc $?"; tail -c 1500 gpurun_out/r05_bench_${n}ranks_same_device.json | tr '\n' ' ' | cut -c1-1500; echo
  grep -v "amdgpu.ids\|^$" gpurun_out/r05_bench_${n}ranks_same_device.err | tail -5
done
timeout -s KILL 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=6 2>&1 | tail -14 > gpurun_out/r05_gputests.log
cat gpurun_out/r05_gputests.log | cut -c1-200
