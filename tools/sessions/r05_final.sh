#!/bin/bash
# usage (GPU box): tools/r05_final.sh - everything the round's profiles/ are made of, in one lease
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout -s KILL 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 2>&1 | tail -24 > gpurun_out/r05_gputests.log
tail -4 gpurun_out/r05_gputests.log
bash tools/capture_profiles.sh r05 > /dev/null 2>&1
bash tools/solve_profiles.sh r05 2>&1 | tail -14 | cut -c1-300
bash tools/sqp_kstats.sh launch4 60 r05_sqp_launch4_60 2>&1 | tail -12
bash tools/sqp_kstats.sh polar_tsto 150 r05_sqp_polar_tsto_150 > /dev/null 2>&1
bash tools/sqp_kstats.sh polar_tsto 10 r05_sqp_polar_tsto > /dev/null 2>&1
bash tools/sqp_pmc.sh launch4 3 r05_sqp_launch4 > /dev/null 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r05_bench_driver_style.json
OG_BENCH_SAME_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 50 --reps 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_bench_2ranks_same_device.json
for w in low_thrust_r1 launch4_r1; do
  timeout 600 python bench.py --workload $w --quick 2>/dev/null | tail -1 > gpurun_out/r05_bench_$w.json
done
cut -c1-200 gpurun_out/r05_bench_driver_style.json
for w in polar_tsto low_thrust launch4; do
  OG_MODULE_HIPFLAGS=-DOGK_TRACE=1 OGPSX_TRACE=1 OGPSX_SWEEP=fused timeout 600 python tools/trace_fused.py $w 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_trace_fused_$w.txt
done
( timeout 300 python tools/stress_sqp.py polar_tsto 100; timeout 300 python tools/stress_sqp.py launch4 12 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_stress_sqp.txt
timeout 300 python tools/stress_determinism.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_stress_determinism.txt
tail -3 gpurun_out/r05_stress_sqp.txt gpurun_out/r05_stress_determinism.txt
