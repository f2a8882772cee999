#!/bin/bash
# usage (GPU box): tools/r04_final.sh - everything the round's profiles/ are made of, in one lease
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -12 > gpurun_out/r04_gputests.log
tail -7 gpurun_out/r04_gputests.log | head -3
bash tools/capture_profiles.sh r04 > /dev/null 2>&1
bash tools/solve_profiles.sh r04 2>&1 | tail -14 | cut -c1-300
bash tools/sqp_kstats.sh launch4 60 r04_sqp_launch4_60 2>&1 | tail -12
bash tools/sqp_kstats.sh polar_tsto 150 r04_sqp_polar_tsto_150 > /dev/null 2>&1
bash tools/sqp_kstats.sh polar_tsto 10 r04_sqp_polar_tsto > /dev/null 2>&1
bash tools/sqp_pmc.sh launch4 3 r04_sqp_launch4 > /dev/null 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r04_bench_driver_style.json
OG_BENCH_SAME_DEVICE=1 timeout 900 python bench.py --single-process --gpus 4 --workload low_thrust --steps 100 --reps 5 2>/dev/null | tail -1 > gpurun_out/r04_bench_single_process_x4_low_thrust.json
cut -c1-200 gpurun_out/r04_bench_driver_style.json
bash tools/wide_timeline.sh 3 > gpurun_out/r04_wide_timeline.txt 2>&1
bash tools/rows_hist.sh polar_tsto 10 > gpurun_out/r04_rows_hist_polar_tsto.txt 2>&1
for w in polar_tsto low_thrust launch4; do
  OG_EXTRA_HIPFLAGS=-DOGK_TRACE=1 OGPSX_TRACE=1 OGPSX_SWEEP=fused timeout 600 python tools/trace_fused.py $w 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_trace_fused_$w.txt
done
