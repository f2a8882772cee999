#!/bin/bash
# usage (on the GPU box, via gpurun): tools/capture_profiles.sh r02
# bench.py for the bench workloads, then rocprofv3 --kernel-trace --stats and separate --pmc passes (FETCH_SIZE,
# WRITE_SIZE, the MFMA counters) of the same command, all into gpurun_out/<round>/ (tools/summarize_profiles.py
# condenses that into profiles/<round>_*).  --pmc is never combined with a trace domain other than kernel-trace.
R=${GRAFT_REPO_ROOT:-/root/repo}
rnd=${1:-r02}
out=$R/gpurun_out/$rnd
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for w in polar_tsto goddard low_thrust launch4; do
    timeout 1200 python $R/bench.py --workload $w --cpu-seconds 6 --solve-starts 1 $([ $w = launch4 ] && echo --no-solve) 2>/dev/null | tail -1 > $out/bench_$w.json
done
Q="--quick --no-cpu-baseline --sqp-iterations 0"
for w in polar_tsto low_thrust launch4; do
    rm -rf $out/ktrace_$w
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ktrace_$w -o b -- \
        python $R/bench.py --workload $w --steps 200 --warmup 20 --reps 25 $Q > $out/ktrace_$w.log 2>&1
    for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf $out/pmc_${c}_$w
        timeout 600 rocprofv3 --pmc $c --output-format csv -d $out/pmc_${c}_$w -o b -- \
            python $R/bench.py --workload $w --steps 50 --warmup 5 --reps 2 $Q > $out/pmc_${c}_$w.log 2>&1
    done
    rm -rf $out/pmc_MFMA_$w
    timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv \
        -d $out/pmc_MFMA_$w -o b -- python $R/bench.py --workload $w --steps 50 --warmup 5 --reps 2 $Q > $out/pmc_MFMA_$w.log 2>&1
done
# the two-launch form of the default workload next to the one-launch form
w=polar_tsto
export OGPSX_SWEEP=split
rm -rf $out/ktrace_${w}_split
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ktrace_${w}_split -o b -- \
    python $R/bench.py --workload $w --steps 200 --warmup 20 --reps 5 $Q > $out/ktrace_${w}_split.log 2>&1
timeout 600 python $R/bench.py --workload $w --quick 2>/dev/null | tail -1 > $out/bench_${w}_split.json
unset OGPSX_SWEEP
# keep the merged directory small: only the csv summaries travel back
find $out -name "*.csv" -size +8M -delete
find $out -type f ! -name "*.csv" ! -name "*.json" ! -name "*.jsonl" ! -name "*.txt" ! -name "*.log" -delete
$R/tools/_build/gpu_probe > $out/gpu_probe.txt 2>&1
# columns per light workgroup on the small configurations
for w in goddard polar_tsto_shipped low_thrust_shipped brachistochrone; do
  for c in 4 8 16; do
    OG_FUSED_COLS=$c timeout 300 python $R/bench.py --workload $w --quick 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w cols $c n', d['config']['n'], 'us/step %.2f' % (1e3*d['ms_per_step']))"
  done
done > $out/cols_small.txt 2>&1
du -sh $out
