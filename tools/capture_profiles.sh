#!/bin/bash
# usage (on the GPU box, via gpurun): tools/capture_profiles.sh r01
# Runs bench.py for the four bench workloads, then rocprofv3 --kernel-trace --stats and separate
# --pmc passes (FETCH_SIZE, WRITE_SIZE, the MFMA counters) for three of them, the hardware probe
# and the SQP-core kernel statistics, all into gpurun_out/<round>/ (tools/summarize_profiles.py
# condenses that into profiles/<round>_*).
R=${GRAFT_REPO_ROOT:-/root/repo}
rnd=${1:-r01}
out=$R/gpurun_out/$rnd
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for w in polar_tsto goddard low_thrust launch4; do
    timeout 600 python $R/bench.py --workload $w 2>/dev/null | tail -1 > $out/bench_$w.json
done
for w in polar_tsto low_thrust launch4; do
    rm -rf $out/ktrace_$w
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ktrace_$w -o b -- \
        python $R/bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --sqp-iterations 0 --sqp-reference-iterations 0 > $out/ktrace_$w.log 2>&1
    for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf $out/pmc_${c}_$w
        timeout 600 rocprofv3 --pmc $c --output-format csv -d $out/pmc_${c}_$w -o b -- \
            python $R/bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --sqp-iterations 0 --sqp-reference-iterations 0 > $out/pmc_${c}_$w.log 2>&1
    done
    rm -rf $out/pmc_MFMA_$w
    timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv \
        -d $out/pmc_MFMA_$w -o b -- python $R/bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --sqp-iterations 0 --sqp-reference-iterations 0 > $out/pmc_MFMA_$w.log 2>&1
done
# the two-launch form of the default workload next to the fused one
w=polar_tsto
export OGPSX_SWEEP=split
rm -rf $out/ktrace_${w}_split
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ktrace_${w}_split -o b -- \
    python $R/bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --sqp-iterations 0 --sqp-reference-iterations 0 > $out/ktrace_${w}_split.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $out/pmc_${c}_${w}_split
    timeout 600 rocprofv3 --pmc $c --output-format csv -d $out/pmc_${c}_${w}_split -o b -- \
        python $R/bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --sqp-iterations 0 --sqp-reference-iterations 0 > $out/pmc_${c}_${w}_split.log 2>&1
done
timeout 600 python $R/bench.py --workload $w --no-cpu-baseline --sqp-iterations 0 2>/dev/null | tail -1 > $out/bench_${w}_split.json
unset OGPSX_SWEEP
# keep the merged directory small: only the csv summaries travel back
find $out -name "*.csv" -size +8M -delete
find $out -type f ! -name "*.csv" ! -name "*.json" ! -name "*.jsonl" ! -name "*.txt" ! -name "*.log" -delete
$R/tools/_build/gpu_probe > $out/gpu_probe.txt 2>&1
$R/tools/sqp_kstats.sh polar_tsto 8 sqp_polar_tsto > $out/sqp_kstats_polar_tsto.txt 2>&1
cp $R/gpurun_out/sqp_polar_tsto_kernel_stats.csv $out/ 2>/dev/null
du -sh $out
