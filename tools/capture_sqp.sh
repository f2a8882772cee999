#!/bin/bash
# usage (GPU box): tools/capture_sqp.sh <round> - the part of a round's profiles that follows the SQP core: the bench
# lines (their SQP leg), the driver-style line, solve timings and cold start (tools/solve_profiles.sh) and the
# rocprofv3 kernel statistics of the core over the first 10 and 150 major iterations and over a run to exit mode 0
R=${GRAFT_REPO_ROOT:-/root/repo}
rnd=${1:-r03}
out=$R/gpurun_out/$rnd
mkdir -p $out
for w in polar_tsto goddard low_thrust launch4; do
    timeout 900 python $R/bench.py --workload $w --cpu-seconds 6 2>/dev/null | tail -1 > $out/bench_$w.json
done
timeout 900 python $R/bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $out/bench_driver_style.json
$R/tools/solve_profiles.sh $rnd > /dev/null 2>&1
$R/tools/sqp_kstats.sh polar_tsto 10 ${rnd}_sqp_polar_tsto > $out/sqp_kstats_10.txt 2>&1
$R/tools/sqp_kstats.sh polar_tsto 150 ${rnd}_sqp_polar_tsto_150 > $out/sqp_kstats_150.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sqpk_conv -o b -- \
    python $R/tools/sqp_converge.py polar_tsto 400 > $out/sqp_converge.txt 2>&1 )
f=$(ls /tmp/sqpk_conv/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/${rnd}_sqp_polar_tsto_converge_kernel_stats.csv
grep -v amdgpu.ids $out/sqp_converge.txt | tail -2
cat $out/sqp_kstats_150.txt | tail -14
