#!/bin/bash
# usage (GPU box): tools/recapture_ktrace.sh <round> <workload> [times] - the kernel-trace pass of tools/capture_profiles.sh again for
# one workload (the average of ogk_fused moves by +-0.5 us from box to box and run to run: every sample is printed, the
# last run's table is what gpurun_out/<round>/ktrace_<workload> holds)
R=${GRAFT_REPO_ROOT:-/root/repo}
rnd=$1; w=$2; times=${3:-3}
out=$R/gpurun_out/$rnd
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
Q="--quick --no-cpu-baseline --sqp-iterations 0"
for i in $(seq $times); do
    rm -rf $out/ktrace_$w
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ktrace_$w -o b -- \
        python $R/bench.py --workload $w --steps 200 --warmup 20 --reps ${REPS:-25} $Q > $out/ktrace_$w.log 2>&1
    python - $out/ktrace_$w <<'PY'
import csv, glob, sys
for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/*kernel_stats.csv")[0])):
    if "ogk_fused" in r["Name"]: print("ogk_fused calls %s avg %.2f us min %.2f max %.2f" % (r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
find $out -name "*.csv" -size +8M -delete
find $out -type f ! -name "*.csv" ! -name "*.json" ! -name "*.jsonl" ! -name "*.txt" ! -name "*.log" -delete
