#!/bin/bash
# usage: tools/kstats.sh <workload> [extra env assignments...]   (run on the GPU box)
# prints rocprofv3 kernel-trace average durations of the two engine kernels for one bench run
R=${GRAFT_REPO_ROOT:-/root/repo}
w=$1; shift
tag=$(echo "$w $*" | tr ' =' '__')
out=/tmp/kstats_$tag
( cd /tmp && export TMPDIR=/tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o b -- python $R/bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline > $out.log 2>&1 )
python - "$out" "$tag" <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+"/*kernel_stats.csv")
if not f: print(sys.argv[2],"NO STATS"); sys.exit()
for r in csv.DictReader(open(f[0])):
    if "ogk_" in r["Name"]: print("%-40s %-28s calls %s avg %.2f us min %.2f"%(sys.argv[2], r["Name"].split("::")[-1][:28], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
