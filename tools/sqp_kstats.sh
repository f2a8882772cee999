#!/bin/bash
# usage: tools/sqp_kstats.sh <problem> <maxiter> [tag]   (run on the GPU box)
# rocprofv3 kernel-trace statistics of one Problem.solve(sqp_core="hip") run -> gpurun_out/<tag>_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
p=$1; it=$2; tag=${3:-sqp_$p}
out=/tmp/sqpk_$tag
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o b -- python $R/tools/sqp_solve.py $p $it 1e-6 hip > $out.log 2>&1 )
grep -v amdgpu.ids $out.log | grep -v rocprofv3 | tail -3
f=$(ls $out/*kernel_stats.csv 2>/dev/null | head -1)
mkdir -p $R/gpurun_out
[ -n "$f" ] && cp $f $R/gpurun_out/${tag}_kernel_stats.csv && python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    import re
    nm=re.search(r"((k_\w+|ogk_\w+|__amd\w+)(<[\d, ]+>)?)", r["Name"]); nm=nm.group(1) if nm else r["Name"][:34]
    print("%-22s calls %7s total %9.2f ms avg %9.2f us  %5.1f%%"%(nm[:22], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
