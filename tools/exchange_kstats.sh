#!/bin/bash
# usage (GPU box): tools/exchange_kstats.sh [workload] - rocprofv3 kernel statistics of the sharded step with the
# exchange path forced on one rank (sweep, pack, ncclAllGather, unpack)
R=${GRAFT_REPO_ROOT:-/root/repo}
w=${1:-polar_tsto}
out=/tmp/exch_$w
cd /tmp && export TMPDIR=/tmp
MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o b -- \
    python $R/bench.py --gpus 1 --workload $w --steps 200 --warmup 20 --reps 5 --force-collective --quick > $out.log 2>&1
f=$(ls $out/*kernel_stats.csv 2>/dev/null | head -1)
python - "$f" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]:
    nm=re.sub(r"\(anonymous namespace\)::","",r["Name"])[:60]
    print("%-60s calls %6s avg %8.2f us min %8.2f"%(nm, r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
