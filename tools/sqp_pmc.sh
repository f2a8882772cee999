#!/bin/bash
# usage (GPU box): tools/sqp_pmc.sh <problem> <maxiter> [tag] - HBM traffic of the SQP core's kernels: separate rocprofv3
# --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with a trace domain) over one Problem.solve(sqp_core="hip") run,
# per-kernel averages into gpurun_out/<tag>_pmc.txt (FETCH_SIZE counts 32-byte... see MI355X_MICROARCH.md: KB units,
# x2 on gfx950 for the read side as in tools/summarize_profiles.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
p=$1; it=$2; tag=${3:-sqp_$p}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/sqp_pmc_$c
    timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/sqp_pmc_$c -o b -- python $R/tools/sqp_solve.py $p $it 1e-6 hip > /tmp/sqp_pmc_$c.log 2>&1
done
python - "$R/gpurun_out/${tag}_pmc.txt" <<'PY'
import csv, glob, re, sys, collections
out = open(sys.argv[1], "w")
agg = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/sqp_pmc_%s/*counter_collection.csv" % c)
    if not f:
        out.write("no counter file for %s\n" % c); continue
    for r in csv.DictReader(open(f[0])):
        nm = re.search(r"((k_\w+|ogk_\w+|Cijk_\w{0,40})(<[\d, ]+>)?)", r["Kernel_Name"])
        if not nm: continue
        a = agg.setdefault(nm.group(1), {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
        a[r["Counter_Name"]][0] += float(r["Counter_Value"]); a[r["Counter_Name"]][1] += 1
out.write("# per launch: FETCH_SIZE and WRITE_SIZE in KB as counted; fabric traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 note of the guide)\n")
for k, a in sorted(agg.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"][0] + kv[1]["WRITE_SIZE"][0])):
    fn, wn = max(a["FETCH_SIZE"][1], 1), max(a["WRITE_SIZE"][1], 1)
    f, w = a["FETCH_SIZE"][0] / fn, a["WRITE_SIZE"][0] / wn
    out.write("%-24s launches %6d  FETCH_SIZE %10.1f KB  WRITE_SIZE %10.1f KB  -> %8.2f MB per launch\n" % (k, fn, f, w, (2 * f + w) / 1024))
out.close()
print(open(sys.argv[1]).read())
PY
