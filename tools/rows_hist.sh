#!/bin/bash
# usage (GPU box): tools/rows_hist.sh [problem] [maxiter] - durations of the active-set kernels over the first iterations of a solve
R=${GRAFT_REPO_ROOT:-/root/repo}
p=${1:-polar_tsto}; it=${2:-10}
out=/tmp/rows_hist
rm -rf $out
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $out -o b -- python $R/tools/sqp_solve.py $p $it 1e-6 hip > $out.log 2>&1 )
python - $out <<'PY'
import csv, glob, sys, numpy as np
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
for key in ("k_rows_decide", "k_rows_apply"):
    d = np.array([(e - s) / 1e3 for s, e, n in ev if key in n])
    print(key, "calls", len(d), "total %.1f ms" % (d.sum() / 1e3), "mean %.1f" % d.mean(), "median %.1f" % np.median(d))
    edges = [0, 2, 5, 8, 12, 16, 20, 30, 50, 80, 120, 200]
    h, _ = np.histogram(d, edges)
    for lo, hi, c in zip(edges[:-1], edges[1:], h):
        sel = d[(d >= lo) & (d < hi)]
        print("   %4d-%4d us: %5d calls, %7.2f ms" % (lo, hi, c, sel.sum() / 1e3))
# gaps between consecutive active-set kernels
seq = [(s, e, n) for s, e, n in ev if "k_rows_" in n]
gaps = np.array([(seq[i + 1][0] - seq[i][1]) / 1e3 for i in range(len(seq) - 1) if seq[i + 1][0] - seq[i][1] < 50_000])
print("gaps between consecutive k_rows_* kernels: mean %.2f us, median %.2f, total %.1f ms" % (gaps.mean(), np.median(gaps), gaps.sum() / 1e3))
# duration of decide against its position in the subproblem (active-set size grows)
dd = [(s, (e - s) / 1e3) for s, e, n in ev if "k_rows_decide" in n]
first = dd[:600]
for i in range(0, len(first), 60):
    seg = np.array([x[1] for x in first[i:i + 60]])
    print("   decide calls %3d-%3d: mean %.1f us max %.1f" % (i, i + 59, seg.mean(), seg.max()))
PY
