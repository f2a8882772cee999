#!/bin/bash
# usage (GPU box): tools/wide_timeline.sh [maxiter] - kernel trace of a short C5 solve; per-queue busy time and the kernels of one
# 64-reflector block of the wide sweep in the middle of the last subproblem, with their start offsets (who waits for whom)
R=${GRAFT_REPO_ROOT:-/root/repo}
it=${1:-3}
out=/tmp/wide_tl
rm -rf $out
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $out -o b -- python $R/tools/sqp_solve.py launch4 $it 1e-6 hip > $out.log 2>&1 )
python - $out <<'PY'
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
def nm(r):
    m = re.search(r"((k_\w+|ogk_\w+|Cijk_\w{0,24}|__amd\w+)(<[\d, ]+>)?)", r["Kernel_Name"]); return m.group(1) if m else r["Kernel_Name"][:30]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm(r), r.get("Queue_Id", "?")) for r in rows))
# the last subproblem's wide sweep: from the last k_gemm_tn-before-first-panel ... find panel launches
pan = [i for i, e in enumerate(ev) if e[2].startswith("k_lq_panel16_wide")]
# sweeps = runs of panels separated by > 5 ms
starts = [pan[0]] + [pan[i] for i in range(1, len(pan)) if ev[pan[i]][0] - ev[pan[i-1]][0] > 3_000_000]
last = starts[-1]
t0 = ev[last][0]
# end of sweep: the last wide panel of that run
lastpan = [i for i in pan if i >= last][-1]
t1 = max(e[1] for e in ev[last:lastpan + 40])
sw = [e for e in ev if e[0] >= t0 and e[0] <= t1]
print("last sweep: %.2f ms, %d kernels" % ((t1 - t0) / 1e6, len(sw)))
by_q = collections.defaultdict(float)
for s, e, n, q in sw: by_q[q] += (e - s) / 1e6
print("busy ms per queue:", dict(by_q))
by_n = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, q in sw: by_n[(q, n)][0] += 1; by_n[(q, n)][1] += (e - s) / 1e3
for (q, n), (c, t) in sorted(by_n.items(), key=lambda kv: -kv[1][1])[:24]:
    print("  queue %-3s %-28s calls %5d total %8.2f ms avg %7.1f us" % (q, n, c, t / 1e3, t / c))
# one block in the middle: between the 4k-th and 4(k+1)-th panels
pp = [e for e in sw if e[2].startswith("k_lq_panel16_wide")]
k = (len(pp) // 8) * 4
b0, b1 = pp[k][0], pp[k + 4][0]
print("block %d of %d: %.1f us between its first panel and the next block's" % (k // 4, len(pp) // 4, (b1 - b0) / 1e3))
for s, e, n, q in sw:
    if b0 <= s < b1: print("   +%7.1f us  %7.1f us  q%-3s %s" % ((s - b0) / 1e3, (e - s) / 1e3, q, n))
PY
