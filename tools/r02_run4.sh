#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02d; mkdir -p $out
cd $R
(timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $out/gputest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err
for w in goddard low_thrust launch4; do
  timeout 600 python bench.py --workload $w --sqp-iterations 0 --cpu-seconds 4 > $out/bench_$w.json 2> $out/bench_$w.err
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --force-collective --quick > $out/bench_force_collective.json 2> $out/bench_force_collective.err
cat $out/gputest.log; tail -c 600 $out/bench_default.err; for f in $out/bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("  unreadable", e); sys.exit()
r=d["roofline"]
print("  n", d["config"]["n"], "us/step %.2f [%.2f..%.2f]" % (1e3*d["ms_per_step"],1e3*d["timed_region"]["ms_per_step_min"],1e3*d["timed_region"]["ms_per_step_max"]), "evals/s %.3g"%d["value"], "kernel us %.2f"%(1e3*r["kernel_ms_mean"]), "frac %.3f"%r["frac"], "fill peak", r.get("measured_fill_peak_GBs"), "copy", r.get("measured_copy_peak_GBs"))
for k in ("self_check","dense_sweep_ms","host_api_ms_per_sweep","host_api_dense_transfer_ms_per_sweep","speedup_vs_cpu_baseline","speedup_vs_cpu_baseline_all_cores","speedup_vs_cpu_baseline_batch_last"):
    print("   ",k,d.get(k))
for k in ("cpu_baseline","cpu_baseline_all_cores","cpu_baseline_batch_last"):
    if k in d: print("   ",k,d[k].get("value"),d[k].get("cores"))
print("    parallelism", d["config"]["parallelism"])
PY
done
