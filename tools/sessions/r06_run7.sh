#!/bin/bash
# round 6, call 7: the N > 1 dry runs on the one device (declared as such on the line), N = 2, 4, 8
mkdir -p gpurun_out/r06
export OG_BENCH_SAME_DEVICE=1
for n in 2 4 8; do
  timeout 600 python bench.py --gpus $n --steps 20 --warmup 5 --reps 5 2>gpurun_out/r06/ranks$n.err | tail -1 > gpurun_out/r06/bench_${n}ranks_same_device.json
  python -c "
import json; r=json.load(open('gpurun_out/r06/bench_${n}ranks_same_device.json')); print($n, {k:r.get(k) for k in ('n_gpus','distinct_devices','ranks_per_device','rccl_ranks','dry_run','invalid','ms_per_step','ms_per_step_without_collective','message_bytes_per_rank')})"
done
unset OG_BENCH_SAME_DEVICE
# and what an N > 1 command does on a box that has ONE GPU without the dry-run declaration: it must fail loudly
timeout 300 python bench.py --gpus 2 --steps 5 --warmup 1 --reps 1 > gpurun_out/r06/two_ranks_one_gpu.log 2>&1; echo "two ranks, one GPU, no dry-run flag: exit code $?"
tail -3 gpurun_out/r06/two_ranks_one_gpu.log | cut -c1-300
