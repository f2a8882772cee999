#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02g; mkdir -p $out
cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'us/step %.2f' % (1e3*d['ms_per_step']), 'local %.2f' % (1e3*d['ms_per_step_without_collective']), d['config']['parallelism'])"; }
for w in polar_tsto launch4; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 50 --warmup 5 --force-collective --quick --workload $w 2>$out/direct_$w.err | tail -1 | line "direct $w"
OG_BENCH_TORCH_ALLGATHER=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 50 --warmup 5 --force-collective --quick --workload $w 2>$out/torch_$w.err | tail -1 | line "torch $w"
done
tail -3 $out/direct_polar_tsto.err
