"""N > 1 path on CPU: two processes over gloo shard the FD columns, all-gather the padded
J_T slabs, and must reproduce the single-process Jacobian bit for bit (SURVEY.md section 8(e):
column sharding does not change any arithmetic).  The per-rank evaluator here is the CPU twin;
on the GPU box the same sharding helpers drive HipEngine (bench.py)."""
import os
import socket

import numpy as np
import pytest

from opengoddard_amd import sharding


def test_column_ranges_cover_everything_once():
    for n in (1, 7, 81, 201, 1442, 6148):
        for world in (1, 2, 3, 4, 8):
            b = sharding.block_rows(n, world)
            seen = np.zeros(n, dtype=int)
            for r in range(world):
                lo, hi = sharding.column_range(n, r, world)
                assert 0 <= lo <= hi <= n and hi - lo <= b
                seen[lo:hi] += 1
            assert np.all(seen == 1)
            assert sharding.gathered_shape(n, 5, world) == (b * world, 5)


def _worker(rank, world, port, name, out_dir):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from opengoddard_amd import _native, problems
        from oracle import np_path, twin
        prob, obj = problems.build(name)
        tw = twin.Twin(prob, obj)
        lb, ub = np_path.bounds_arrays(prob)
        x = np.clip(prob.p, lb, ub)
        h = _native.fd_step(x, lb, ub)
        lo, hi = sharding.column_range(tw.n, rank, world)
        local = torch.zeros((sharding.block_rows(tw.n, world), tw.m), dtype=torch.float64)
        if hi > lo:
            local[:hi - lo] = torch.from_numpy(tw.sweep(x, h, np.arange(lo, hi))[1])
        full = torch.empty(sharding.gathered_shape(tw.n, tw.m, world), dtype=torch.float64)
        sharding.all_gather_jt(local, full)
        np.save(os.path.join(out_dir, "jt_rank%d.npy" % rank), full.numpy()[:tw.n])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("brachistochrone", 2), ("goddard", 2)])
def test_two_rank_gloo_all_gather_reassembles_jacobian(name, world, tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, name, str(tmp_path)), nprocs=world, join=True)
    from opengoddard_amd import _native, problems
    from oracle import np_path, twin
    prob, obj = problems.build(name)
    tw = twin.Twin(prob, obj)
    lb, ub = np_path.bounds_arrays(prob)
    x = np.clip(prob.p, lb, ub)
    full = tw.sweep(x, _native.fd_step(x, lb, ub))[1]
    for r in range(world):
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "jt_rank%d.npy" % r)), full)
