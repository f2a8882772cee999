"""N > 1 path on CPU: processes over gloo run :class:`opengoddard_amd.sharding.ShardedSweep` - the class
``bench.py --gpus N`` runs on the GPUs - with a backend made of the CPU twin and the tracer's pattern in
place of :class:`~opengoddard_amd.sharding.HipBackend`: every rank sweeps its block of FD columns into its
rows of a zero-initialised replica, packs the static non-zeros, ONE all-gather of equal messages exchanges
them, every rank scatters the others' entries.  The replica of every rank must equal the single-process
Jacobian bit for bit (column sharding does not change any arithmetic), also through
a point where F(x0) has non-finite rows (NaN in those rows of EVERY column, of every rank's block) and back."""
import os
import socket

import numpy as np
import pytest

from opengoddard_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def test_plan_lays_out_equal_messages():
    rng = np.random.default_rng(3)
    for n in (1, 5, 64, 201):
        counts = rng.integers(0, 40, n)
        indptr = np.concatenate([[0], np.cumsum(counts)])
        for world in (1, 2, 3, 8, n + 2):
            B, block_vals, off = sharding.plan(indptr, world)
            assert B == sharding.block_rows(n, world)
            used = np.zeros(block_vals * world, dtype=int)
            for j in range(n):
                r = j // B
                assert r * block_vals <= off[j] and off[j] + counts[j] <= (r + 1) * block_vals
                used[off[j]:off[j] + counts[j]] += 1
            assert used.max(initial=0) <= 1                      # no two columns share a slot
            lo = [sharding.column_range(n, r, world) for r in range(world)]
            assert block_vals == max(indptr[b] - indptr[a] for a, b in lo)


class TwinBackend:
    """CPU stand-in for sharding.HipBackend (test infrastructure): the twin evaluates a block of columns, the
    pattern comes from the tracer, fill / pack / unpack are the NumPy statement of the kernels' contract."""

    def __init__(self, prob, obj):
        import torch
        from opengoddard_amd import codegen
        from oracle import twin
        self.torch = torch
        self.tw = twin.Twin(prob, obj)
        self.indptr, self.rows = codegen.sparsity(self.tw.program)
        self.n, self.m = self.tw.n, self.tw.m
        self.dirty = False               # the previous step left a NaN fill in the replica
        self.bad_now = False

    def pattern_indptr(self):
        return self.indptr

    def zeros(self, *shape):
        return self.torch.zeros(shape, dtype=self.torch.float64)

    def empty(self, *shape):
        return self.torch.empty(shape, dtype=self.torch.float64)

    def upload(self, v):
        return self.torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64))

    def make_plan(self, world):
        self.world = world
        self.B, self.block_vals, self.off = sharding.plan(self.indptr, world)
        return self.B, self.block_vals

    def register_block(self, replica, lo, hi):
        pass                             # the replica starts as zeros

    def sweep(self, x, h, lo, hi, replica, F0):
        F, JT = self.tw.sweep(x.numpy(), h.numpy(), np.arange(lo, hi))
        F0.numpy()[:] = F
        self.z = F - F
        self.bad_now = bool(np.isnan(self.z).any())
        block = replica.numpy()[lo:hi]
        if self.bad_now or self.dirty:   # what the registered buffer's state word triggers on the device
            block[:] = self.z[None, :]
        for j in range(lo, hi):          # persistent-zero output: only the pattern positions are written
            r = self.rows[self.indptr[j]:self.indptr[j + 1]]
            block[j - lo, r] = JT[j - lo, r]
        assert np.array_equal(block, JT, equal_nan=True)

    def pack(self, rank, lo, hi, replica, send):
        out, rep = send.numpy(), replica.numpy()
        for j in range(lo, hi):
            r = self.rows[self.indptr[j]:self.indptr[j + 1]]
            o = self.off[j] - rank * self.block_vals
            out[o:o + r.size] = rep[j, r]

    def unpack(self, rank, recv, replica):
        buf, rep = recv.numpy(), replica.numpy()
        lo, hi = sharding.column_range(self.n, rank, self.world)
        for j in list(range(0, lo)) + list(range(hi, self.n)):
            if self.bad_now or self.dirty:
                rep[j, :] = self.z
            r = self.rows[self.indptr[j]:self.indptr[j + 1]]
            rep[j, r] = buf[self.off[j]:self.off[j] + r.size]
        self.dirty = self.bad_now


def _points(prob):
    from oracle import np_path
    lb, ub = np_path.bounds_arrays(prob)
    x_ok = np.clip(prob.p, lb, ub)
    x_bad = x_ok.copy()
    x_bad[prob.index_states(2, 0, 7)] = 0.0              # mass = 0 at one node -> division by zero
    rng = np.random.default_rng(11)
    x_other = np.clip(x_ok + 1e-3 * rng.standard_normal(x_ok.size), lb, ub)
    return lb, ub, [x_ok, x_bad, x_other]


def _worker(rank, world, port, name, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from opengoddard_amd import _native, problems
        prob, obj = problems.build(name)
        be = TwinBackend(prob, obj)
        sh = sharding.ShardedSweep(be, be.n, be.m, rank, world)
        lb, ub, points = _points(prob)
        for k, x in enumerate(points):
            h = _native.fd_step(x, lb, ub)
            sh.step(be.upload(x), be.upload(h))
            np.save(os.path.join(out_dir, "jt_rank%d_point%d.npy" % (rank, k)), sh.replica.numpy())
            np.save(os.path.join(out_dir, "f_rank%d_point%d.npy" % (rank, k)), sh.F0.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("goddard", 2), ("goddard", 3)])
def test_gloo_ranks_reassemble_the_jacobian(name, world, tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, name, str(tmp_path)), nprocs=world, join=True)
    from opengoddard_amd import _native, problems
    from oracle import twin
    prob, obj = problems.build(name)
    tw = twin.Twin(prob, obj)
    lb, ub, points = _points(prob)
    saw_nan = False
    for k, x in enumerate(points):
        F, full = tw.sweep(x, _native.fd_step(x, lb, ub))
        saw_nan = saw_nan or bool(np.isnan(full).any())
        for r in range(world):
            got = np.load(os.path.join(str(tmp_path), "jt_rank%d_point%d.npy" % (r, k)))
            assert np.array_equal(got, full, equal_nan=True), "rank %d point %d" % (r, k)
            assert np.array_equal(np.load(os.path.join(str(tmp_path), "f_rank%d_point%d.npy" % (r, k))), F,
                                  equal_nan=True)
    assert saw_nan


def test_bench_launches_its_own_ranks_when_there_is_no_launcher(monkeypatch):
    """``python bench.py --gpus N`` without WORLD_SIZE in the environment (how the driver starts N = 1, and what a
    user types for N > 1): bench.py starts N ranks of itself under torch.distributed.run on 127.0.0.1 with its own
    arguments, and hands the exit code on.  With WORLD_SIZE set (a launcher is there) it does not."""
    import importlib
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = list(cmd), env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "1"])
    assert bench.main() == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--standalone" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "4", "--steps", "5", "--warmup", "1"]
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0" or "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ
    # under a launcher: no second launch (it fails later, for lack of a GPU here, not by launching again)
    seen.clear()
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(SystemExit):
        bench.main()
    assert not seen


def test_bench_counts_physical_devices_and_refuses_a_line_that_is_not_n_gpus_over_rccl():
    """VERDICT r5 #4 / ADVICE r5: ``distinct_devices`` keys on the physical GPU (host + uuid / PCI address), so N ranks
    piled onto one device report 1 - not N as the (ordinal, pid) pairs of round 5 did - and an N > 1 line stands only
    if every rank has its own GPU AND RCCL's communicator has N ranks; the declared dry run is exempt and labelled."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    same = [{"rank": r, "pid": 100 + r, "device": 0, "physical_device": "box/uuid:GPU-aa|pci:0000:05:00"} for r in range(4)]
    census = bench.device_census(same)
    assert census["distinct_devices"] == 1 and census["ranks_per_device"] == {"box/uuid:GPU-aa|pci:0000:05:00": [0, 1, 2, 3]}
    line = dict(census, rccl_ranks=None)
    why = bench.multi_gpu_verdict(line, 4, same_device=False)
    assert why and "1 physical device" in why and "RCCL" in why
    assert bench.multi_gpu_verdict(line, 4, same_device=True) is None           # the labelled dry run
    spread = [{"rank": r, "pid": 100 + r, "device": r, "physical_device": "box/pci:0000:%02x:00" % (5 + r)} for r in range(4)]
    good = dict(bench.device_census(spread), rccl_ranks=4)
    assert good["distinct_devices"] == 4 and bench.multi_gpu_verdict(good, 4, False) is None
    # every rank on its own GPU but the exchange fell back to something that is not the 4-rank communicator
    assert "RCCL" in bench.multi_gpu_verdict(dict(good, rccl_ranks=None), 4, False)
    assert bench.multi_gpu_verdict({"distinct_devices": 1}, 1, False) is None   # N = 1: nothing to check
