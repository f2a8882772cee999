"""Column sharding of the forward-difference sweep across GPUs.

The n FD columns are independent given x0 (``scipy/optimize/_numdiff.py:592-620`` has no
cross-iteration dependency), so rank r of W evaluates the contiguous block
``[r*B, min(n, (r+1)*B))`` with ``B = ceil(n / W)``; F(x0) is recomputed on every rank (one
extra column) instead of being broadcast.  Every rank keeps a full ``n x m`` replica of the
transposed Jacobian that was zeroed ONCE; per step it sweeps its own block into its rows of
the replica (persistent-zero output: only the non-zeros are written), packs those non-zeros
(static pattern, 1.5-9 % of the block), and ONE all-gather of equal-sized messages
(``all_gather_into_tensor``: RCCL over xGMI with backend ``nccl``; ``gloo`` in the CPU tests)
hands every rank the others' packed blocks, which it scatters into its replica.  The message is
``block_vals * 8`` bytes per rank instead of ``B * m * 8`` (C3, W=2: 0.34 MB instead of 9.3 MB).

Two ways to run it, same layout (:func:`plan`), same results bit for bit:

* one process per GPU, ``torch.distributed`` - :class:`ShardedSweep` (``bench.py --gpus N``);
* one process driving several GPUs - ``HipEngine(devices=[...])`` / ``Problem.solve(devices=...)``,
  i.e. ``og_comm_init`` + ``og_multi_fd_sweep`` in ``libogpsx.so`` (``ncclCommInitAll`` + grouped
  ``ncclAllGather``).

:class:`ShardedSweep` talks to the device through a small backend object; :class:`HipBackend` is the
product (C ABI: ``og_fd_sweep_dev``, ``og_shard_pack_dev``, ``og_shard_unpack_dev``), and the CPU test
drives the very same class with a backend made of the oracle (``tests/test_sharding_gloo.py``).
"""
from __future__ import annotations

import numpy as np


def block_rows(n, world):
    return -(-int(n) // int(world))


def column_range(n, rank, world):
    """FD columns owned by ``rank``: ``(lo, hi)``, possibly empty for trailing ranks."""
    b = block_rows(n, world)
    lo = min(int(n), rank * b)
    return lo, min(int(n), lo + b)


def plan(indptr, world):
    """Layout of the exchanged messages for a pattern ``indptr`` (n+1 prefix sums of entries per column):
    ``(B, block_vals, offsets)`` - rank r's message holds the packed entries of its columns in pattern order,
    padded to ``block_vals`` (the largest block), and ``offsets[j]`` is where column j's entries start in the
    concatenation of all messages.  Identical to ``og_shard_plan`` (tested on the GPU)."""
    indptr = np.asarray(indptr, dtype=np.int64)
    n = indptr.size - 1
    B = block_rows(n, world)
    starts = np.minimum(n, np.arange(world) * B)
    ends = np.minimum(n, starts + B)
    block_vals = int(max(0, (indptr[ends] - indptr[starts]).max()))
    rank_of = np.arange(n) // B if n else np.zeros(0, dtype=np.int64)
    offsets = rank_of * block_vals + indptr[:-1] - indptr[starts[rank_of]] if n else np.zeros(0, dtype=np.int64)
    return B, block_vals, offsets.astype(np.int64)


class HipBackend:
    """Device side of :class:`ShardedSweep` on one MI355X: torch tensors for memory and streams, every
    operation a call into ``libogpsx.so``."""

    def __init__(self, engine, torch_device):
        import torch
        self.torch, self.engine, self.device = torch, engine, torch_device
        self.stream = torch.cuda.current_stream(torch_device).cuda_stream
        self._lib = engine._lib

    def pattern_indptr(self):
        return self.engine.pattern()[0]

    def zeros(self, *shape):
        return self.torch.zeros(shape, dtype=self.torch.float64, device=self.device)

    def empty(self, *shape):
        return self.torch.empty(shape, dtype=self.torch.float64, device=self.device)

    def upload(self, host_vector):
        return self.torch.from_numpy(np.ascontiguousarray(host_vector, dtype=np.float64)).to(self.device)

    def make_plan(self, world):
        import ctypes as C
        from . import _native
        B, bv = C.c_int32(), C.c_int64()
        _native.check(self._lib.og_shard_plan(self.engine._handle, int(world), C.byref(B), C.byref(bv)), "og_shard_plan")
        return B.value, bv.value

    def register_block(self, replica, lo, hi):
        if hi > lo:
            self.engine.register_jt_dev(replica[lo:hi].data_ptr(), lo, hi, self.stream)

    def bind_local_sweep(self, lo, hi, replica, F0):
        """``run(x_ptr, h_ptr)`` that enqueues this rank's block (``og_fd_sweep_dev``) with everything that does not
        change from step to step resolved ONCE - the view of the block, its address, F0's, the handle, the stream: per
        step the host pays one ctypes call instead of a tensor slice, four ``data_ptr()`` and three Python frames more.
        Measured (round 5, ``profiles/r05_host_ab.txt``, same lease, alternating): 6.4-6.6 us per step either way - the
        loop is bound by the GPU's launch-to-launch period, not by the host; kept because it is the shorter path."""
        if hi <= lo:
            return None
        from . import _native
        fn, handle, stream = self._lib.og_fd_sweep_dev, self.engine._handle, self.stream
        block, f0 = replica[lo:hi].data_ptr(), F0.data_ptr()
        lo, hi = int(lo), int(hi)

        def run(x_ptr, h_ptr):
            rc = fn(handle, x_ptr, h_ptr, lo, hi, block, f0, stream)
            if rc:
                _native.check(rc, "og_fd_sweep_dev")
        return run

    def sweep(self, x, h, lo, hi, replica, F0):
        if hi > lo:
            self.engine.sweep_dev(x.data_ptr(), h.data_ptr(), lo, hi, replica[lo:hi].data_ptr(), F0.data_ptr(),
                                  self.stream)
        else:                        # more ranks than columns: this rank only evaluates F(x0)
            self.engine.eval_dev(x.data_ptr(), F0.data_ptr(), self.stream)

    def sweep_and_pack(self, rank, x, h, lo, hi, replica, F0, send):
        """The rank's block and its message in ONE launch (``og_shard_sweep_dev``)."""
        from . import _native
        block = replica[lo:hi] if hi > lo else replica
        _native.check(self._lib.og_shard_sweep_dev(self.engine._handle, int(rank), x.data_ptr(), h.data_ptr(),
                                                   block.data_ptr(), F0.data_ptr(), send.data_ptr(), self.stream),
                      "og_shard_sweep_dev")

    def pack(self, rank, lo, hi, replica, send):
        from . import _native
        block = replica[lo:hi] if hi > lo else replica
        _native.check(self._lib.og_shard_pack_dev(self.engine._handle, int(rank), block.data_ptr(), send.data_ptr(),
                                                  self.stream), "og_shard_pack_dev")

    def unpack(self, rank, recv, replica):
        from . import _native
        _native.check(self._lib.og_shard_unpack_dev(self.engine._handle, int(rank), recv.data_ptr(),
                                                    replica.data_ptr(), self.stream), "og_shard_unpack_dev")

    # ---- the exchange: RCCL directly on this backend's stream when a communicator could be made ----
    direct = False
    direct_note = "torch.distributed.all_gather_into_tensor"

    def init_direct_rccl(self, rank, world, group=None):
        """``ncclCommInitRank`` inside ``libogpsx.so`` (``og_shard_comm_init``): the all-gather then runs on the
        stream the sweep, pack and unpack kernels are on, without the process group's own stream and events in
        between.  The unique id travels through ``torch.distributed`` (any backend).  Falls back silently to
        ``all_gather_into_tensor`` - every rank takes the same decision."""
        import ctypes as C
        import torch.distributed as dist
        torch = self.torch
        from . import _native
        ok = 1
        buf = (C.c_uint8 * 128)()
        if rank == 0:
            ok = 1 if self._lib.og_shard_comm_unique_id(buf) == 0 else 0
        on_gpu = dist.get_backend(group) == "nccl"
        t = torch.tensor([ok] + list(bytes(buf)), dtype=torch.int32, device=self.device if on_gpu else "cpu")
        dist.broadcast(t, src=0, group=group)
        host = t.cpu().tolist()
        if host[0] == 1:
            ident = (C.c_uint8 * 128)(*[int(v) & 0xff for v in host[1:]])
            ok = 1 if self._lib.og_shard_comm_init(self.engine._handle, ident, int(rank), int(world)) == 0 else 0
        else:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=self.device if on_gpu else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        self.direct = bool(int(flag.item()) == 1)
        if self.direct:
            self.direct_note = "ncclAllGather on the sweep's stream (og_shard_all_gather_dev)"
        else:
            self._lib.og_shard_comm_destroy(self.engine._handle)
            msg = self._lib.og_last_error()
            self.direct_note = "torch.distributed.all_gather_into_tensor (direct RCCL unavailable: %s)" % (
                msg.decode() if msg else "?")
        return self.direct

    host_staged = False     # dry runs with a CPU process group (gloo): the messages travel through host memory

    def all_gather(self, send, recv, group=None):
        if self.host_staged:
            import torch.distributed as dist
            self.torch.cuda.current_stream(self.device).synchronize()
            out = self.torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_gather_into_tensor(out, send.cpu(), group=group)
            recv.copy_(out)
        elif self.direct:
            from . import _native
            _native.check(self._lib.og_shard_all_gather_dev(self.engine._handle, send.data_ptr(), recv.data_ptr(),
                                                            self.stream), "og_shard_all_gather_dev")
        else:
            import torch.distributed as dist
            dist.all_gather_into_tensor(recv, send, group=group)


class ShardedSweep:
    """One rank's share of the column-sharded sweep: ``step(x, h)`` leaves the WHOLE transposed Jacobian in
    ``self.replica`` (n x m) and F(x0) in ``self.F0`` on every rank."""

    def __init__(self, backend, n, m, rank, world, group=None, exchange_alone=False):
        self.backend, self.n, self.m, self.rank, self.world, self.group = backend, int(n), int(m), int(rank), int(world), group
        self.exchange_alone = bool(exchange_alone)       # run pack / all-gather / unpack with one rank too (plumbing checks)
        self.lo, self.hi = column_range(n, rank, world)
        self.B, self.block_vals, self.offsets = plan(backend.pattern_indptr(), world)
        made = backend.make_plan(world)
        assert made == (self.B, self.block_vals), "shard plan of the library differs from sharding.plan: %r" % (made,)
        self.replica = backend.zeros(self.n, self.m)
        self.F0 = backend.empty(self.m)
        self.send = backend.zeros(max(self.block_vals, 1))
        self.recv = backend.empty(max(self.block_vals, 1) * self.world)
        backend.register_block(self.replica, self.lo, self.hi)
        bind = getattr(backend, "bind_local_sweep", None)
        self._local = bind(self.lo, self.hi, self.replica, self.F0) if bind else None

    @property
    def message_bytes(self):
        return 8 * self.block_vals

    def bound_step(self, x, h, gather=True):
        """A zero-argument callable doing ``step(x, h, gather)`` for these two device vectors: where the step is this
        rank's block alone (one rank, or ``gather=False``) every address is captured and the call is the ctypes call
        and nothing else."""
        if self._local is not None and (not gather or (self.world == 1 and not self.exchange_alone)):
            local, x_ptr, h_ptr = self._local, x.data_ptr(), h.data_ptr()
            keep = (x, h)                                   # (the vectors must outlive the callable)

            def run(local=local, x_ptr=x_ptr, h_ptr=h_ptr, keep=keep):
                local(x_ptr, h_ptr)
            return run
        return lambda: self.step(x, h, gather=gather)

    def step(self, x, h, gather=True):
        """``x``, ``h``: device vectors of the backend.  ``gather=False`` stops after this rank's own block."""
        be = self.backend
        if not gather or (self.world == 1 and not self.exchange_alone):
            if self._local is not None:
                self._local(x.data_ptr(), h.data_ptr())
            else:
                be.sweep(x, h, self.lo, self.hi, self.replica, self.F0)
            return self.replica
        if hasattr(be, "sweep_and_pack"):
            be.sweep_and_pack(self.rank, x, h, self.lo, self.hi, self.replica, self.F0, self.send)
        else:
            be.sweep(x, h, self.lo, self.hi, self.replica, self.F0)
            be.pack(self.rank, self.lo, self.hi, self.replica, self.send)
        if hasattr(be, "all_gather"):
            be.all_gather(self.send, self.recv, group=self.group)
        else:
            import torch.distributed as dist
            dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        be.unpack(self.rank, self.recv, self.replica)
        return self.replica
