"""Column sharding of the forward-difference sweep across GPUs.

The n FD columns are independent given x0 (``scipy/optimize/_numdiff.py:592-620`` has no
cross-iteration dependency), so rank r of W evaluates the contiguous block
``[r*B, min(n, (r+1)*B))`` with ``B = ceil(n / W)``; F(x0) is recomputed on every rank (one
extra column) instead of being broadcast.  The transposed Jacobian is row-major ``n x m``, so
each rank's block is one contiguous slab and a single RCCL all-gather over xGMI reassembles
it; the last block is padded to ``B`` rows so that all ranks contribute equal-sized messages
(``all_gather_into_tensor``).  One process per GPU, ``torch.distributed`` (backend ``nccl`` is
RCCL on ROCm; the CPU tests use ``gloo``).
"""
from __future__ import annotations


def block_rows(n, world):
    return -(-int(n) // int(world))


def column_range(n, rank, world):
    """FD columns owned by ``rank``: ``(lo, hi)``, possibly empty for trailing ranks."""
    b = block_rows(n, world)
    lo = min(int(n), rank * b)
    return lo, min(int(n), lo + b)


def gathered_shape(n, m, world):
    return (block_rows(n, world) * world, int(m))


def all_gather_jt(local_block, out, group=None):
    """All-gather the per-rank slabs (``block_rows x m`` each, zero-padded) into ``out``
    (``block_rows*world x m``); the caller slices ``out[:n]``."""
    import torch.distributed as dist
    dist.all_gather_into_tensor(out, local_block, group=group)
    return out
