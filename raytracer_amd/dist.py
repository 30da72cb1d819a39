"""Multi-GPU partitioning of a render: one process per GPU, torch.distributed for the
one exchange step (the sum of the per-GPU accumulation buffers).

Samples are independent and a sample's random stream depends only on
(pixel, sample index, seed), so the natural partition is by sample index: rank r of
`world` renders its own contiguous range of sample indices for every pixel, and the
per-rank sum buffers add up to the image one GPU would render for the union of the
ranges.  Large frames can additionally be cut into row bands (tiles): a grid of
`tile_groups` x (world / tile_groups) ranks, each rendering one band with one share of
the samples.  The only collective is a sum-reduce of the W*H*3 float buffers onto the
root (RCCL over xGMI with the "nccl" backend; "gloo" on CPU in the tests).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import List, Optional, Sequence, Tuple


def split_range(n: int, parts: int, index: int) -> Tuple[int, int]:
    """(start, count) of share `index` when n items are cut into `parts` near-equal
    contiguous shares (the first n % parts shares get one more)."""
    if parts <= 0 or not 0 <= index < parts:
        raise ValueError("bad partition")
    q, r = divmod(n, parts)
    start = index * q + min(index, r)
    return start, q + (1 if index < r else 0)


def sample_range(spp_total: int, rank: int, world: int, sample_base: int = 0) -> Tuple[int, int]:
    """Strong scaling: this rank's (first sample index, count) of spp_total samples."""
    start, count = split_range(spp_total, world, rank)
    return sample_base + start, count


def weak_sample_range(spp_per_rank: int, rank: int, sample_base: int = 0) -> Tuple[int, int]:
    """Weak scaling: every rank renders spp_per_rank samples, ranges disjoint."""
    return sample_base + rank * spp_per_rank, spp_per_rank


def tile_and_samples(height: int, spp_total: int, rank: int, world: int, tile_groups: int,
                     sample_base: int = 0) -> Tuple[int, int, int, int]:
    """Row band + sample share for a tile_groups x (world/tile_groups) grid of ranks.
    Returns (row0, row1, first sample index, sample count)."""
    if world % tile_groups:
        raise ValueError("world must be a multiple of tile_groups")
    replicas = world // tile_groups
    band, rep = divmod(rank, replicas)
    r0, nrows = split_range(height, tile_groups, band)
    s0, ns = split_range(spp_total, replicas, rep)
    return r0, r0 + nrows, sample_base + s0, ns


def reduce_sum_to_root(accum, dst: int = 0):
    """Adds every rank's accumulation buffer onto rank `dst` (in place there)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum


def spawn_local_ranks(script: str, argv: Sequence[str], nprocs: int, env: Optional[dict] = None,
                      python: Optional[str] = None) -> int:
    """Starts `nprocs` copies of `script argv...` as child processes, one rank each, with the
    environment torch.distributed's env:// rendezvous reads (RANK, LOCAL_RANK, WORLD_SIZE,
    LOCAL_WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT=a free port), and waits for them.
    Rank 0 inherits this process's stdout; the other ranks' stdout is discarded (stderr is kept).
    Returns 0 when every rank exited with 0; otherwise the first failing rank's code, after
    ending the ranks that were still running (by their own process ids).

    The caller must not have touched the GPU: the children are fresh processes (no fork of an
    initialised runtime, no exec from one), each of which selects its own device."""
    if nprocs < 1:
        raise ValueError("nprocs must be at least 1")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    base = dict(os.environ if env is None else env)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs: List[subprocess.Popen] = []
    for rank in range(nprocs):
        e = dict(base, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(nprocs),
                 LOCAL_WORLD_SIZE=str(nprocs), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        out = None if rank == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([python or sys.executable, script, *argv], env=e, stdout=out))
    rc = 0
    pending = set(range(nprocs))
    while pending:
        for r in sorted(pending):
            try:
                code = procs[r].wait(timeout=0.2)
            except subprocess.TimeoutExpired:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                for other in pending:  # a rank died: the others would wait in a collective for ever
                    procs[other].terminate()
    return rc
