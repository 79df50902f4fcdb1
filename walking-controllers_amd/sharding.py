"""
Multi-GPU sharding of a batch of independent robot instances.

Instances share only batch-constant data, so the batch is split into contiguous blocks —
rank r owns instances [r*B/G, (r+1)*B/G) — and there is NO collective on the data path.
The only exchange the north_star names is optional: scatter the inputs from rank 0 and
gather the solutions back (RCCL over xGMI when the backend is "nccl"; "gloo" in CPU tests).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence, Tuple

import numpy as np


def shard_range(global_batch: int, world: int, rank: int) -> Tuple[int, int]:
    """(first, count) of rank's contiguous block; the remainder goes to the first ranks."""
    base, rem = divmod(global_batch, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def scatter_batch(dist, full: Optional[Dict[str, "object"]], like: Dict[str, "object"], src: int = 0) -> Dict[str, "object"]:
    """Scatter every tensor of `full` (only meaningful on `src`; first dim = world * per-rank
    rows, equal shards) into per-rank tensors shaped like `like`."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    out = {}
    for k, ref in like.items():
        dst = torch.empty_like(ref)
        chunks = list(full[k].chunk(world, dim=0)) if rank == src else None
        dist.scatter(dst, [c.contiguous() for c in chunks] if chunks is not None else None, src=src)
        out[k] = dst
    return out


def gather_batch(dist, local: Dict[str, "object"], dst: int = 0) -> Optional[Dict[str, "object"]]:
    """Gather per-rank result tensors (equal shards) on `dst`, concatenated in rank order."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    out = {}
    for k, t in local.items():
        bufs = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
        dist.gather(t.contiguous(), bufs, dst=dst)
        if rank == dst:
            out[k] = torch.cat(bufs, dim=0)
    return out if rank == dst else None


def solve_sharded(dist, global_batch: int, make_inputs: Callable[[int, int], Dict[str, np.ndarray]],
                  solve: Callable[[Dict[str, np.ndarray]], Dict[str, np.ndarray]],
                  exchange: bool = False, device: str = "cpu") -> Optional[Dict[str, np.ndarray]]:
    """One sharded pass.  Without `exchange` every rank generates its own block
    (`make_inputs(first, count)` is counter-based, so the rows equal the full batch's);
    with `exchange` rank 0 generates the full batch and scatters it.  Results are gathered
    on rank 0.  `solve` is the per-shard solver (the HIP path in production)."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    assert global_batch % world == 0, "equal shards (weak scaling): global batch must divide by the world size"
    first, count = shard_range(global_batch, world, rank)
    if exchange:
        template = make_inputs(0, count)
        like = {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in template.items()}
        full = None
        if rank == 0:
            fb = make_inputs(0, global_batch)
            full = {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in fb.items()}
        got = scatter_batch(dist, full, like)
        inputs = {k: v.cpu().numpy() for k, v in got.items()}
    else:
        inputs = make_inputs(first, count)
    res = solve(inputs)
    gathered = gather_batch(dist, {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in res.items()})
    return None if gathered is None else {k: v.cpu().numpy() for k, v in gathered.items()}
