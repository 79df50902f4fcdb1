"""
Multi-GPU sharding of a batch of independent robot instances.

Instances share only batch-constant data, so the batch is split into contiguous blocks —
rank r owns instances [r*B/G, (r+1)*B/G) — and there is NO collective on the data path.
The only exchange the north_star names is optional: scatter the inputs from rank 0 and
gather the solutions back (RCCL over xGMI when the backend is "nccl"; "gloo" in CPU tests) -
ONE scatter and ONE gather per step over per-rank slabs (ShardSlabs; layout in include/wcqp.h).
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


class ShardSlabs:
    """The exchange of one rank as TWO buffers (include/wcqp.h: shard slabs): every input array of its block of robots in one
    contiguous tensor, every output in another, laid out by the library (wcqp_slab_layout_for) so that the solve reads and
    writes them in place.  A step's exchange is then ONE scatter and ONE gather, whatever the backend - not one per array.

    rank `src` additionally holds one input and one output slab PER DESTINATION (`peers_in`, `peers_out`).
    device: where the slabs live ("cpu" for gloo tests).  With a GPU device and a gloo group (a rehearsal on one card) the
    collectives are staged through host copies of the slabs."""

    def __init__(self, dist, batch: int, ref_len: int, device="cpu", src: int = 0):
        import torch
        from . import capi
        self.dist, self.torch, self.src = dist, torch, src
        self.world, self.rank = (dist.get_world_size(), dist.get_rank()) if dist is not None else (1, 0)
        self.layout = capi.SlabLayout.make(batch, ref_len)
        self.device = torch.device(device)
        mk = lambda n: torch.zeros(int(n), dtype=torch.uint8, device=self.device)
        self.inp, self.out = mk(self.layout.in_bytes), mk(self.layout.out_bytes)
        self.peers_in = [mk(self.layout.in_bytes) for _ in range(self.world)] if self.rank == src else None
        self.peers_out = [mk(self.layout.out_bytes) for _ in range(self.world)] if self.rank == src else None
        self.staged = dist is not None and self.device.type != "cpu" and dist.get_backend() != "nccl"
        if self.staged:
            self._h_in, self._h_out = self.inp.cpu(), self.out.cpu()
            self._h_peers_in = [t.cpu() for t in self.peers_in] if self.peers_in else None
            self._h_peers_out = [t.cpu() for t in self.peers_out] if self.peers_out else None

    @staticmethod
    def _views(buf, shapes):
        import torch
        tt = {np.float64: torch.float64, np.int32: torch.int32, np.uint32: torch.int32}
        out = {}
        for k, (off, dt, shp) in shapes.items():
            n = int(np.prod(shp)) * np.dtype(dt).itemsize
            out[k] = buf[off:off + n].view(tt[dt]).view(*shp)
        return out

    def in_views(self, buf=None):
        """name -> tensor view of an input slab's arrays (no copy)."""
        return self._views(self.inp if buf is None else buf, self.layout.in_shapes())

    def out_views(self, buf=None):
        return self._views(self.out if buf is None else buf, self.layout.out_shapes())

    def fill_peer(self, r: int, arrays: Dict[str, np.ndarray]):
        """src only: destination r's input slab <- its block's arrays (numpy, ABI layout)."""
        v = self.in_views(self.peers_in[r])
        for k, t in v.items():
            a = np.ascontiguousarray(arrays[k], dtype=np.int32 if k == "hull_nc" else np.float64)
            t.copy_(self.torch.from_numpy(a).reshape(t.shape))

    def step(self):
        """The step record of this rank: pointers into its own two slabs."""
        return self.layout.step(self.inp.data_ptr(), self.out.data_ptr())

    def scatter(self):
        """ONE collective: every rank's input slab <- src's per-destination slab."""
        if self.dist is None:
            self.inp.copy_(self.peers_in[0])
            return
        if self.staged:
            if self.rank == self.src:
                for h, t in zip(self._h_peers_in, self.peers_in):
                    h.copy_(t)
            self.dist.scatter(self._h_in, self._h_peers_in if self.rank == self.src else None, src=self.src)
            self.inp.copy_(self._h_in)
        else:
            self.dist.scatter(self.inp, self.peers_in if self.rank == self.src else None, src=self.src)

    def gather(self):
        """ONE collective: src's per-source output slabs <- every rank's output slab."""
        if self.dist is None:
            self.peers_out[0].copy_(self.out)
            return
        if self.staged:
            self._h_out.copy_(self.out)
            self.dist.gather(self._h_out, self._h_peers_out if self.rank == self.src else None, dst=self.src)
            if self.rank == self.src:
                for h, t in zip(self._h_peers_out, self.peers_out):
                    t.copy_(h)
        else:
            self.dist.gather(self.out, self.peers_out if self.rank == self.src else None, dst=self.src)

    def gathered(self) -> Optional[Dict[str, np.ndarray]]:
        """src only: the outputs of all ranks, concatenated in rank order (numpy)."""
        if self.rank != self.src:
            return None
        per = [self.out_views(b) for b in self.peers_out]
        res = {}
        for k in per[0]:
            a = np.concatenate([p[k].cpu().numpy() for p in per], axis=0)
            res[k] = a.view(np.uint32) if k in ("mpc_active", "active_lower", "active_upper") else a
        return res

    @property
    def bytes_per_step(self) -> int:
        """What one step's scatter + gather move per rank."""
        return int(self.layout.in_bytes + self.layout.out_bytes)


def solve_sharded_slabs(dist, global_batch: int, ref_len: int, make_inputs: Callable[[int, int], Dict[str, np.ndarray]],
                        solve_step: Callable[["object", int], None], device="cpu") -> Optional[Dict[str, np.ndarray]]:
    """One sharded pass in the slab form: rank 0 generates every rank's block, ONE scatter, `solve_step(step_record, count)` on each
    rank (the record's pointers point into the rank's slabs), ONE gather.  Returns the gathered outputs on rank 0."""
    world, rank = (dist.get_world_size(), dist.get_rank()) if dist is not None else (1, 0)
    assert global_batch % world == 0, "equal shards (weak scaling): global batch must divide by the world size"
    count = global_batch // world
    sl = ShardSlabs(dist, count, ref_len, device=device)
    if rank == 0:
        for r in range(world):
            sl.fill_peer(r, make_inputs(r * count, count))
    sl.scatter()
    solve_step(sl.step(), count)
    sl.gather()
    return sl.gathered()


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
