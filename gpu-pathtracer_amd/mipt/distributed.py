"""Frame assembly across the GPUs of one node (one process per GPU, torch.distributed).

The reference is single-device; this is the multi-GPU row of the design.  Work shards with
no collective on the data path (scene replicated, pixels/samples independent); the only
exchange is the framebuffer at the end:

* tile sharding   -- rank r renders tiles t with t % world == r; every rank packs the tiles
  it owns into a dense buffer, one all-gather moves each slice over its own direct xGMI
  link (no ring through the frame), every rank unpacks all slices.  Bit-identical to a
  single-GPU render.
* sample sharding -- rank r renders its own sample range of the whole frame; the films
  (rgb sums and sample counts) are summed with one reduce / all-reduce.

`pack` / `unpack` are callables so the same code runs on HIP kernels
(Film.pack_tiles / Film.unpack_tiles) and, in the CPU tests, on numpy.
"""
from __future__ import annotations

from typing import Callable

import torch
import torch.distributed as dist


def tile_shard_kwargs(rank: int, world: int, tile_size: int = 64) -> dict:
    return {"tile_size": tile_size, "shard_index": rank, "shard_count": world}


def sample_shard_kwargs(rank: int, world: int, spp_total: int) -> dict:
    """Contiguous sample ranges, sizes differing by at most one."""
    base, extra = divmod(spp_total, world)
    begin = rank * base + min(rank, extra)
    return {"sample_begin": begin, "sample_count": base + (1 if rank < extra else 0)}


def assemble_tiles(packed_sizes: list, pack: Callable[[int, torch.Tensor], None],
                   unpack: Callable[[int, torch.Tensor], None], device, group=None) -> None:
    """packed_sizes[r] = float4 slots of rank r's shard.  pack(r, buf) fills buf (n,4) with this
    rank's tiles; unpack(r, buf) scatters rank r's tiles into the local film."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_max = max(packed_sizes)
    mine = torch.zeros((n_max, 4), dtype=torch.float32, device=device)
    pack(rank, mine[: packed_sizes[rank]])
    everyone = torch.empty((world, n_max, 4), dtype=torch.float32, device=device)
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.current_stream().synchronize()
    dist.all_gather_into_tensor(everyone.view(-1), mine.view(-1), group=group)
    for r in range(world):
        if r != rank:
            unpack(r, everyone[r, : packed_sizes[r]])


def assemble_samples(film: torch.Tensor, dst: int | None = 0, group=None) -> None:
    """Sum the ranks' films in place (on dst, or on every rank when dst is None)."""
    if dst is None:
        dist.all_reduce(film, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM, group=group)
