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


def step_shard_kwargs(rank: int, world: int, step: int, spp: int, scaling: str = "strong",
                      sharding: str = "samples", tile_size: int = 64):
    """The pt_render_params fields that give rank `rank` its share of step `step`, or None if
    the rank has nothing to render in it.  scaling "strong": a step is `spp` samples per pixel
    of the frame, split over the ranks; "weak": `spp` samples per pixel per rank, so a step of
    the frame is `spp * world`.  Over all ranks and steps every (pixel, sample index) is
    rendered exactly once, whatever the world size."""
    per_step = spp if scaling == "strong" else spp * world
    if sharding == "tiles":
        return dict(sample_begin=step * per_step, sample_count=per_step,
                    **tile_shard_kwargs(rank, world, tile_size))
    rng = sample_shard_kwargs(rank, world, per_step)
    if rng["sample_count"] == 0:
        return None
    return dict(sample_begin=step * per_step + rng["sample_begin"], sample_count=rng["sample_count"])


def build_bvh_once(leaf_bounds, options, rank: int, group=None, cache_path: str | None = None):
    """One host BVH build for the node, not one per rank: rank 0 builds and saves the tree
    (pt_bvh_save), the others wait and load it (pt_bvh_load checks the key of the inputs and the
    checksum; a rank whose load fails builds for itself).  Returns (nodes, how).  Collective."""
    import os
    import tempfile
    import mipt
    if cache_path is None:
        cache_path = os.path.join(tempfile.gettempdir(), "mipt_bvh_%s_%d.bvh"
                                  % (os.environ.get("MASTER_PORT", "0"), os.getuid()))
    how = "built by rank 0, loaded by the others (pt_bvh_save / pt_bvh_load)"
    nodes = None
    if rank == 0:
        if os.path.exists(cache_path):
            os.remove(cache_path)
        nodes, _ = mipt.build_bvh_cached(leaf_bounds, cache_path, options)
    dist.barrier(group=group)
    if rank != 0:
        nodes, loaded = mipt.build_bvh_cached(leaf_bounds, cache_path, options)
        if not loaded:
            how = "rank %d could not load rank 0's tree and built its own" % rank
    dist.barrier(group=group)
    if rank == 0 and os.path.exists(cache_path):
        os.remove(cache_path)
    return nodes, how


_abandoned = []


def abandoned_threads() -> int:
    """Threads left inside a communicator set-up that did not return in time (make_comm).  A
    process that has any should leave with os._exit once its results are out: interpreter
    shutdown would wait on them."""
    return sum(1 for t in _abandoned if t.is_alive())


def _comm_with_timeout(mipt, ctx, raw: bytes, world: int, rank: int, timeout_s: float):
    """mipt.Comm(...) on a helper thread, given up after timeout_s: ncclCommInitRank blocks until
    every rank has joined, so one rank that cannot join would otherwise hang all of them."""
    import threading
    box = {}

    def run():
        try:
            box["comm"] = mipt.Comm(ctx, raw, world, rank)
        except mipt.MiptError as e:
            box["error"] = e

    t = threading.Thread(target=run, name="mipt-comm-init", daemon=True)
    t.start()
    t.join(timeout_s if timeout_s and timeout_s > 0 else None)
    if t.is_alive():
        _abandoned.append(t)
        return None, "timed out after %.0f s" % timeout_s
    if "comm" in box:
        return box["comm"], ""
    return None, str(box.get("error", "failed"))


def make_comm(ctx, rank: int, world: int, prefer_abi: bool = True, group=None, timeout_s: float = 0.0):
    """The communicator of the frame assembly: the library's own RCCL communicator
    (pt_comm_create; its 128-byte id travels over torch.distributed), or None when the ranks
    do not all have it -- the caller then assembles with torch.distributed, which is RCCL as
    well.  Returns (comm or None, description).  Collective: every rank calls it.  With
    timeout_s > 0 a rank gives up on a communicator that has not come up by then, and with it
    every rank does."""
    import mipt
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    uid = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0 and prefer_abi and dev == "cuda":
        try:
            if mipt.comm_available():
                uid.copy_(torch.frombuffer(bytearray(mipt.comm_unique_id()), dtype=torch.uint8))
        except mipt.MiptError:
            uid.zero_()
    dist.broadcast(uid, src=0, group=group)
    raw = uid.cpu().numpy().tobytes()
    mine = 1 if (any(raw) and dev == "cuda" and mipt.comm_available()) else 0
    agreed = torch.tensor([mine], dtype=torch.int32, device=dev)
    dist.all_reduce(agreed, op=dist.ReduceOp.MIN, group=group)
    if int(agreed.item()) == 0:
        return None, "torch.distributed (RCCL)"
    comm, why = _comm_with_timeout(mipt, ctx, raw, world, rank, timeout_s)
    # a rank whose communicator did not come up takes every rank back to torch.distributed
    made = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(made, op=dist.ReduceOp.MIN, group=group)
    if int(made.item()) == 0:
        if comm is not None:
            comm.close()
        return None, ("torch.distributed (RCCL; the library's communicator did not come up on every rank%s)"
                      % (": rank %d %s" % (rank, why) if why else ""))
    return comm, "libmipt C ABI over RCCL (pt_film_reduce / pt_film_gather_tiles)"


def assemble_tiles(packed_sizes: list, pack: Callable[[int, torch.Tensor], None],
                   unpack: Callable[[int, torch.Tensor], None], device, group=None,
                   sync: Callable[[], None] | None = None) -> None:
    """packed_sizes[r] = float4 slots of rank r's shard.  pack(r, buf) fills buf (n,4) with this
    rank's tiles; unpack(r, buf) scatters rank r's tiles into the local film.

    pack and unpack may run on a stream torch knows nothing about (a pt_context's own): `sync`
    must then wait for that stream (Context.synchronize).  It is called after pack, before the
    collective reads the buffer, and after the last unpack, before the buffers go back to
    torch's allocator."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_max = max(packed_sizes)
    mine = torch.zeros((n_max, 4), dtype=torch.float32, device=device)
    everyone = torch.empty((world, n_max, 4), dtype=torch.float32, device=device)
    on_gpu = device is not None and torch.device(device).type == "cuda"
    if on_gpu:
        torch.cuda.current_stream().synchronize()      # the zero fill above, before pack writes
    pack(rank, mine[: packed_sizes[rank]])
    if sync is not None:
        sync()
    if on_gpu:
        torch.cuda.current_stream().synchronize()
    dist.all_gather_into_tensor(everyone.view(-1), mine.view(-1), group=group)
    if on_gpu:
        torch.cuda.current_stream().synchronize()      # the gathered slices, before unpack reads
    for r in range(world):
        if r != rank:
            unpack(r, everyone[r, : packed_sizes[r]])
    if sync is not None:
        sync()


def assemble_samples(film: torch.Tensor, dst: int | None = 0, group=None) -> None:
    """Sum the ranks' films in place (on dst, or on every rank when dst is None)."""
    if dst is None:
        dist.all_reduce(film, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM, group=group)
