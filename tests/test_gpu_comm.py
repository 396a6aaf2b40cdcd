"""The RCCL entry points of the C ABI (pt_comm_*, pt_film_reduce, pt_film_gather_tiles) on the one
GPU a test box has: a communicator of one rank, made both ways (by unique id as one process per
GPU does, and by pt_comm_create_all as one process driving several GPUs does).  With one rank the
assembled frame must be the rank's own film, bit for bit; what several ranks exchange is covered
on the CPU (tests/test_distributed_cpu.py) through the same sharding rules."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _film_with_content(mipt, scenes, ctx):
    vtx, idx, mats, mids = scenes.cornell_box()
    nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx), mipt.bvh_options(**{"bvh.builder": "sah"}))
    sc = mipt.Scene(ctx, nodes, vtx, idx, mats, mids)
    cam = mipt.look_at_camera((278.0, 273.0, -800.0), (278.0, 273.0, 0.0), fovy_deg=39.3, aspect=1.5)
    film = mipt.Film(ctx, 96, 64)
    mipt.render(ctx, sc, cam, mipt.render_params(96, 64, sample_count=3, max_depth=3, seed=9), film)
    return sc, film


def test_one_rank_communicator_by_unique_id(mipt, scenes, gpu_ctx):
    assert mipt.comm_available()
    uid = mipt.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    comm = mipt.Comm(gpu_ctx, uid, 1, 0)
    sc, film = _film_with_content(mipt, scenes, gpu_ctx)
    try:
        assert (comm.rank, comm.size) == (0, 1)
        before = film.read()
        assert before[..., 3].min() == 3
        comm.reduce_film(film, root=0)
        comm.reduce_film(film, root=-1)                 # all-reduce form
        comm.gather_tiles(film, 16, root=0)
        comm.gather_tiles(film, 16, root=-1)
        np.testing.assert_array_equal(film.read().view(np.uint32), before.view(np.uint32))
        with pytest.raises(mipt.MiptError):
            comm.reduce_film(film, root=1)              # no such rank
    finally:
        film.close()
        sc.close()
        comm.close()
    with pytest.raises(mipt.MiptError):
        mipt.Comm(gpu_ctx, uid, 2, 2)                   # rank outside the communicator


def test_context_outlives_its_communicators(mipt):
    """A communicator uses its context's device and stream until it is destroyed: the context
    refuses to go first."""
    ctx = mipt.Context(0)
    comm = mipt.Comm(ctx, mipt.comm_unique_id(), 1, 0)
    with pytest.raises(mipt.MiptError, match="communicator"):
        ctx.close()
    film = mipt.Film(ctx, 32, 32)
    comm.reduce_film(film, root=0)                      # the context is still whole
    ctx.synchronize()
    film.close()
    comm.close()
    ctx.close()


def test_one_process_communicators(mipt, scenes, gpu_ctx):
    """pt_comm_create_all: what examples/ptrender.cpp --gpus N uses."""
    handles = (C.c_void_p * 1)(gpu_ctx._h)
    out = (C.c_void_p * 1)()
    assert mipt.lib.pt_comm_create_all(handles, 1, out) == mipt.PT_OK, mipt.last_error()
    sc, film = _film_with_content(mipt, scenes, gpu_ctx)
    try:
        assert mipt.lib.pt_comm_size(out[0]) == 1 and mipt.lib.pt_comm_rank(out[0]) == 0
        before = film.read()
        assert mipt.lib.pt_film_reduce(film._h, out[0], 0) == mipt.PT_OK, mipt.last_error()
        np.testing.assert_array_equal(film.read().view(np.uint32), before.view(np.uint32))
    finally:
        film.close()
        sc.close()
        mipt.lib.pt_comm_destroy(out[0])
    two = (C.c_void_p * 2)(gpu_ctx._h, gpu_ctx._h)
    out2 = (C.c_void_p * 2)()
    assert mipt.lib.pt_comm_create_all(two, 2, out2) == mipt.PT_ERR_INVALID      # two contexts on one GPU


@pytest.mark.parametrize("world,root", [(2, 0), (3, 2), (5, 1), (8, 0), (8, 7)])
def test_root_staging_layout_for_uneven_tile_counts(mipt, scenes, gpu_ctx, world, root):
    """What the root of pt_film_gather_tiles does with the peers' slices, replayed on one GPU:
    the slices (pt_film_packed_size pixels each, whole tiles; some ranks own fewer tiles than
    others, some tiles are cut by the frame's edge) lie side by side in one staging buffer in rank order, the
    root's own left out, and unpacking each from its offset assembles the frame bit for bit."""
    import torch
    sc, full = _film_with_content(mipt, scenes, gpu_ctx)        # 96 x 64
    W, H, tile = 96, 64, 24                                     # 4 x 3 tiles, the last row cut
    try:
        ref = full.read()
        slots = [full.packed_size(tile, k, world) for k in range(world)]
        assert sum(slots) == 12 * tile * tile and all(n % (tile * tile) == 0 for n in slots)
        assert (max(slots) > min(slots)) == (12 % world != 0)
        # every rank's film holds its own tiles only
        parts = []
        for k in range(world):
            part = mipt.Film(gpu_ctx, W, H)
            if slots[k]:
                buf = torch.empty((slots[k], 4), dtype=torch.float32, device="cuda")
                gpu_ctx.synchronize()
                full.pack_tiles(tile, k, world, buf.data_ptr())
                part.unpack_tiles(tile, k, world, buf.data_ptr())
                gpu_ctx.synchronize()
            parts.append(part)
        staging = torch.full((sum(slots) - slots[root], 4), float("nan"), dtype=torch.float32, device="cuda")
        off = 0
        for k in range(world):                                  # what the peers send, where the root receives it
            if k == root or slots[k] == 0:
                continue
            gpu_ctx.synchronize()
            parts[k].pack_tiles(tile, k, world, staging[off:].data_ptr())
            off += slots[k]
        assert off == staging.shape[0]
        off = 0
        for k in range(world):                                  # the root's unpacking loop
            if k == root or slots[k] == 0:
                continue
            parts[root].unpack_tiles(tile, k, world, staging[off:].data_ptr())
            off += slots[k]
        gpu_ctx.synchronize()
        np.testing.assert_array_equal(parts[root].read().view(np.uint32), ref.view(np.uint32))
        for p in parts:
            p.close()
    finally:
        full.close()
        sc.close()
