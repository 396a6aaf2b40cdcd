"""pt_trace_closest_device_static: the closest-hit query that allocates nothing.

Checked through the C ABI against pt_trace_closest_device (the entry point it is a variant of)
and against the CPU oracle, bit for bit, on two calls in a row, the second with fewer rays --
the static scratch array keeps the first call's contents -- and on a deep scene whose rays
outgrow the LDS stack, so that the static array's stack slots are really used.
"""
import numpy as np
import pytest

from conftest import random_rays, same_hits

pytestmark = pytest.mark.gpu


def _device_trace(torch, sc, rays, static, flags=0):
    d_r = torch.from_numpy(rays.view(np.float32).reshape(-1, 8).copy()).cuda()
    d_h = torch.full_like(d_r, -7.0)
    torch.cuda.synchronize()
    if static:
        # device memory in use must not grow over the call (second call on: the first one may
        # load the code object)
        sc.trace_closest_device_static(d_r.data_ptr(), len(rays), d_h.data_ptr(), flags)
        sc.ctx.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        d_h.fill_(-7.0)
        torch.cuda.synchronize()
        sc.trace_closest_device_static(d_r.data_ptr(), len(rays), d_h.data_ptr(), flags)
        sc.ctx.synchronize()
        assert torch.cuda.mem_get_info()[0] >= free0, "the static entry point allocated device memory"
    else:
        sc.trace_closest_device(d_r.data_ptr(), len(rays), d_h.data_ptr(), flags)
    sc.ctx.synchronize()
    return d_h.cpu().numpy().view(sc_hit_dtype).reshape(-1)


sc_hit_dtype = None


def test_static_entry_point_matches_allocating_one(mipt, oracle, scenes, gpu_ctx):
    import torch
    global sc_hit_dtype
    sc_hit_dtype = mipt.hit_dtype
    vtx, idx, mats, mids = scenes.atrium(65536, seed=42)
    nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx), mipt.bvh_options(**{"bvh.builder": "sah"}))
    sc = mipt.Scene(gpu_ctx, nodes, vtx, idx, mats, mids)
    try:
        for n, seed in ((300000, 11), (4099, 12)):           # the second call is the smaller one
            rays = scenes.surface_rays(vtx, idx, n, seed=seed)
            want = oracle.closest(nodes, vtx, idx, rays, nthreads=8)
            ref = _device_trace(torch, sc, rays, static=False)
            got = _device_trace(torch, sc, rays, static=True)
            assert len(same_hits(got, ref)) == 0, "static entry point differs from pt_trace_closest_device"
            assert len(same_hits(got, want)) == 0, "static entry point differs from the oracle"
            exact = _device_trace(torch, sc, rays, static=True, flags=mipt.PT_TRACE_EXACT_ORDER)
            assert len(same_hits(exact, want)) == 0
        # no rays at all, and more than a call takes
        sc.trace_closest_device_static(0, 0, 0)
        with pytest.raises(mipt.MiptError) as e:
            sc.trace_closest_device_static(1, (1 << 31), 1)
        assert e.value.code == mipt.PT_ERR_INVALID and "maximum" in e.value.message
    finally:
        sc.close()


def test_static_entry_point_uses_its_stack_slots(mipt, oracle, scenes, gpu_ctx):
    """Random triangles under the median-split builder: deep, overlapping subtrees in which rays
    keep more nodes pending than the LDS stack holds, so the stack slots in the static array
    (and beyond them the fixed-order walk) are exercised.  Then a small call on the same scene."""
    import torch
    global sc_hit_dtype
    sc_hit_dtype = mipt.hit_dtype
    vtx, idx, _, _ = scenes.random_soup(200000, seed=5)
    # the median-split builder over random triangles gives deep, overlapping subtrees
    nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx), mipt.bvh_options(**{"bvh.builder": "median"}))
    sc = mipt.Scene(gpu_ctx, nodes, vtx, idx)
    try:
        for n, seed in ((200000, 3), (1000, 4)):
            rays = random_rays(n, seed, dtype=mipt.ray_dtype)
            want = oracle.closest(nodes, vtx, idx, rays, nthreads=8)
            ref = _device_trace(torch, sc, rays, static=False)
            got = _device_trace(torch, sc, rays, static=True)
            assert len(same_hits(got, ref)) == 0
            assert len(same_hits(got, want)) == 0
    finally:
        sc.close()


def test_static_entry_point_under_graph_capture(mipt, oracle, scenes, gpu_ctx):
    """What the entry point is for: a caller that may not allocate.  After one ordinary call (the
    first call on a kernel asks the runtime for its occupancy, which cannot be captured) a
    second call is captured into a HIP graph on the context's stream and replayed twice; every
    replay gives the oracle's hits.  Single-level and two-level scenes."""
    import torch
    import two_level_util as tl
    from test_gpu_two_level import _build
    vtx, idx, mats, mids = scenes.atrium(65536, seed=42)
    nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx), mipt.bvh_options(**{"bvh.builder": "sah"}))
    flat = mipt.Scene(gpu_ctx, nodes, vtx, idx, mats, mids)
    flat_rays = scenes.surface_rays(vtx, idx, 100000, seed=21)
    flat_want = oracle.closest(nodes, vtx, idx, flat_rays, nthreads=8)
    tsc = tl.scene()
    t, two = _build(mipt, oracle, gpu_ctx, tsc)
    two_rays = tl.rays_for(50000, 6, mipt.ray_dtype)
    two_want = t.closest(two_rays, nthreads=8)
    stream = torch.cuda.Stream()
    try:
        gpu_ctx.set_stream(stream.cuda_stream)
        # each scene with its default walk and with the four-child nodes (other kernels)
        for sc, rays, want, quad in ((flat, flat_rays, flat_want, -1), (two, two_rays, two_want, -1),
                                     (flat, flat_rays, flat_want, 1), (two, two_rays, two_want, 1)):
            gpu_ctx.set_option("extend.quad", quad)
            n = len(rays)
            d_r = torch.from_numpy(rays.view(np.float32).reshape(-1, 8).copy()).cuda()
            d_h = torch.full_like(d_r, -7.0)
            torch.cuda.synchronize()
            sc.trace_closest_device_static(d_r.data_ptr(), n, d_h.data_ptr())      # the warming call
            gpu_ctx.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                sc.trace_closest_device_static(d_r.data_ptr(), n, d_h.data_ptr())
            for _ in range(2):
                d_h.fill_(-7.0)
                torch.cuda.synchronize()
                graph.replay()
                torch.cuda.synchronize()
                got = d_h.cpu().numpy().view(mipt.hit_dtype).reshape(-1)
                assert len(same_hits(got, want, nan_equal=True)) == 0
            del graph
    finally:
        gpu_ctx.synchronize()
        gpu_ctx.set_option("extend.quad", -1)
        gpu_ctx.set_stream(None)                       # back to the context's own stream
        flat.close()
        two.close()
