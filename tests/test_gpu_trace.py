"""Parity of the HIP traversal kernels with the CPU oracle, through the C ABI.

Bar: bit-exact pt_hit records (shapeid, primid, padding, and the raw bits of b0,b1,b2,t)
against IntersectSceneClosest restated in oracle/pt_oracle.c, on the same nodes and rays.
"""
import numpy as np
import pytest

from conftest import random_rays, same_hits
from test_oracle import cornell_rays, UNIT_TRI

pytestmark = pytest.mark.gpu


def _build(mipt, vtx, idx, builder="sah", **opts):
    lb = mipt.mesh_leaf_bounds(vtx, idx)
    return mipt.build_bvh(lb, mipt.bvh_options(**{"bvh.builder": builder, **opts}))


def _check_all_modes(mipt, oracle, ctx, nodes, vtx, idx, rays, expect_ties=None, nan_equal=False):
    want = oracle.closest(nodes, vtx, idx, rays, nthreads=8)
    sc = mipt.Scene(ctx, nodes, vtx, idx)
    try:
        exact = sc.trace_closest(rays, mipt.PT_TRACE_EXACT_ORDER)
        bad = same_hits(exact, want, nan_equal)
        assert len(bad) == 0, (f"fixed-order kernel differs from the oracle on {len(bad)} rays, "
                               f"first {bad[:8]}: {exact[bad[:3]]} vs {want[bad[:3]]}")
        got, st = sc.trace_closest(rays, mipt.PT_TRACE_DEFAULT, stats=True)
        bad = same_hits(got, want, nan_equal)
        assert len(bad) == 0, (f"ordered kernel differs from the oracle on {len(bad)} of "
                               f"{len(rays)} rays, first {bad[:8]}: {got[bad[:3]]} vs {want[bad[:3]]}")
        assert st.rays == len(rays)
        if expect_ties is not None:
            assert expect_ties(st.tie_rays), f"tie_rays = {st.tie_rays}"
        # the same tree as four-child nodes with quantised boxes (what scenes beyond the caches
        # are walked by): every record the same again
        ctx.set_option("extend.quad", 1)
        try:
            quad, stq = sc.trace_closest(rays, mipt.PT_TRACE_DEFAULT, stats=True)
        finally:
            ctx.set_option("extend.quad", -1)
        bad = same_hits(quad, want, nan_equal)
        assert len(bad) == 0, (f"four-child traversal differs from the oracle on {len(bad)} of "
                               f"{len(rays)} rays, first {bad[:8]}: {quad[bad[:3]]} vs {want[bad[:3]]}")
        # the same again with the waves' last rays handed to the wave-per-ray search after one
        # round instead of 64 (what a launch does with rays that would hold it up): the rays that
        # went that way are counted, and every record is the same
        ctx.set_option("extend.tail_budget", 1)
        try:
            for q in (0, 1):
                ctx.set_option("extend.quad", q)
                handed, sth = sc.trace_closest(rays, mipt.PT_TRACE_DEFAULT, stats=True)
                bad = same_hits(handed, want, nan_equal)
                assert len(bad) == 0, (f"with rays handed to the wave-per-ray search ({sth.long_rays} of them, "
                                       f"{'four' if q else 'two'}-child walk) {len(bad)} of {len(rays)} rays "
                                       f"differ from the oracle, first {bad[:8]}: {handed[bad[:3]]} vs {want[bad[:3]]}")
                # (the hand-over is part of the four-child kernels only)
                if q == 0:
                    assert sth.long_rays == 0
                elif len(rays) >= 4096 and idx.shape[0] >= 1000:
                    assert sth.long_rays > 0
        finally:
            ctx.set_option("extend.tail_budget", 64)
            ctx.set_option("extend.quad", -1)
        # occlusion queries (IntersectSceneAny) on the same rays, their reach set around the
        # nearest hit so that some are blocked and some are not: two-child and four-child walk
        ra = rays.copy()
        t = want["uvwt"][:, 3]
        hit = (want["primid"] >= 0) & np.isfinite(t)
        scale = np.random.default_rng(len(rays)).uniform(0.5, 1.5, len(rays)).astype(np.float32)
        ra["o"][hit, 3] = t[hit] * scale[hit]
        want_any = oracle.any_hit(nodes, vtx, idx, ra, nthreads=8)
        for q in (0, 1):
            ctx.set_option("extend.quad", q)
            try:
                got_any = sc.trace_any(ra)
            finally:
                ctx.set_option("extend.quad", -1)
            bad = np.nonzero(got_any != want_any)[0]
            assert len(bad) == 0, (f"occlusion ({'four' if q else 'two'}-child walk) differs from the oracle on "
                                   f"{len(bad)} of {len(rays)} rays, first {bad[:8]}")
        return got, st
    finally:
        sc.close()


def test_unit_triangle_and_conventions(mipt, oracle, gpu_ctx):
    idx = np.array([[0, 1, 2]], dtype=np.int32)
    nodes = _build(mipt, UNIT_TRI, idx)
    rays = np.zeros(4, dtype=mipt.ray_dtype)
    rays["o"] = [(0, 0, 5, 1e30), (2, 2, 5, 1000.0), (0, 0, 5, 4.999), (0, 0, -5, 1e30)]
    rays["d"] = [(0, 0, -1, 0), (0, 0, -1, 0), (0, 0, -1, 0), (0, 0, 1, 0)]
    got, _ = _check_all_modes(mipt, oracle, gpu_ctx, nodes, UNIT_TRI, idx, rays)
    np.testing.assert_array_equal(got["uvwt"][0], np.float32([0.25, 0.5, 0.25, 5.0]))
    assert tuple(got["primid"]) == (0, -1, -1, 0) and tuple(got["shapeid"]) == (0, -1, -1, 0)
    assert (got["padding"] == [667, 0]).all()                      # bvh.glslh:429-430
    np.testing.assert_array_equal(got["uvwt"][1], np.float32([0, 0, 0, 1000.0]))


def test_fallback_quad(mipt, oracle, scenes, gpu_ctx):
    vtx, idx, _, _ = scenes.fallback_quad()                        # src/main.cpp:918-921
    nodes = _build(mipt, vtx, idx)
    rays = random_rays(20000, 1, extent=3.0, dtype=mipt.ray_dtype)
    got, _ = _check_all_modes(mipt, oracle, gpu_ctx, nodes, vtx, idx, rays)
    assert set(np.unique(got["primid"])) == {-1, 0, 1}


@pytest.mark.parametrize("builder", ["sah", "median"])
@pytest.mark.parametrize("n_tri,seed", [(1, 0), (2, 1), (7, 2), (1000, 3), (50000, 4)])
def test_random_soup(mipt, oracle, scenes, gpu_ctx, n_tri, seed, builder):
    vtx, idx, _, _ = scenes.random_soup(n_tri, seed=seed)
    nodes = _build(mipt, vtx, idx, builder)
    rays = random_rays(100000, seed + 50, dtype=mipt.ray_dtype)
    got, st = _check_all_modes(mipt, oracle, gpu_ctx, nodes, vtx, idx, rays)
    if n_tri == 1:
        assert st.tie_rays == 0          # the single-triangle root must not fake a second candidate
    if n_tri >= 1000:
        assert (got["primid"] >= 0).mean() > 0.1
        # random geometry: near ties are rare
        assert st.tie_rays < 50


def test_cornell_primary_rays_256(mipt, oracle, scenes, gpu_ctx):
    """Baseline config #1 on the GPU: Cornell box, 256x256 pixel-centre primary rays, hit ids
    bit-exact against the oracle, shared-edge ties included (they are re-resolved in the
    reference's order; the count is reported)."""
    vtx, idx, _, _ = scenes.cornell_box()
    nodes = _build(mipt, vtx, idx)
    rays = cornell_rays(256, 256).astype(mipt.ray_dtype)
    got, st = _check_all_modes(mipt, oracle, gpu_ctx, nodes, vtx, idx, rays)
    print(f"cornell 256x256: {st.tie_rays} rays re-resolved in skip-link order")
    assert (got["primid"] >= 0).mean() > 0.9


def test_ties_are_resolved_in_reference_order(mipt, oracle, gpu_ctx):
    """Coplanar duplicates and shared edges: every ray has several candidates at (nearly)
    the same distance; the answer must still be the reference's (last accepted in skip-link
    order), and the tie path must actually have run."""
    quad = np.array([[0, 0, 0], [2, 0, 0], [2, 2, 0], [0, 2, 0]], dtype=np.float32)
    vtx = np.concatenate([quad, quad, quad + np.float32([0, 0, 1e-7])])
    idx = np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7], [8, 9, 10], [8, 10, 11]], np.int32)
    for builder in ("sah", "median"):
        nodes = _build(mipt, vtx, idx, builder)
        n = 64
        g = (np.arange(n) + 0.5) / n * 2
        xx, yy = np.meshgrid(g, g)
        rays = np.zeros(n * n + n, dtype=mipt.ray_dtype)
        rays["o"][: n * n, 0] = xx.ravel()
        rays["o"][: n * n, 1] = yy.ravel()
        rays["o"][n * n:, 0] = g          # exactly on the shared diagonal
        rays["o"][n * n:, 1] = g
        rays["o"][:, 2] = 4
        rays["o"][:, 3] = 1e30
        rays["d"][:, 2] = -1
        _check_all_modes(mipt, oracle, gpu_ctx, nodes, vtx, idx, rays,
                         expect_ties=lambda t: t == n * n + n)


def test_axis_aligned_rays_and_flat_boxes(mipt, oracle, scenes, gpu_ctx):
    """Directions with zero components (1/d = inf, 0*inf = NaN in the slab test) against
    axis-aligned geometry with zero-thickness boxes."""
    vtx, idx, _, _ = scenes.cornell_box()
    nodes = _build(mipt, vtx, idx)
    rng = np.random.default_rng(3)
    n = 30000
    rays = np.zeros(n, dtype=mipt.ray_dtype)
    rays["o"][:, :3] = rng.uniform(1, 548, size=(n, 3)).astype(np.float32)
    rays["o"][:, 3] = 1e30
    axis = rng.integers(0, 3, n)
    sign = rng.choice([-1.0, 1.0], n)
    rays["d"][np.arange(n), axis] = sign
    # a third of them start exactly on a wall plane
    rays["o"][: n // 3, 0] = 0.0
    _check_all_modes(mipt, oracle, gpu_ctx, nodes, vtx, idx, rays)


def test_reference_raygen_and_depth_shade(mipt, oracle, scenes, gpu_ctx):
    """pt_raygen (depth.frag:47-83) and pt_depth_shade (:90-107) bit-exact, then the whole
    reference frame: raygen -> closest hit -> grey."""
    cam = mipt.look_at_camera((0.0, 0.0, -10.0), (0, 0, 0), fovy_deg=60.0, aspect=4 / 3,
                              mode=mipt.PT_CAMERA_REFERENCE)     # main_granite.cpp:306-308
    W, H = 160, 120
    rays = gpu_ctx.raygen(cam, W, H)
    ocam = oracle.camera_from(tuple(cam.eye), tuple(cam.c0), tuple(cam.c1), tuple(cam.c2),
                              cam.sx, cam.sy)
    want = oracle.raygen_reference(ocam, W, H)
    np.testing.assert_array_equal(rays.view(np.uint32), want.view(np.uint32))
    assert rays["o"][0, 3] == np.float32(20000.0)                # tmax after the vec4 divide
    # scene scaled so the reference's x20 origin sees it
    vtx, idx, _, _ = scenes.random_soup(3000, seed=8, extent=60.0, size=12.0)
    vtx = vtx + np.float32([0, 40, 0])
    nodes = _build(mipt, vtx, idx)
    got, _ = _check_all_modes(mipt, oracle, gpu_ctx, nodes, vtx, idx, rays)
    assert (got["primid"] >= 0).mean() > 0.05
    rgba = gpu_ctx.depth_shade(got)
    np.testing.assert_array_equal(rgba.view(np.uint32), oracle.depth_shade(got).view(np.uint32))


def test_moller_trumbore_mode_and_any_hit(mipt, oracle, scenes, gpu_ctx):
    vtx, idx, _, _ = scenes.random_soup(20000, seed=12)
    nodes = _build(mipt, vtx, idx)
    rays = random_rays(60000, 99, dtype=mipt.ray_dtype)
    rays["o"][::2, 3] = 6.0                                      # short rays for occlusion
    sc = mipt.Scene(gpu_ctx, nodes, vtx, idx)
    try:
        got = sc.trace_closest(rays, mipt.PT_TRACE_MOLLER_TRUMBORE)
        want = oracle.closest(nodes, vtx, idx, rays, nthreads=8, moller_trumbore=True)
        assert len(same_hits(got, want)) == 0
        want_occ = oracle.any_hit(nodes, vtx, idx, rays, nthreads=8)
        occ = sc.trace_any(rays)                                 # two-child nodes, near child first
        np.testing.assert_array_equal(occ, want_occ)
        assert 0.05 < occ.mean() < 0.95
        gpu_ctx.set_option("any.exact_order", 1)                 # the reference's skip-link walk
        try:
            np.testing.assert_array_equal(sc.trace_any(rays), want_occ)
        finally:
            gpu_ctx.set_option("any.exact_order", 0)
    finally:
        sc.close()


def test_int_bits_links_on_device(mipt, oracle, scenes, gpu_ctx):
    vtx, idx, _, _ = scenes.random_soup(4000, seed=5)
    lb = mipt.mesh_leaf_bounds(vtx, idx)
    nf = mipt.build_bvh(lb, mipt.bvh_options(**{"bvh.builder": "sah"}))
    ni = mipt.build_bvh(lb, mipt.bvh_options(**{"bvh.builder": "sah", "mipt.link_encoding": "int"}))
    rays = random_rays(50000, 6, dtype=mipt.ray_dtype)
    want = oracle.closest(nf, vtx, idx, rays, nthreads=8)
    sc = mipt.Scene(gpu_ctx, ni, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS)
    try:
        for flags in (mipt.PT_TRACE_EXACT_ORDER, mipt.PT_TRACE_DEFAULT):
            assert len(same_hits(sc.trace_closest(rays, flags), want)) == 0
    finally:
        sc.close()


def test_deep_tree_overflows_the_lds_stack_safely(mipt, oracle, gpu_ctx):
    """A chain-shaped tree (every split peels one triangle) is deeper than the LDS stack;
    such rays are handed to the stackless kernel and still match."""
    n = 200
    k = np.arange(n, dtype=np.float64)
    x = 1.5 ** np.minimum(k, 60) + k          # spacing grows so SAH keeps peeling one off
    tri = np.array([[0, -1, -1], [0, 1, -1], [0, 0, 1]], dtype=np.float64) * 0.4
    vtx = (tri[None] + np.stack([x, 0 * x, 0 * x], 1)[:, None]).reshape(-1, 3).astype(np.float32)
    idx = np.arange(3 * n, dtype=np.int32).reshape(-1, 3)
    nodes = _build(mipt, vtx, idx)
    rays = np.zeros(2 * n, dtype=mipt.ray_dtype)
    rays["o"][:n] = [(-5, 0, 0, 1e30)] * n            # along the chain: every box is entered
    rays["d"][:n, 0] = 1
    rays["o"][n:, 0] = x.astype(np.float32) + 0.3     # and from the far end backwards
    rays["o"][n:, 3] = 1e30
    rays["d"][n:, 0] = -1
    _check_all_modes(mipt, oracle, gpu_ctx, nodes, vtx, idx, rays)


def test_many_exact_ties_from_duplicated_triangles(mipt, oracle, scenes, gpu_ctx):
    """Coincident triangles: every hit on one is an exact tie, which the reference's visiting
    order decides.  With a third of the mesh duplicated (some of it three times) most waves
    hold several such rays at once; they are re-walked in batches, with the ordered traversal's
    distance as the starting bound, and must still come out as the fixed-order walk leaves them."""
    vtx, idx, _, _ = scenes.random_soup(6000, seed=17)
    idx = np.concatenate([idx, idx[:2000], idx[:700]])
    nodes = _build(mipt, vtx, idx)
    rays = random_rays(120000, 23, dtype=mipt.ray_dtype)
    rays["o"][::7, 3] = 3.0                                   # some with a finite reach
    got, st = _check_all_modes(mipt, oracle, gpu_ctx, nodes, vtx, idx, rays,
                               expect_ties=lambda t: t > 5000)
    dup = got["primid"] >= 6000
    assert dup.sum() > 1000 or (got["primid"] < 2000).sum() > 1000


def test_pairs_of_near_coincident_triangles(mipt, oracle, scenes, gpu_ctx):
    """Two candidates within the tie window are decided at their two leaves, without a walk over
    the tree: exact copies, copies with the vertices rotated (another rounding of the same
    plane), copies moved by a few units in the last place, and copies moved far enough to be
    outside the window for some rays and inside for others."""
    vtx, idx, _, _ = scenes.random_soup(4000, seed=29)
    n = idx.shape[0]
    moved = vtx.copy()
    moved[:, 2] = np.nextafter(np.nextafter(moved[:, 2], np.float32(np.inf)), np.float32(np.inf))
    far = vtx.copy()
    far[:, 2] *= np.float32(1 + 3e-6)
    v = np.concatenate([vtx, moved, far])
    nv = vtx.shape[0]
    idx = np.concatenate([idx, idx[:800], idx[800:1600][:, [1, 2, 0]], idx[1600:2400] + nv,
                          idx[2400:3200] + 2 * nv]).astype(np.int32)
    for builder in ("sah", "median"):
        nodes = _build(mipt, v, idx, builder)
        rays = random_rays(150000, 31, dtype=mipt.ray_dtype)
        rays["o"][::5, 3] = 2.5
        got, st = _check_all_modes(mipt, oracle, gpu_ctx, nodes, v, idx, rays,
                                   expect_ties=lambda t: t > 3000)
        assert (got["primid"] >= n).sum() > 500 and ((got["primid"] >= 0) & (got["primid"] < 3200)).sum() > 500
    # the same through the tree built on the GPU (its own table of leaves)
    sc = mipt.Scene(gpu_ctx, None, v, idx)
    try:
        want = oracle.closest(sc.read_nodes(), v, idx, rays, nthreads=8)
        assert len(same_hits(sc.trace_closest(rays), want)) == 0
    finally:
        sc.close()


def test_face_ids_of_a_flattened_multi_mesh_scene(mipt, oracle, scenes, gpu_ctx):
    """Several meshes in one tree (intersector_skip_links.cpp:169-268): a hit reports the
    shape_id / prim_id of its Face record instead of 0 / triangle index, in every query mode,
    on host-built and GPU-built trees; misses and everything else in the record stay as they are,
    and removing the table restores the shader's checked-in behaviour."""
    vtx, idx, _, _ = scenes.random_soup(3000, seed=41)
    n = idx.shape[0]
    shape = (np.arange(n) // 700).astype(np.int32) + 10            # five "meshes", ids 10..14
    prim = (np.arange(n) % 700).astype(np.int32)
    rays = random_rays(60000, 43, dtype=mipt.ray_dtype)
    nodes = _build(mipt, vtx, idx)
    want = oracle.closest(nodes, vtx, idx, rays, nthreads=8)
    hit = want["primid"] >= 0
    mapped = want.copy()
    mapped["shapeid"][hit] = shape[want["primid"][hit]]
    mapped["primid"][hit] = prim[want["primid"][hit]]
    assert hit.sum() > 1000 and (~hit).sum() > 1000
    for sc in (mipt.Scene(gpu_ctx, nodes, vtx, idx), ):
        try:
            sc.set_face_ids(shape, prim)
            for flags in (mipt.PT_TRACE_DEFAULT, mipt.PT_TRACE_EXACT_ORDER):
                assert len(same_hits(sc.trace_closest(rays, flags), mapped)) == 0
            d_r = __import__("torch").from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
            d_h = __import__("torch").empty_like(d_r)
            sc.trace_closest_device_static(d_r.data_ptr(), len(rays), d_h.data_ptr())
            gpu_ctx.synchronize()
            got = d_h.cpu().numpy().view(mipt.hit_dtype).reshape(-1)
            assert len(same_hits(got, mapped)) == 0
            sc.set_face_ids(None, None)
            assert len(same_hits(sc.trace_closest(rays), want)) == 0
            with pytest.raises(mipt.MiptError, match="together"):
                mipt._check(mipt.lib.pt_scene_set_face_ids(sc._h, mipt._ptr(shape), None, len(shape)))
            # one entry per triangle of the scene, no fewer: a short table would be read past its end
            assert sc.triangle_count() == len(shape)
            with pytest.raises(mipt.MiptError, match="ids for a scene of"):
                sc.set_face_ids(shape[:-1], prim[:-1])
        finally:
            sc.close()
    # a tree built on the GPU: compare with the walk of its own array
    sc = mipt.Scene(gpu_ctx, None, vtx, idx)
    try:
        w2 = oracle.closest(sc.read_nodes(), vtx, idx, rays, nthreads=8)
        h2 = w2["primid"] >= 0
        m2 = w2.copy()
        m2["shapeid"][h2] = shape[w2["primid"][h2]]
        m2["primid"][h2] = prim[w2["primid"][h2]]
        sc.set_face_ids(shape, prim)
        assert len(same_hits(sc.trace_closest(rays), m2)) == 0
    finally:
        sc.close()


def _comb_tree(oracle, vtx, idx):
    """A hand-made skip-link tree of depth n: inner node i holds triangle i and inner node i+1
    (pre-order: inner_0, leaf_0, inner_1, leaf_1, ..., leaf_{n-2}, leaf_{n-1}; float links)."""
    n = idx.shape[0]
    leaves = oracle.inflate(oracle.mesh_leaf_bounds(vtx, idx))
    nodes = np.zeros(2 * n - 1, dtype=leaves.dtype)
    lo = np.minimum.accumulate(leaves["pmin"][::-1, :3], 0)[::-1]      # box of triangles i..n-1
    hi = np.maximum.accumulate(leaves["pmax"][::-1, :3], 0)[::-1]
    for i in range(n - 1):
        nodes["pmin"][2 * i, :3], nodes["pmax"][2 * i, :3] = lo[i], hi[i]
        nodes["pmin"][2 * i, 3], nodes["pmax"][2 * i, 3] = -1, -1      # inner; nothing follows its subtree
        nodes[2 * i + 1] = leaves[i]
        nodes["pmin"][2 * i + 1, 3], nodes["pmax"][2 * i + 1, 3] = i, 2 * i + 2
    nodes[2 * n - 2] = leaves[n - 1]
    nodes["pmin"][2 * n - 2, 3], nodes["pmax"][2 * n - 2, 3] = n - 1, -1
    return nodes


@pytest.mark.parametrize("n", [25, 33, 45, 75, 150])
def test_stack_goes_on_in_global_memory(mipt, oracle, gpu_ctx, n):
    """Trees deeper than the LDS stack: a ray that enters the comb from its deep end holds one
    pending leaf per level.  Up to the room in global memory such rays stay in the voted loop
    (no re-resolved rays), beyond it they are left to the fixed-order walk; either way the hits
    are the oracle's, for closest-hit and occlusion queries, with many rays in flight so that
    lanes spill and reload independently."""
    x = np.arange(n, dtype=np.float64) * 1.25
    tri = np.array([[0, -1, -1], [0, 1, -1], [0, 0, 1]], dtype=np.float64) * 0.4
    vtx = (tri[None] + np.stack([x, 0 * x, 0 * x], 1)[:, None]).reshape(-1, 3).astype(np.float32)
    idx = np.arange(3 * n, dtype=np.int32).reshape(-1, 3)
    nodes = _comb_tree(oracle, vtx, idx)
    mipt.validate_bvh(nodes, n, mipt.PT_LINKS_FLOAT_VALUE)
    rng = np.random.default_rng(n)
    m = 20000
    rays = np.zeros(m, dtype=mipt.ray_dtype)
    # from beyond the last triangle back along the comb, slightly tilted, so that every level's
    # nearer child is the rest of the comb; a third of them miss everything and unwind the
    # whole stack; some start in the middle
    rays["o"][:, 0] = np.where(rng.random(m) < 0.7, x[-1] + 1.0, rng.uniform(0, x[-1], m))
    rays["o"][:, 1:3] = rng.uniform(-0.45, 0.45, (m, 2))
    rays["o"][:, 3] = 1e30
    rays["d"][:, 0] = -1
    rays["d"][:, 1:3] = rng.uniform(-0.004, 0.004, (m, 2))
    rays["d"][::3, 1] = rng.uniform(0.01, 0.02, len(rays[::3]))        # grazing: these mostly miss
    # pending nodes a lane can hold: LDS entries (one is kept free) + global slots.  The LDS
    # entries are set by hand here; left alone the library picks 16 or 24 by the tree's depth,
    # which the last round checks too (any number of re-resolved rays, same hits).
    for entries in (30, 12, 9, -1):          # 9: the smallest the option takes (a four-child walk gets 12)
        gpu_ctx.set_option("extend.stack_entries", entries)
        # every wave finishes its own rays here (this launch is all tail: one chunk per wave; with
        # the default budget the deep rays would go to the wave-per-ray search, whose stack holds them)
        gpu_ctx.set_option("extend.tail_budget", 0)
        try:
            limit = entries - 1 + 48
            expect = None if entries < 0 else ((lambda t: t == 0) if n - 1 <= limit else (lambda t: t > 0))
            got, st = _check_all_modes(mipt, oracle, gpu_ctx, nodes, vtx, idx, rays, expect_ties=expect)
        finally:
            gpu_ctx.set_option("extend.stack_entries", -1)
            gpu_ctx.set_option("extend.tail_budget", 64)
    assert (got["primid"] >= 0).mean() > 0.3 and (got["primid"] < 0).mean() > 0.05
    sc = mipt.Scene(gpu_ctx, nodes, vtx, idx)
    try:
        rays["o"][:, 3] = rng.uniform(0.5, 1.5 * x[-1], m)
        np.testing.assert_array_equal(sc.trace_any(rays), oracle.any_hit(nodes, vtx, idx, rays, nthreads=8))
    finally:
        sc.close()


def test_non_finite_rays_follow_the_reference_arithmetic(mipt, oracle, scenes, gpu_ctx):
    """NaN / infinite / zero-direction rays: the shader's arithmetic "accepts" a NaN distance
    on the first triangle it tests and then culls everything, so the answer depends on the
    visiting order; such rays must come out as the fixed-order traversal leaves them and must
    not walk the whole tree."""
    vtx, idx, _, _ = scenes.random_soup(5000, seed=31)
    nodes = _build(mipt, vtx, idx)
    rays = random_rays(4096, 32, dtype=mipt.ray_dtype)
    nan, inf = np.float32(np.nan), np.float32(np.inf)
    rays["d"][0::8, 0] = nan
    rays["d"][1::8, :3] = nan
    rays["o"][2::8, 1] = nan
    rays["o"][3::8, 2] = inf
    rays["d"][4::8, :3] = 0.0
    rays["o"][5::8, 3] = nan          # NaN tmax
    rays["o"][6::8, 3] = inf          # infinite tmax is an ordinary ray
    got, st = _check_all_modes(mipt, oracle, gpu_ctx, nodes, vtx, idx, rays, nan_equal=True)
    assert st.tie_rays >= 6 * 512
    assert st.box_tests / len(rays) < 200


def test_empty_and_error_paths(mipt, scenes, gpu_ctx):
    vtx, idx, _, _ = scenes.fallback_quad()
    nodes = _build(mipt, vtx, idx)
    sc = mipt.Scene(gpu_ctx, nodes, vtx, idx)
    try:
        assert sc.trace_closest(np.zeros(0, dtype=mipt.ray_dtype)).shape == (0,)
        assert sc.trace_any(np.zeros(0, dtype=mipt.ray_dtype)).shape == (0,)
    finally:
        sc.close()
    with pytest.raises(mipt.MiptError):
        mipt.Scene(gpu_ctx, nodes[:-1], vtx, idx)                 # truncated node array
    with pytest.raises(mipt.MiptError, match="outside the vertex buffer"):
        mipt.Scene(gpu_ctx, nodes, vtx[:3], idx)
    with pytest.raises(mipt.MiptError, match="out of range"):
        mipt.Context(10 ** 6)


def test_sponza_class_scene_primary_and_secondary(mipt, oracle, scenes, gpu_ctx):
    """262,144-triangle stand-in: 1/16 of a 1080p frame of primary rays plus incoherent
    surface rays, bit-exact; and the work counters the bench's roofline uses are sane."""
    vtx, idx, _, _ = scenes.atrium()
    nodes = _build(mipt, vtx, idx)
    cam = scenes.atrium_camera()
    rays = gpu_ctx.raygen(cam, 480, 270)
    sec = scenes.surface_rays(vtx, idx, 100000, seed=7).astype(mipt.ray_dtype)
    allrays = np.concatenate([rays, sec])
    want, cnt = oracle.closest(nodes, vtx, idx, allrays, nthreads=8, counters=True)
    sc = mipt.Scene(gpu_ctx, nodes, vtx, idx)
    try:
        got, st = sc.trace_closest(allrays, stats=True)
        assert len(same_hits(got, want)) == 0
        ex, st_ex = sc.trace_closest(allrays, mipt.PT_TRACE_EXACT_ORDER, stats=True)
        assert len(same_hits(ex, want)) == 0
        # the fixed-order kernel does exactly the oracle's work
        assert (st_ex.box_tests, st_ex.tri_tests) == (cnt.box_tests, cnt.tri_tests)
        # the ordered kernel tests fewer triangles (it reaches the nearest hit sooner)
        assert st.tri_tests < cnt.tri_tests
        print(f"atrium: oracle {cnt.box_tests / len(allrays):.1f} box + "
              f"{cnt.tri_tests / len(allrays):.1f} tri tests/ray; ordered "
              f"{st.box_tests / len(allrays):.1f} + {st.tri_tests / len(allrays):.1f}; "
              f"ties {st.tie_rays}")
    finally:
        sc.close()


def test_origin_cell_order_does_not_change_hits(mipt, oracle, scenes, gpu_ctx):
    """trace.sort_rays=1 takes the rays in Morton order of their origins (a locality aid for
    scenes beyond the caches); every hit lands at its ray's own index, unchanged."""
    vtx, idx, _, _ = scenes.random_soup(20000, seed=31)
    nodes = _build(mipt, vtx, idx)
    rays = random_rays(100003, 17, dtype=mipt.ray_dtype)           # not a multiple of 64
    rays["o"][5, 0] = np.nan                                        # a NaN origin sorts somewhere, too
    sc = mipt.Scene(gpu_ctx, nodes, vtx, idx)
    try:
        want = oracle.closest(nodes, vtx, idx, rays, nthreads=8)
        gpu_ctx.set_option("trace.sort_rays", 1)
        try:
            got = sc.trace_closest(rays)
        finally:
            gpu_ctx.set_option("trace.sort_rays", -1)
        assert len(same_hits(got, want, nan_equal=True)) == 0
    finally:
        sc.close()


def test_tree_with_spatial_splits_on_device(mipt, oracle, scenes, gpu_ctx):
    """A triangle held by several leaves (bvh.sah.use_splits): same hits as the oracle on that
    array, and its second leaf is not mistaken for a competing candidate (no re-resolution)."""
    from test_host_bvh import _sliver_scene
    vtx, idx = _sliver_scene(scenes, n=6000, seed=4)
    lb = mipt.mesh_leaf_bounds(vtx, idx)
    nodes = mipt.build_bvh(lb, mipt.bvh_options(**{"bvh.builder": "sah", "bvh.sah.use_splits": 1,
                                                   "bvh.sah.max_split_depth": 16,
                                                   "bvh.sah.min_overlap": 1e-4}))
    assert (nodes.shape[0] + 1) // 2 > idx.shape[0]
    rays = random_rays(100000, 23, dtype=mipt.ray_dtype)
    sc = mipt.Scene(gpu_ctx, nodes, vtx, idx)
    try:
        got, st = sc.trace_closest(rays, stats=True)
        want = oracle.closest(nodes, vtx, idx, rays, nthreads=8)
        assert len(same_hits(got, want)) == 0
        assert st.tie_rays < 20
        gpu_ctx.set_option("extend.persistent", 0)              # the per-thread loop, too
        try:
            got2, st2 = sc.trace_closest(rays, stats=True)
        finally:
            gpu_ctx.set_option("extend.persistent", 1)
        assert len(same_hits(got2, want)) == 0 and st2.tie_rays < 20
        rays["o"][:, 3] = 7.0
        np.testing.assert_array_equal(sc.trace_any(rays), oracle.any_hit(nodes, vtx, idx, rays, nthreads=8))
    finally:
        sc.close()


def test_two_contexts_driven_from_two_threads(mipt, oracle, scenes):
    """One context per host thread (the threading contract of the C ABI): both threads trace
    and render at the same time on the same GPU and get the single-threaded answers."""
    import threading
    vtx, idx, mats, mids = scenes.atrium(65536, seed=3)
    nodes = _build(mipt, vtx, idx)
    rays = scenes.surface_rays(vtx, idx, 200_000, seed=1).astype(mipt.ray_dtype)
    want = oracle.closest(nodes, vtx, idx, rays, nthreads=8)
    cam = scenes.atrium_camera(1.5)
    results, errors = {}, []

    def work(k):
        try:
            ctx = mipt.Context(0)
            sc = mipt.Scene(ctx, nodes, vtx, idx, mats, mids)
            film = mipt.Film(ctx, 96, 64)
            hits = [sc.trace_closest(rays) for _ in range(3)]
            mipt.render(ctx, sc, cam, mipt.render_params(96, 64, sample_count=4, max_depth=4, seed=5), film)
            results[k] = (hits, film.read())
            film.close(); sc.close(); ctx.close()
        except Exception as e:          # surfaced below: an exception in a thread is otherwise lost
            errors.append(e)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k in range(2):
        for h in results[k][0]:
            assert len(same_hits(h, want)) == 0
    np.testing.assert_array_equal(results[0][1].view(np.uint32), results[1][1].view(np.uint32))
