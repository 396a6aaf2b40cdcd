"""Full-size configurations of the baseline, checked through the oracle on samples and
through size-independent properties on everything.

* config 4: San-Miguel-class scene (9.3 M triangles, 18.5 M nodes > 2^24, so the reference's
  float-valued links cannot express it), 3840x2160, one of 8 tile shards, primary + 1 bounce.
* config 5: Lucy-class closed surface (28 M triangles), tree built on the GPU, 2^22 incoherent
  secondary rays: both traversal orders, both ray orders, one answer; a sample against the CPU
  restatement walking the GPU-built tree.
* config 2: Cornell box 1920x1080, 64 spp, depth 4, Lambertian: the whole frame on the GPU,
  a random subset of pixels recomputed by the CPU restatement.
* config 3: Sponza-class scene (262 K triangles) 1920x1080, depth 8, Lambert + GGX: whole frames
  (a few samples per pixel, the bench's own parameters) against the CPU restatement, every pixel,
  without and with next-event estimation.
"""
import numpy as np
import pytest

from conftest import same_hits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big_scene(mipt, scenes):
    vtx, idx, mats, mids = scenes.clutter(9_000_000, seed=11)
    lb = mipt.mesh_leaf_bounds(vtx, idx)
    with pytest.raises(mipt.MiptError, match="2\\^24"):
        mipt.build_bvh(lb, mipt.bvh_options(**{"bvh.builder": "sah"}))    # float links cannot index it
    nodes = mipt.build_bvh(lb, mipt.bvh_options(**{"bvh.builder": "sah", "mipt.link_encoding": "int"}))
    assert nodes.shape[0] == 2 * idx.shape[0] - 1 > 2 ** 24
    return vtx, idx, mats, mids, nodes


def test_san_miguel_class_4k_tile_shard(mipt, oracle, scenes, gpu_ctx, big_scene):
    vtx, idx, mats, mids, nodes = big_scene
    W, H, TS, WORLD, RANK = 3840, 2160, 64, 8, 3
    sc = mipt.Scene(gpu_ctx, nodes, vtx, idx, mats, mids, link_encoding=mipt.PT_LINKS_INT_BITS)
    try:
        cam = mipt.look_at_camera((-17.0, 10.5, -16.0), (3.0, 1.0, 2.0), fovy_deg=50.0, aspect=W / H)   # above the blob layer
        # -- closest hits of this shard's camera rays plus incoherent surface rays
        xy = mipt.shard_layout(W, H, TS, RANK, WORLD)
        xy = xy[xy[:, 0] >= 0]
        all_rays = gpu_ctx.raygen(cam, W, H)
        rays = np.concatenate([all_rays[xy[:, 1].astype(np.int64) * W + xy[:, 0]],
                               scenes.surface_rays(vtx, idx, 200_000, seed=7).astype(mipt.ray_dtype)])
        del all_rays
        got, st = sc.trace_closest(rays, stats=True)
        exact = sc.trace_closest(rays, mipt.PT_TRACE_EXACT_ORDER)
        assert len(same_hits(got, exact)) == 0            # two traversal orders, one answer
        pick = np.random.default_rng(0).choice(len(rays), 150_000, replace=False)
        want = oracle.closest(nodes, vtx, idx, rays[pick], nthreads=16, int_bits=True)
        assert len(same_hits(got[pick], want)) == 0       # and it is the oracle's
        assert (got["primid"] >= 0).mean() > 0.9
        assert got["primid"].max() > 2 ** 23              # leaves beyond the float-exact range are hit
        print(f"9.3M tris: {st.box_tests / len(rays):.1f} box + {st.tri_tests / len(rays):.1f} tri "
              f"tests per ray, {st.tie_rays} ties of {len(rays)}")
        # -- the shard's part of the frame, primary + 1 bounce
        film = mipt.Film(gpu_ctx, W, H)
        p = mipt.render_params(W, H, sample_count=1, max_depth=2, seed=5, tile_size=TS,
                               shard_index=RANK, shard_count=WORLD, background=(0.5, 0.6, 0.8))
        rs = mipt.render(gpu_ctx, sc, cam, p, film, stats=True)
        img = film.read()
        owned = np.zeros((H, W), dtype=bool)
        owned[xy[:, 1], xy[:, 0]] = True
        assert (img[..., 3] == owned).all()               # exactly the owned tiles got one sample
        assert np.isfinite(img).all() and img[owned][:, :3].mean() > 0
        assert rs.rays_primary == owned.sum() and 0 < rs.rays_bounce <= rs.rays_primary
        # rendering the same samples again doubles the sums exactly (no hidden state)
        mipt.render(gpu_ctx, sc, cam, p, film)
        np.testing.assert_array_equal(film.read(), 2 * img)
        film.close()
    finally:
        sc.close()


def test_lucy_class_28m_secondary_rays_gpu_built_tree(mipt, oracle, scenes, gpu_ctx):
    vtx, idx, _, _ = scenes.displaced_sphere(28_000_000, seed=7)
    assert idx.shape[0] > 27_000_000
    sc = mipt.Scene(gpu_ctx, None, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS)   # pt_scene_build
    try:
        rays = scenes.surface_rays(vtx, idx, 1 << 22, seed=7).astype(mipt.ray_dtype)
        got, st = sc.trace_closest(rays, stats=True)      # > 256 MB of traversal data: origin-cell order
        gpu_ctx.set_option("trace.sort_rays", 0)
        try:
            plain = sc.trace_closest(rays)                # queue order
        finally:
            gpu_ctx.set_option("trace.sort_rays", -1)
        assert len(same_hits(got, plain)) == 0
        sub = np.random.default_rng(3).choice(len(rays), 300_000, replace=False)
        assert len(same_hits(sc.trace_closest(rays[sub], mipt.PT_TRACE_EXACT_ORDER), got[sub])) == 0
        assert 0.4 < (got["primid"] >= 0).mean() < 0.95
        assert got["primid"].max() > 2 ** 24
        nodes = sc.read_nodes()                           # the tree the GPU built, 58 M nodes
        # the slivers around the two poles are in several leaves each: an eighth more leaves
        leaves = (nodes.shape[0] + 1) // 2
        assert idx.shape[0] < leaves < 1.25 * idx.shape[0]
        pick = sub[:20_000]
        want = oracle.closest(nodes, vtx, idx, rays[pick], nthreads=16, int_bits=True)
        assert len(same_hits(got[pick], want)) == 0
        occ = sc.trace_any(rays[sub])
        assert ((occ != 0) == (got["primid"][sub] >= 0)).mean() > 0.999   # edge slack of the any-hit test only
        print(f"28M tris: {st.box_tests / len(rays):.1f} box + {st.tri_tests / len(rays):.1f} tri tests per ray, "
              f"{st.tie_rays} ties of {len(rays)}")
    finally:
        sc.close()


def test_scene_beyond_four_gib_of_records(mipt, oracle, scenes, gpu_ctx):
    """72 M triangles: 72 M two-child nodes (4.6 GB) and 3.5 GB of triangle records, beyond what
    32-bit byte offsets reach (2^26 nodes, 89 M records), so the kernels compiled for 64-bit
    record addresses run; 144 M skip-link nodes with integer links.  Tree built on the GPU; the
    persistent kernels in both ray orders, the fixed-order kernel and the occlusion query agree,
    and a sample equals the CPU restatement walking the tree read back from the device."""
    vtx, idx, _, _ = scenes.displaced_sphere(72_000_000, seed=9)
    assert idx.shape[0] - 1 > 2 ** 26
    sc = mipt.Scene(gpu_ctx, None, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS)
    try:
        rays = scenes.surface_rays(vtx, idx, 1 << 20, seed=11).astype(mipt.ray_dtype)
        got, st = sc.trace_closest(rays, stats=True)
        gpu_ctx.set_option("trace.sort_rays", 0)
        try:
            plain = sc.trace_closest(rays)
        finally:
            gpu_ctx.set_option("trace.sort_rays", -1)
        assert len(same_hits(got, plain)) == 0
        assert 0.4 < (got["primid"] >= 0).mean() < 0.95
        assert got["primid"].max() > 2 ** 26               # triangles whose records lie beyond 4 GiB
        pick = np.random.default_rng(5).choice(len(rays), 50_000, replace=False)
        assert len(same_hits(sc.trace_closest(rays[pick], mipt.PT_TRACE_EXACT_ORDER), got[pick])) == 0
        nodes = sc.read_nodes()
        assert idx.shape[0] <= (nodes.shape[0] + 1) // 2 < 1.25 * idx.shape[0]
        want = oracle.closest(nodes, vtx, idx, rays[pick[:20_000]], nthreads=16, int_bits=True)
        assert len(same_hits(got[pick[:20_000]], want)) == 0
        occ = sc.trace_any(rays[pick])
        assert ((occ != 0) == (got["primid"][pick] >= 0)).mean() > 0.999
        print(f"72M tris: {st.box_tests / len(rays):.1f} box + {st.tri_tests / len(rays):.1f} tri tests per ray")
    finally:
        sc.close()


def test_cornell_1080p_64spp_depth4_full_size(mipt, oracle, scenes, gpu_ctx):
    vtx, idx, mats, mids = scenes.cornell_box()
    nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx), mipt.bvh_options(**{"bvh.builder": "sah"}))
    W, H = 1920, 1080
    cam = mipt.look_at_camera((278.0, 273.0, -800.0), (278.0, 273.0, 0.0), fovy_deg=39.3, aspect=W / H)
    sc = mipt.Scene(gpu_ctx, nodes, vtx, idx, mats, mids)
    film = mipt.Film(gpu_ctx, W, H)
    try:
        p = mipt.render_params(W, H, sample_count=64, max_depth=4, seed=1)
        st = mipt.render(gpu_ctx, sc, cam, p, film, stats=True)
        img = film.read()
        assert (img[..., 3] == 64).all() and np.isfinite(img).all()
        assert st.paths == W * H * 64
        rng = np.random.default_rng(1)
        ref = np.zeros_like(img)
        px = np.stack([rng.integers(0, W, 600), rng.integers(0, H, 600)], 1)
        for x, y in np.unique(px, axis=0):
            oracle.render(nodes, vtx, idx, mats, mids, cam, p, film=ref, rect=(x, y, x + 1, y + 1))
        sel = ref[..., 3] > 0
        assert sel.sum() >= 590
        err = np.sqrt(np.mean((img[sel][:, :3] - ref[sel][:, :3]) ** 2)) / np.sqrt(np.mean(ref[sel][:, :3] ** 2))
        assert err <= 1e-3                                 # the stated tolerance
        np.testing.assert_array_equal(img[sel].view(np.uint32), ref[sel].view(np.uint32))   # measured: exact
        print(f"cornell 1080p 64spp: {st.rays_primary + st.rays_bounce} rays, {st.tie_rays} ties, "
              f"{sel.sum()} pixels recomputed on the CPU bit-identical")
    finally:
        film.close()
        sc.close()


def test_millions_of_rays_with_ties_and_deep_trees_match_the_oracle(mipt, oracle, scenes, gpu_ctx):
    """Bit-exact hits at a ray count where rare rounding cases show up: 2 M incoherent rays on
    the benchmark scene with a tenth of its triangles duplicated (150 000 exact ties, re-walked
    in batches from a bounded start; flat wall triangles have boxes without depth, whose entry
    distance can round past the hit distance), and on a GPU-built, GPU-optimised million-triangle
    tree read back from the device."""
    n = 1 << 21
    vtx, idx, _, _ = scenes.atrium(262144, seed=42)
    rng = np.random.default_rng(3)
    idx2 = np.concatenate([idx, idx[rng.choice(idx.shape[0], idx.shape[0] // 10, replace=False)]])
    nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx2), mipt.bvh_options(**{"bvh.builder": "sah"}))
    rays = scenes.surface_rays(vtx, idx, n, seed=5).astype(mipt.ray_dtype)
    sc = mipt.Scene(gpu_ctx, nodes, vtx, idx2)
    try:
        got, st = sc.trace_closest(rays, stats=True)
        assert st.tie_rays > 100_000
        assert len(same_hits(got, oracle.closest(nodes, vtx, idx2, rays, nthreads=16))) == 0
    finally:
        sc.close()
    vtx, idx, _, _ = scenes.clutter(1_000_000, seed=11)
    rays = scenes.surface_rays(vtx, idx, n, seed=9).astype(mipt.ray_dtype)
    gpu_ctx.set_option("build.optimize_passes", 16)
    try:
        sc = mipt.Scene(gpu_ctx, None, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS)
    finally:
        gpu_ctx.set_option("build.optimize_passes", 6)
    try:
        nodes = sc.read_nodes()
        got = sc.trace_closest(rays)
        assert len(same_hits(got, oracle.closest(nodes, vtx, idx, rays, nthreads=16, int_bits=True))) == 0
    finally:
        sc.close()


def test_awkward_rays_match_the_oracle(mipt, oracle, scenes, gpu_ctx):
    """A million rays built to sit on the scene's special cases: along the axes through vertices
    and edge midpoints of the axis-aligned walls, from vertex to vertex inside the planes of the
    walls, from points on surfaces, from a million units away, and a fifth of them with a reach
    that ends exactly on the surface aimed at.  The last kind is where the fixed order matters in
    a way no pair of near-equal hits shows: a distance that rounds to an ulp beyond the reach can
    be accepted, after which the shader culls at that distance and lets in a triangle it would
    have turned down at the reach itself."""
    vtx, idx, _, _ = scenes.atrium(262144, seed=42)
    nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx), mipt.bvh_options(**{"bvh.builder": "sah"}))
    rays = scenes.awkward_rays(vtx, idx, 1 << 20, seed=21).astype(mipt.ray_dtype)
    sc = mipt.Scene(gpu_ctx, nodes, vtx, idx)
    try:
        want = oracle.closest(nodes, vtx, idx, rays, nthreads=16)
        for persistent in (1, 0):
            gpu_ctx.set_option("extend.persistent", persistent)
            got, st = sc.trace_closest(rays, stats=True)
            assert len(same_hits(got, want)) == 0
            assert st.tie_rays > 100_000 and 0.6 < (got["primid"] >= 0).mean() < 0.95
        want_any = oracle.any_hit(nodes, vtx, idx, rays, nthreads=16)
        np.testing.assert_array_equal(sc.trace_any(rays), want_any)
        # The four-child walk (what scenes beyond the caches get) on the same rays.  Occlusion
        # rays along an axis are the case its coarse boxes cannot decide: a ray lying in a
        # bounding plane of an inner box misses that box in the reference (0 * inf on one side of
        # the slab) but not a flat leaf box in the same plane, where the Moller-Trumbore test
        # then reports a hit.
        gpu_ctx.set_option("extend.quad", 1)
        try:
            assert len(same_hits(sc.trace_closest(rays), want)) == 0
            np.testing.assert_array_equal(sc.trace_any(rays), want_any)
        finally:
            gpu_ctx.set_option("extend.quad", -1)
        # the shader's Moller-Trumbore closest hit clamps the distance with max(0, t): +0 for t = -0
        mt = sc.trace_closest(rays, mipt.PT_TRACE_MOLLER_TRUMBORE)
        want_mt = oracle.closest(nodes, vtx, idx, rays, nthreads=16, moller_trumbore=True)
        assert len(same_hits(mt, want_mt, True)) == 0
        assert (mt["uvwt"][:, 3].view(np.uint32) == 0x80000000).sum() == 0
    finally:
        gpu_ctx.set_option("extend.persistent", 1)
        sc.close()


def test_awkward_rays_in_a_box_of_axis_aligned_walls(mipt, oracle, scenes, gpu_ctx):
    """The same kinds of rays in the Cornell box, where every surface lies in an axis plane and
    every box of the tree has faces in those planes: closest hits and occlusion, two-child and
    four-child walk."""
    vtx, idx, _, _ = scenes.cornell_box()
    nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx), mipt.bvh_options(**{"bvh.builder": "sah"}))
    rays = scenes.awkward_rays(vtx, idx, 1 << 18, seed=44).astype(mipt.ray_dtype)
    sc = mipt.Scene(gpu_ctx, nodes, vtx, idx)
    try:
        want = oracle.closest(nodes, vtx, idx, rays, nthreads=16)
        want_any = oracle.any_hit(nodes, vtx, idx, rays, nthreads=16)
        assert 0.05 < want_any.mean() < 0.99
        for quad in (0, 1):
            gpu_ctx.set_option("extend.quad", quad)
            assert len(same_hits(sc.trace_closest(rays), want)) == 0
            np.testing.assert_array_equal(sc.trace_any(rays), want_any)
    finally:
        gpu_ctx.set_option("extend.quad", -1)
        sc.close()


def test_sponza_class_1080p_depth8_whole_frame(mipt, oracle, scenes, gpu_ctx):
    """The benchmark's frame at its full size -- same scene, tree options, camera, depth, seed,
    background and Russian-roulette depth as bench.py -- against the CPU restatement of the same
    step: every bit of every pixel, the ray counts too.  Two samples per pixel without and one
    with next-event estimation (27 M rays in all); with the default split of the workgroups'
    LDS and with the whole of it given to the stacks."""
    vtx, idx, mats, mids = scenes.atrium(262144, seed=42)
    nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx),
                           mipt.bvh_options(**{"bvh.builder": "sah", "mipt.optimize_passes": "2"}))
    W, H = 1920, 1080
    cam = scenes.atrium_camera(W / H)
    sc = mipt.Scene(gpu_ctx, nodes, vtx, idx, mats, mids)
    film = mipt.Film(gpu_ctx, W, H)
    try:
        for flags, spp in ((0, 2), (mipt.PT_RENDER_NEE, 1)):
            p = mipt.render_params(W, H, sample_begin=16, sample_count=spp, max_depth=8, seed=42,
                                   rr_start_depth=3, background=(0.55, 0.65, 0.8), flags=flags)
            want, rays = oracle.render(nodes, vtx, idx, mats, mids, cam, p, nthreads=16)
            for cache_nodes in (-1, 0):
                gpu_ctx.set_option("extend.cache_nodes", cache_nodes)
                gpu_ctx.set_option("extend.stack_entries", -1 if cache_nodes < 0 else 30)
                film.clear()
                st = mipt.render(gpu_ctx, sc, cam, p, film, stats=True)
                assert st.rays_primary + st.rays_bounce + st.rays_shadow == rays
                got = film.read()
                diff = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=-1).sum())
                assert diff == 0, f"{diff} of {W * H} pixels differ from the CPU restatement"
            print(f"1080p depth 8 {'with' if flags else 'without'} NEE: {rays} rays, film bit-identical")
    finally:
        gpu_ctx.set_option("extend.cache_nodes", -1)
        gpu_ctx.set_option("extend.stack_entries", -1)
        film.close()
        sc.close()
