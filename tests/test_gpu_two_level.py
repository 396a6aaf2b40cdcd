"""Two-level scenes on the GPU against the CPU restatement, through the C ABI: hits bit for
bit (instance, triangle, barycentrics, distance), occlusion, and the path tracer's film.  The
specification is this project's own (the reference only documents the node format,
src/bvh/bvh_builder.h:29-33): parity unpinned."""
import numpy as np
import pytest

import two_level_util as tl
from conftest import same_hits
from test_gpu_render import REL_RMSE_TOL, rel_rmse

pytestmark = pytest.mark.gpu


def _build(mipt, oracle, ctx, sc, builder="sah"):
    opt = mipt.bvh_options(**{"bvh.builder": builder})
    meshes = []
    for v, i, mids in sc["meshes"]:
        meshes.append((mipt.build_bvh(mipt.mesh_leaf_bounds(v, i), opt), v, i, mids))
    top = mipt.build_bvh(mipt.instance_bounds(meshes, sc["transforms"], sc["mesh_of"]), opt)
    t = oracle.TwoLevel([m[:3] for m in meshes], sc["transforms"], sc["mesh_of"], top)
    scene = mipt.Scene.instanced(ctx, meshes, sc["transforms"], sc["mesh_of"], top,
                                 materials=tl.materials(mipt),
                                 material_offsets=sc["material_offsets"])
    return t, scene


@pytest.mark.parametrize("builder", ["sah", "median"])
def test_closest_and_any_hits_are_bit_exact(mipt, oracle, gpu_ctx, builder):
    sc = tl.scene()
    t, scene = _build(mipt, oracle, gpu_ctx, sc, builder)
    try:
        assert scene.instance_count() == len(sc["mesh_of"])
        rays = tl.rays_for(60000, 4, mipt.ray_dtype)
        want, (nb, nt) = t.closest(rays, nthreads=8, counters=True)
        got, st = scene.trace_closest(rays, stats=True)
        bad = same_hits(got, want, nan_equal=True)
        assert bad.size == 0, f"{bad.size} rays differ, first {bad[:5]}"
        assert (got["primid"] >= 0).sum() > 20000
        assert len(set(got["shapeid"][got["primid"] >= 0])) >= len(sc["mesh_of"]) - 1
        assert st.box_tests > 0 and st.tri_tests > 0
        print(f"{st.tie_rays} tie rays; boxes/ray {st.box_tests / rays.shape[0]:.1f} "
              f"(fixed order {nb / rays.shape[0]:.1f}), triangles/ray {st.tri_tests / rays.shape[0]:.2f} "
              f"(fixed order {nt / rays.shape[0]:.2f})")
        exact = scene.trace_closest(rays, flags=mipt.PT_TRACE_EXACT_ORDER)
        assert same_hits(exact, want, nan_equal=True).size == 0
        np.testing.assert_array_equal(scene.trace_any(rays), t.any_hit(rays, nthreads=8))
        gpu_ctx.set_option("any.exact_order", 1)
        try:
            np.testing.assert_array_equal(scene.trace_any(rays), t.any_hit(rays, nthreads=8))
        finally:
            gpu_ctx.set_option("any.exact_order", 0)
        # the top of the top-level tree cached in LDS (off by default on two-level scenes), with few
        # stack entries left per lane: same answers
        gpu_ctx.set_option("extend.cache_nodes", 160)
        gpu_ctx.set_option("extend.stack_entries", 12)
        try:
            assert same_hits(scene.trace_closest(rays), want, nan_equal=True).size == 0
            np.testing.assert_array_equal(scene.trace_any(rays), t.any_hit(rays, nthreads=8))
        finally:
            gpu_ctx.set_option("extend.cache_nodes", -1)
            gpu_ctx.set_option("extend.stack_entries", -1)
        # one thread per ray instead of persistent waves: same answers
        gpu_ctx.set_option("extend.persistent", 0)
        try:
            assert same_hits(scene.trace_closest(rays), want, nan_equal=True).size == 0
            np.testing.assert_array_equal(scene.trace_any(rays), t.any_hit(rays, nthreads=8))
        finally:
            gpu_ctx.set_option("extend.persistent", 1)
        # the meshes' trees as four-child nodes with quantised boxes (what scenes whose meshes lie
        # beyond the caches are walked by): every record the same again, also with few stack entries
        # (a four-child step pushes up to three) and through the rays along the axes
        for entries in (-1, 12):
            gpu_ctx.set_option("extend.quad", 1)
            gpu_ctx.set_option("extend.stack_entries", entries)
            try:
                quad, stq = scene.trace_closest(rays, stats=True)
                assert same_hits(quad, want, nan_equal=True).size == 0
                assert 0 < stq.box_tests and stq.tri_tests <= st.tri_tests
                np.testing.assert_array_equal(scene.trace_any(rays), t.any_hit(rays, nthreads=8))
                axis = rays[:20000].copy()
                axis["d"][:, :3] = np.eye(3, dtype=np.float32)[np.arange(len(axis)) % 3] * \
                    np.where(np.arange(len(axis)) % 2, -1.0, 1.0).astype(np.float32)[:, None]
                assert same_hits(scene.trace_closest(axis), t.closest(axis, nthreads=8), nan_equal=True).size == 0
                np.testing.assert_array_equal(scene.trace_any(axis), t.any_hit(axis, nthreads=8))
            finally:
                gpu_ctx.set_option("extend.quad", -1)
                gpu_ctx.set_option("extend.stack_entries", -1)
        # a batch smaller than one wave, and an empty one
        assert same_hits(scene.trace_closest(rays[:37]), want[:37], nan_equal=True).size == 0
        assert scene.trace_closest(rays[:0]).shape[0] == 0
        with pytest.raises(mipt.MiptError, match="two-level"):
            scene.trace_closest(rays[:8], flags=mipt.PT_TRACE_MOLLER_TRUMBORE)
        with pytest.raises(mipt.MiptError, match="two-level"):
            scene.read_nodes()
    finally:
        scene.close()


@pytest.mark.parametrize("flags", [0, "nee"])
def test_render_matches_the_cpu_restatement(mipt, oracle, gpu_ctx, flags):
    sc = tl.scene(n_blob_tris=200, n_blobs=7)
    t, scene = _build(mipt, oracle, gpu_ctx, sc)
    W, H = 96, 64
    film = mipt.Film(gpu_ctx, W, H)
    try:
        cam = mipt.look_at_camera((0.0, 3.0, -14.0), (0.0, 0.5, 0.0), fovy_deg=45.0, aspect=W / H)
        p = mipt.render_params(W, H, sample_count=6, max_depth=5, seed=21,
                               flags=mipt.PT_RENDER_NEE if flags else 0,
                               background=(0.05, 0.06, 0.08))
        st = mipt.render(gpu_ctx, scene, cam, p, film, stats=True)
        got = film.read()
        want, rays = t.render(tl.materials(oracle), tl.flat_material_ids(sc), cam, p, nthreads=8)
        err = rel_rmse(got[..., :3], want[..., :3])
        assert err <= REL_RMSE_TOL, f"relative RMSE {err}"
        assert st.rays_primary + st.rays_bounce + st.rays_shadow == rays
        n_diff = int((got.view(np.uint32) != want.view(np.uint32)).sum())
        assert n_diff == 0, f"film differs from the CPU restatement in {n_diff} words"
        assert got[..., :3].mean() > 0.01
        # the same frame with the meshes walked as four-child nodes (closest hits and shadow rays)
        gpu_ctx.set_option("extend.quad", 1)
        try:
            film.clear()
            stq = mipt.render(gpu_ctx, scene, cam, p, film, stats=True)
            assert int((film.read().view(np.uint32) != want.view(np.uint32)).sum()) == 0
            assert stq.rays_primary + stq.rays_bounce + stq.rays_shadow == rays
        finally:
            gpu_ctx.set_option("extend.quad", -1)
        if flags:
            assert st.rays_shadow > 0
        # the reference's grey view of the same scene
        pd = mipt.render_params(W, H, flags=mipt.PT_RENDER_DEPTH_GREY)
        film.clear()
        mipt.render(gpu_ctx, scene, cam, pd, film)
        grey, _ = t.render(tl.materials(oracle), tl.flat_material_ids(sc), cam, pd, nthreads=8)
        np.testing.assert_array_equal(film.read().view(np.uint32), grey.view(np.uint32))
    finally:
        film.close()
        scene.close()


def test_single_identity_instance_equals_the_flat_scene(mipt, oracle, scenes, gpu_ctx):
    """One instance, identity transform: object space is world space, so the two-level scene
    returns the flat scene's hits bit for bit (its one-leaf top level names the instance in
    both children of its only two-child node)."""
    vtx, idx, _ = tl.blob_mesh(500, 12)
    nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx))
    ident = np.zeros((1, 3, 4), dtype=np.float32)
    ident[0, :, :3] = np.eye(3)
    flat = mipt.Scene(gpu_ctx, nodes, vtx, idx)
    top = mipt.build_bvh(mipt.instance_bounds([(nodes, vtx, idx)], ident, [0]))
    assert top.shape[0] == 1
    two = mipt.Scene.instanced(gpu_ctx, [(nodes, vtx, idx)], ident, [0], top)
    try:
        rays = tl.rays_for(30000, 3, mipt.ray_dtype)[4:]
        rays["o"][:, :3] *= 0.3
        a, b = two.trace_closest(rays), flat.trace_closest(rays)
        hit = b["primid"] >= 0
        assert hit.sum() > 3000
        np.testing.assert_array_equal(a["primid"], b["primid"])
        np.testing.assert_array_equal(a["uvwt"][hit].view(np.uint32), b["uvwt"][hit].view(np.uint32))
        np.testing.assert_array_equal(a["shapeid"], np.where(hit, 0, -1))
        np.testing.assert_array_equal(two.trace_any(rays), flat.trace_any(rays))
    finally:
        two.close()
        flat.close()


@pytest.mark.parametrize("encoding", ["float", "int"])
def test_trees_built_by_the_library(mipt, oracle, gpu_ctx, encoding):
    """No tree from the caller: meshes built on the GPU (the single-triangle mesh and the
    quad included), top level built by the library.  The CPU restatement walks the trees read
    back from the scene and must agree bit for bit; hits also agree with the scene whose trees
    the host builder made, wherever no two candidates tie."""
    sc = tl.scene()
    enc = mipt.PT_LINKS_INT_BITS if encoding == "int" else mipt.PT_LINKS_FLOAT_VALUE
    meshes = [(None, v, i, m) for v, i, m in sc["meshes"]]
    scene = mipt.Scene.instanced(gpu_ctx, meshes, sc["transforms"], sc["mesh_of"], None,
                                 materials=tl.materials(mipt), material_offsets=sc["material_offsets"],
                                 link_encoding=enc)
    t_ref, ref_scene = _build(mipt, oracle, gpu_ctx, sc)
    try:
        top = scene.read_top_nodes()
        trees = [scene.read_mesh_nodes(k) for k in range(len(meshes))]
        for k, (_, v, i, _) in enumerate(meshes):
            # one leaf per triangle, or several for long thin ones lying askew in their boxes
            assert i.shape[0] <= (trees[k].shape[0] + 1) // 2 <= 1.5 * i.shape[0] + 64
            mipt.validate_bvh(trees[k], i.shape[0], enc)
        mipt.validate_bvh(top, len(sc["mesh_of"]), enc)
        rays = tl.rays_for(40000, 8, mipt.ray_dtype)
        got = scene.trace_closest(rays)
        assert same_hits(scene.trace_closest(rays, flags=mipt.PT_TRACE_EXACT_ORDER), got, nan_equal=True).size == 0
        if encoding == "float":       # the restatement reads float-valued links only
            t = oracle.TwoLevel([(trees[k], v, i) for k, (_, v, i, _) in enumerate(meshes)],
                                sc["transforms"], sc["mesh_of"], top)
            assert same_hits(got, t.closest(rays, nthreads=8), nan_equal=True).size == 0
            np.testing.assert_array_equal(scene.trace_any(rays), t.any_hit(rays, nthreads=8))
            gpu_ctx.set_option("extend.quad", 1)      # four-child nodes over trees with split leaves
            try:
                assert same_hits(scene.trace_closest(rays), got, nan_equal=True).size == 0
                np.testing.assert_array_equal(scene.trace_any(rays), t.any_hit(rays, nthreads=8))
            finally:
                gpu_ctx.set_option("extend.quad", -1)
        # against the host-built trees: same instance and triangle except at ties
        other = ref_scene.trace_closest(rays)
        same = (got["shapeid"] == other["shapeid"]) & (got["primid"] == other["primid"])
        assert same[4:].mean() > 0.99
        hit = (got["primid"] >= 0) & (other["primid"] >= 0)
        np.testing.assert_allclose(got["uvwt"][hit, 3][4:], other["uvwt"][hit, 3][4:], rtol=1e-5)
        # and the path tracer runs on it
        film = mipt.Film(gpu_ctx, 64, 48)
        cam = mipt.look_at_camera((0.0, 3.0, -14.0), (0.0, 0.5, 0.0), fovy_deg=45.0, aspect=64 / 48)
        st = mipt.render(gpu_ctx, scene, cam, mipt.render_params(64, 48, sample_count=4, max_depth=4, seed=2,
                                                                 flags=mipt.PT_RENDER_NEE), film, stats=True)
        assert st.rays_shadow > 0 and np.isfinite(film.read()).all() and film.read()[..., :3].mean() > 0
        film.close()
    finally:
        scene.close()
        ref_scene.close()


def test_integer_bit_links(mipt, oracle, gpu_ctx):
    """All trees in the integer-bit-pattern link encoding (src/bvh/bvh_builder.h:29-33 as
    documented): same hits as with float-valued links."""
    sc = tl.scene(n_blob_tris=150, n_blobs=5)
    t, ref_scene = _build(mipt, oracle, gpu_ctx, sc)
    opt = mipt.bvh_options(**{"bvh.builder": "sah", "mipt.link_encoding": "int"})
    meshes = [(mipt.build_bvh(mipt.mesh_leaf_bounds(v, i), opt), v, i, m) for v, i, m in sc["meshes"]]
    top = mipt.build_bvh(mipt.instance_bounds(meshes, sc["transforms"], sc["mesh_of"]), opt)
    scene = mipt.Scene.instanced(gpu_ctx, meshes, sc["transforms"], sc["mesh_of"], top,
                                 materials=tl.materials(mipt), material_offsets=sc["material_offsets"],
                                 link_encoding=mipt.PT_LINKS_INT_BITS)
    try:
        rays = tl.rays_for(20000, 6, mipt.ray_dtype)
        want = t.closest(rays, nthreads=8)
        for flags in (mipt.PT_TRACE_DEFAULT, mipt.PT_TRACE_EXACT_ORDER):
            assert same_hits(scene.trace_closest(rays, flags=flags), want, nan_equal=True).size == 0
        np.testing.assert_array_equal(scene.trace_any(rays), t.any_hit(rays, nthreads=8))
    finally:
        scene.close()
        ref_scene.close()


def test_bad_input_is_rejected(mipt, oracle, gpu_ctx):
    sc = tl.scene(n_blob_tris=40, n_blobs=2)
    opt = mipt.bvh_options()
    meshes = [(mipt.build_bvh(mipt.mesh_leaf_bounds(v, i), opt), v, i, m) for v, i, m in sc["meshes"]]
    top = mipt.build_bvh(mipt.instance_bounds(meshes, sc["transforms"], sc["mesh_of"]), opt)
    mats = tl.materials(mipt)

    def make(**kw):
        args = dict(meshes=meshes, transforms=sc["transforms"], mesh_of=sc["mesh_of"], top_nodes=top,
                    materials=mats, material_offsets=sc["material_offsets"])
        args.update(kw)
        return mipt.Scene.instanced(gpu_ctx, **args)

    with pytest.raises(mipt.MiptError, match="names mesh"):
        make(mesh_of=[99] + list(sc["mesh_of"][1:]))
    flat = sc["transforms"].copy()
    flat[2, :, 1] = 0                                      # a transform that squashes y
    with pytest.raises(mipt.MiptError, match="not invertible"):
        make(transforms=flat)
    with pytest.raises(mipt.MiptError, match="material"):
        make(material_offsets=[5] * len(sc["mesh_of"]))
    with pytest.raises(mipt.MiptError, match="in no top-level leaf|references triangle|leaf"):
        make(top_nodes=mipt.build_bvh(mipt.instance_bounds(meshes, sc["transforms"][:-1], sc["mesh_of"][:-1]), opt))
    make().close()
