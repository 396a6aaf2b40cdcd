"""Two-level scenes on the CPU: the restatement's own consistency, and the library's host-only
helper against it.  (The reference documents the format, src/bvh/bvh_builder.h:29-33, but has
no working code for it: everything here is pinned by this project's own specification only.)"""
import numpy as np

import two_level_util as tl


def _build(mipt, oracle, sc):
    meshes = []
    for v, i, _ in sc["meshes"]:
        nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(v, i), mipt.bvh_options(**{"bvh.builder": "sah"}))
        meshes.append((nodes, v, i))
    t = oracle.TwoLevel(meshes, sc["transforms"], sc["mesh_of"])
    t.top = mipt.build_bvh(t.bounds(), mipt.bvh_options(**{"bvh.builder": "sah"}))
    return t


def test_inverse_and_bounds(mipt, oracle):
    sc = tl.scene()
    t = _build(mipt, oracle, sc)
    for m in sc["transforms"]:
        inv = oracle.invert_affine(m).astype(np.float64)
        full, finv = np.eye(4), np.eye(4)
        full[:3], finv[:3] = m, inv
        np.testing.assert_allclose(full @ finv, np.eye(4), atol=2e-6)
    # the library's host helper computes the same boxes, bit for bit
    meshes = [(n, v, i) for (n, v, i) in t.meshes]
    got = mipt.instance_bounds(meshes, sc["transforms"], sc["mesh_of"])
    np.testing.assert_array_equal(got.view(np.uint32), t.bounds().view(np.uint32))
    # and they contain every transformed vertex
    vtx, _, first = t.world_triangles()
    ends = list(first[1:] * 3) + [vtx.shape[0]]
    for k, (a, b) in enumerate(zip(first * 3, ends)):
        assert (vtx[a:b] >= got["pmin"][k, :3] - 1e-4).all() and (vtx[a:b] <= got["pmax"][k, :3] + 1e-4).all()


def test_identity_instance_is_the_flat_scene(mipt, oracle):
    """One instance with the identity transform: the two-level walk returns exactly what the
    single-level walk returns (object space = world space, bit for bit)."""
    v, i, _ = tl.blob_mesh(400, 3)
    nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(v, i))
    ident = np.zeros((1, 3, 4), dtype=np.float32)
    ident[0, :, :3] = np.eye(3)
    t = oracle.TwoLevel([(nodes, v, i)], ident, [0])
    t.top = mipt.build_bvh(t.bounds())
    # (without the non-finite rays: 0 * NaN spreads a NaN component to the whole object-space ray)
    rays = tl.rays_for(20000, 1, oracle.ray_dtype)[4:]
    rays["o"][:, :3] *= 0.3
    a = t.closest(rays)
    b = oracle.closest(nodes, v, i, rays)
    hit = b["primid"] >= 0
    assert hit.sum() > 2000
    np.testing.assert_array_equal(a["primid"], b["primid"])
    np.testing.assert_array_equal(a["shapeid"][hit], 0)
    np.testing.assert_array_equal(a["uvwt"][hit].view(np.uint32), b["uvwt"][hit].view(np.uint32))
    np.testing.assert_array_equal(t.any_hit(rays), oracle.any_hit(nodes, v, i, rays))


def test_two_level_agrees_with_the_flattened_scene(mipt, oracle):
    """General transforms: against a single-level scene of the world-space triangles the
    nearest triangle is the same except where rounding differs (object-space arithmetic), and
    distances agree closely."""
    sc = tl.scene()
    t = _build(mipt, oracle, sc)
    vtx, idx, first = t.world_triangles()
    flat_nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx))
    rays = tl.rays_for(30000, 2, oracle.ray_dtype)[4:]
    a, (nb, nt) = t.closest(rays, nthreads=8, counters=True)
    b = oracle.closest(flat_nodes, vtx, idx, rays, nthreads=8)
    glob = np.where(a["primid"] >= 0, first[np.maximum(a["shapeid"], 0)].astype(np.int64) + a["primid"], -1)
    hit = (glob >= 0) & (b["primid"] >= 0)
    assert hit.sum() > 10000 and nb > 0 and nt > 0
    assert ((glob >= 0) != (b["primid"] >= 0)).mean() < 2e-3
    np.testing.assert_allclose(a["uvwt"][hit, 3], b["uvwt"][hit, 3], rtol=2e-4, atol=1e-4)
    # the two coincident instances: which of them a ray reports is decided by the top-level
    # order and by how the range test rounds (the second copy is searched with the first
    # copy's distance as its limit); either way the triangle is the same one
    n_inst = len(sc["mesh_of"])
    dup = (n_inst - 4, n_inst - 3)
    assert sum((a["shapeid"] == d).sum() for d in dup) > 50
    same = glob[hit] == b["primid"][hit]
    in_dup = np.isin(a["shapeid"][hit], dup)
    assert same[~in_dup].mean() > 0.995
    # occlusion: consistent with the closest-hit query away from triangle edges
    occ = t.any_hit(rays, nthreads=8)
    assert ((occ != 0) != (a["primid"] >= 0)).mean() < 5e-3


def test_two_level_render_matches_flat_render_statistically(mipt, oracle):
    """The same image from the two-level walk and from the flattened scene, up to the rounding
    differences above (a handful of paths diverge): mean radiance within a few percent."""
    sc = tl.scene(n_blob_tris=120, n_blobs=6)
    t = _build(mipt, oracle, sc)
    vtx, idx, _ = t.world_triangles()
    mats, mids = tl.materials(oracle), tl.flat_material_ids(sc)
    flat_nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx))
    cam = mipt.look_at_camera((0.0, 3.0, -14.0), (0.0, 0.5, 0.0), fovy_deg=45.0, aspect=1.5)
    for flags in (0, mipt.PT_RENDER_NEE):
        p = mipt.render_params(48, 32, sample_count=8, max_depth=4, seed=11, flags=flags,
                               background=(0.05, 0.06, 0.08))
        a, ra = t.render(mats, mids, cam, p, nthreads=8)
        b, rb = oracle.render(flat_nodes, vtx, idx, mats, mids, cam, p, nthreads=8)
        assert abs(ra - rb) / rb < 0.01
        ma, mb = a[..., :3].mean(), b[..., :3].mean()
        assert mb > 0 and abs(ma - mb) / mb < 0.05
