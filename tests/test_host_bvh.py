"""Host side of the boundary, on the CPU: builder, linearizer, options, OBJ reader.
The product's skip-link arrays are checked structurally, against the oracle's linearizer on
the same tree, and functionally (oracle traversal over product nodes == linear scan)."""

import numpy as np
import pytest

from conftest import random_rays, same_hits


def _tree_arrays_from_nodes(nodes):
    """Recover (left, right, prim) per node from a skip-link array (pre-order numbering)."""
    n = nodes.shape[0]
    payload, link = nodes["pmin"][:, 3].astype(np.int64), nodes["pmax"][:, 3].astype(np.int64)
    left = np.full(n, -1, np.int32)
    right = np.full(n, -1, np.int32)
    prim = np.full(n, -1, np.int32)
    for i in range(n):
        if payload[i] == -1:
            left[i], right[i] = i + 1, link[i + 1]
        else:
            prim[i] = payload[i]
    return left, right, prim


@pytest.mark.parametrize("builder", ["sah", "median"])
@pytest.mark.parametrize("n_tri,seed", [(1, 0), (2, 1), (5, 2), (333, 3), (5000, 4)])
def test_build_bvh_structure_and_traversal(mipt, oracle, scenes, n_tri, seed, builder):
    vtx, idx, _, _ = scenes.random_soup(n_tri, seed=seed)
    lb = mipt.mesh_leaf_bounds(vtx, idx)
    np.testing.assert_array_equal(lb.view(np.uint32), oracle.mesh_leaf_bounds(vtx, idx).view(np.uint32))
    nodes = mipt.build_bvh(lb, mipt.bvh_options(**{"bvh.builder": builder}))
    assert nodes.shape[0] == 2 * n_tri - 1
    mipt.validate_bvh(nodes, n_tri)
    payload, link = nodes["pmin"][:, 3], nodes["pmax"][:, 3]
    leaf = payload != -1.0
    assert sorted(payload[leaf].astype(int)) == list(range(n_tri))
    assert link[0] == -1.0
    # leaf boxes = inflated triangle boxes, bit for bit (bvh_builder.cpp:53-59)
    infl = oracle.inflate(lb)
    li = np.nonzero(leaf)[0]
    t = payload[li].astype(int)
    np.testing.assert_array_equal(nodes["pmin"][li, :3], infl["pmin"][t, :3])
    np.testing.assert_array_equal(nodes["pmax"][li, :3], infl["pmax"][t, :3])
    # inner boxes = union of their children's
    left, right, _ = _tree_arrays_from_nodes(nodes)
    for i in np.nonzero(~leaf)[0][:200]:
        lo = np.minimum(nodes["pmin"][left[i], :3], nodes["pmin"][right[i], :3])
        hi = np.maximum(nodes["pmax"][left[i], :3], nodes["pmax"][right[i], :3])
        np.testing.assert_array_equal(nodes["pmin"][i, :3], lo)
        np.testing.assert_array_equal(nodes["pmax"][i, :3], hi)
    # the reference traversal over these nodes finds what a linear scan finds
    from oracle.oracle import ray_dtype
    rays = random_rays(3000, seed + 7, dtype=ray_dtype)
    a = oracle.closest(nodes, vtx, idx, rays, nthreads=4)
    b = oracle.brute_closest(vtx, idx, rays, nthreads=4)
    assert len(same_hits(a, b)) == 0


def test_linearizer_matches_oracle_on_same_tree(mipt, oracle, scenes):
    """PlainBvhTranslator::Process restated twice (product C++ and oracle C) over one topology."""
    vtx, idx, _, _ = scenes.random_soup(777, seed=21)
    nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx), mipt.bvh_options(**{"bvh.builder": "sah"}))
    left, right, prim = _tree_arrays_from_nodes(nodes)
    bounds = nodes.copy()
    bounds["pmin"][:, 3] = 0
    bounds["pmax"][:, 3] = 0
    # shuffle the node numbering so neither side can rely on the input already being pre-order
    rng = np.random.default_rng(5)
    n = nodes.shape[0]
    perm = np.concatenate([[0], 1 + rng.permutation(n - 1)])      # new position of old node i
    inv = np.argsort(perm)
    b2, l2, r2, p2 = bounds[inv], left[inv], right[inv], prim[inv]
    l2 = np.where(l2 >= 0, perm[np.maximum(l2, 0)], -1).astype(np.int32)
    r2 = np.where(r2 >= 0, perm[np.maximum(r2, 0)], -1).astype(np.int32)
    got = mipt.linearize(b2, l2, r2, p2)
    want = oracle.linearize(b2, l2, r2, p2)
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
    np.testing.assert_array_equal(got.view(np.uint32), nodes.view(np.uint32))


def test_build_is_deterministic_across_thread_counts(mipt, scenes):
    vtx, idx, _, _ = scenes.random_soup(70000, seed=3)
    lb = mipt.mesh_leaf_bounds(vtx, idx)
    a = mipt.build_bvh(lb, mipt.bvh_options(**{"bvh.builder": "sah", "mipt.num_threads": 1}))
    b = mipt.build_bvh(lb, mipt.bvh_options(**{"bvh.builder": "sah", "mipt.num_threads": 8}))
    np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))


def test_degenerate_inputs(mipt, oracle):
    # all triangles identical: centroid bounds are a point -> index-median fallback
    vtx = np.tile(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), (9, 1))
    idx = np.arange(27, dtype=np.int32).reshape(-1, 3)
    nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx), mipt.bvh_options(**{"bvh.builder": "sah"}))
    mipt.validate_bvh(nodes, 9)
    # zero-area triangle (a point) still gets a leaf
    vtx = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [5, 5, 5]], np.float32)
    idx = np.array([[0, 1, 2], [3, 3, 3]], np.int32)
    mipt.validate_bvh(mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx)), 2)
    # errors: empty input, NaN bounds, index out of range
    with pytest.raises(mipt.MiptError):
        mipt.build_bvh(np.zeros(0, dtype=mipt.bbox_dtype))
    bad = np.zeros(2, dtype=mipt.bbox_dtype)
    bad["pmin"][0, 0] = np.nan
    with pytest.raises(mipt.MiptError, match="not finite"):
        mipt.build_bvh(bad)
    with pytest.raises(mipt.MiptError, match="outside the vertex buffer"):
        mipt.mesh_leaf_bounds(vtx, np.array([[0, 1, 9]], np.int32))


def test_options_mirror_reference_keys(mipt):
    opt = mipt.bvh_options()
    # defaults of src/bvh/bvh_builder.cpp:26-32
    assert (opt.use_sah, opt.use_splits, opt.max_split_depth, opt.num_bins) == (0, 0, 10, 64)
    assert (opt.min_overlap, opt.traversal_cost, opt.extra_node_budget) == (
        pytest.approx(0.05), pytest.approx(10.0), pytest.approx(0.5))
    opt = mipt.bvh_options(**{"bvh.builder": "sah", "bvh.sah.num_bins": 16,
                              "bvh.sah.traversal_cost": 4, "bvh.sah.use_splits": 0})
    assert (opt.use_sah, opt.num_bins, opt.traversal_cost, opt.use_splits) == (1, 16, 4.0, 0)
    with pytest.raises(mipt.MiptError, match="unknown option"):
        mipt.bvh_options(**{"bvh.nope": 1})
    with pytest.raises(mipt.MiptError, match="not a number"):
        mipt.bvh_options(**{"bvh.sah.num_bins": "many"})
    # spatial splits select the other builder (see test_spatial_splits)
    lb = np.zeros(2, dtype=mipt.bbox_dtype)
    lb["pmax"][:, :3] = 1
    assert mipt.build_bvh(lb, mipt.bvh_options(**{"bvh.sah.use_splits": 1})).shape[0] >= 3


def test_validate_rejects_broken_arrays(mipt, scenes):
    vtx, idx, _, _ = scenes.random_soup(50, seed=1)
    nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx))
    mipt.validate_bvh(nodes, 50)
    for mutate in (lambda a: a["pmax"].__setitem__((0, 3), 5.0),            # root must link to -1
                   lambda a: a["pmax"].__setitem__((3, 3), 1.0),            # backwards link
                   lambda a: a["pmin"].__setitem__((np.nonzero(a["pmin"][:, 3] != -1)[0][0], 3), 77.0),
                   lambda a: a["pmin"].__setitem__((5, 0), np.nan)):
        broken = nodes.copy()
        mutate(broken)
        with pytest.raises(mipt.MiptError):
            mipt.validate_bvh(broken, 50)
    with pytest.raises(mipt.MiptError):
        mipt.validate_bvh(nodes[:-1], 50)
    # a child's box that sticks out of its parent's: the ordered traversal and the fixed-order
    # loop would no longer agree on which leaves a ray reaches
    broken = nodes.copy()
    broken["pmax"][1, 0] = nodes["pmax"][0, 0] + 1.0
    with pytest.raises(mipt.MiptError, match="not inside the box of its parent"):
        mipt.validate_bvh(broken, 50)


def test_int_bits_link_encoding(mipt, scenes):
    """The integer encoding documented at src/bvh/bvh_builder.h:29-33 carries the same tree."""
    vtx, idx, _, _ = scenes.random_soup(300, seed=2)
    lb = mipt.mesh_leaf_bounds(vtx, idx)
    f = mipt.build_bvh(lb, mipt.bvh_options(**{"bvh.builder": "sah"}))
    i = mipt.build_bvh(lb, mipt.bvh_options(**{"bvh.builder": "sah", "mipt.link_encoding": "int"}))
    mipt.validate_bvh(i, 300, link_encoding=mipt.PT_LINKS_INT_BITS)
    np.testing.assert_array_equal(f["pmin"][:, :3], i["pmin"][:, :3])
    np.testing.assert_array_equal(f["pmin"][:, 3].astype(np.int32), i["pmin"][:, 3].view(np.int32))
    np.testing.assert_array_equal(f["pmax"][:, 3].astype(np.int32), i["pmax"][:, 3].view(np.int32))
    assert (i["pmin"][:, 3].view(np.uint32) == 0xFFFFFFFF).sum() == 299


def test_obj_reader(mipt, scenes, tmp_path):
    vtx, idx, mats, mids = scenes.cornell_box()
    assert vtx.shape == (64, 3) and idx.shape == (32, 3)
    assert mats.shape[0] == 4 and mids.max() == 3
    light = mats[mids[10]]           # 6th quad = light
    np.testing.assert_allclose(light["ke"], [17, 12, 4])
    np.testing.assert_allclose(mats[mids[6]]["kd"], [0.14, 0.45, 0.091])   # right wall: green
    # polygons fan-triangulated, negative indices, v/vt/vn forms, no material -> default
    p = tmp_path / "t.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 2 0\n"
                 "f 1/1/1 2/2/2 3//3 4 5\nf -5 -4 -3\n")
    m = mipt.Mesh(str(p))
    np.testing.assert_array_equal(m.idx, [[0, 1, 2], [0, 2, 3], [0, 3, 4], [0, 1, 2]])
    assert m.materials.shape[0] == 1 and (m.material_ids == 0).all()
    assert m.n_shapes == 1 and (m.shape_ids == 0).all()
    np.testing.assert_array_equal(m.shape_prim_ids, [0, 1, 2, 3])
    # shapes as tinyobj cuts them: an o / g statement ends a shape that has faces; each triangle
    # keeps its shape and its index within it (the Face records of intersector_skip_links.h:82-92)
    q = tmp_path / "shapes.obj"
    q.write_text("o first\ng also_first\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n"
                 "g second\nf 1 2 3\nf 2 3 4\no third\no still_third\nf 1 3 4\ng empty_tail\n")
    ms = mipt.Mesh(str(q))
    assert ms.n_shapes == 3
    np.testing.assert_array_equal(ms.shape_ids, [0, 0, 1, 1, 2])
    np.testing.assert_array_equal(ms.shape_prim_ids, [0, 1, 0, 1, 0])
    # "scale model into unit cube" (src/main.cpp:865-880)
    m = mipt.Mesh(str(p), normalize=True)
    ext = m.vtx.max(axis=0) - m.vtx.min(axis=0)
    assert ext.max() == pytest.approx(1.0) and abs(m.vtx.max(axis=0) + m.vtx.min(axis=0)).max() < 1e-6
    with pytest.raises(mipt.MiptError, match="cannot open"):
        mipt.Mesh(str(tmp_path / "missing.obj"))
    (tmp_path / "bad.obj").write_text("v 0 0 0\nf 1 2 3\n")
    with pytest.raises(mipt.MiptError, match="out of range"):
        mipt.Mesh(str(tmp_path / "bad.obj"))


def test_atrium_stand_in_counts(scenes):
    vtx, idx, mats, mids = scenes.atrium(65536, seed=42)
    assert idx.shape == (65536, 3) and mids.shape == (65536,)
    assert idx.max() < vtx.shape[0] and np.isfinite(vtx).all()
    v2, i2, _, _ = scenes.atrium(65536, seed=42)
    np.testing.assert_array_equal(vtx, v2)
    np.testing.assert_array_equal(idx, i2)


def test_bvh_cache_round_trip(mipt, scenes, tmp_path):
    """Load-if-valid-else-build: a second call returns the same nodes from the file; other
    inputs or a damaged file fall back to a rebuild."""
    vtx, idx, _, _ = scenes.random_soup(500, seed=3)
    lb = mipt.mesh_leaf_bounds(vtx, idx)
    path = str(tmp_path / "scene.bvh")
    opt = mipt.bvh_options(**{"bvh.builder": "sah"})
    a, cached = mipt.build_bvh_cached(lb, path, opt)
    assert not cached
    b, cached = mipt.build_bvh_cached(lb, path, opt)
    assert cached
    np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
    np.testing.assert_array_equal(a.view(np.uint32), mipt.build_bvh(lb, opt).view(np.uint32))
    # other options: stale, rebuilt and replaced
    c, cached = mipt.build_bvh_cached(lb, path, mipt.bvh_options(**{"bvh.builder": "median"}))
    assert not cached
    # damaged payload: checksum fails, rebuilt
    raw = bytearray(open(path, "rb").read())
    raw[200] ^= 0xFF
    open(path, "wb").write(bytes(raw))
    d, cached = mipt.build_bvh_cached(lb, path, mipt.bvh_options(**{"bvh.builder": "median"}))
    assert not cached
    np.testing.assert_array_equal(c.view(np.uint32), d.view(np.uint32))
    # the thread count does not change the tree, so it does not change the key
    e, cached = mipt.build_bvh_cached(lb, path, mipt.bvh_options(**{"bvh.builder": "median",
                                                                    "mipt.num_threads": "3"}))
    assert cached
    # a header announcing more nodes than the file holds is refused before anything is allocated
    import os
    import struct
    raw = bytearray(open(path, "rb").read())
    raw[12:16] = struct.pack("<I", 0xFFFFFFFF)
    open(path, "wb").write(bytes(raw))
    h = mipt.C.c_void_p()
    key = mipt.lib.pt_bvh_cache_key(lb.ctypes.data, lb.shape[0], None)
    struct.pack_into("<Q", raw, 16, key)
    open(path, "wb").write(bytes(raw))
    assert mipt.lib.pt_bvh_load(path.encode(), key, mipt.C.byref(h)) == mipt.PT_ERR_IO
    assert "file size" in mipt.last_error()
    # saves go through a staging file of their own: none is left behind, and a save that cannot
    # be made (the rename meets a directory) leaves no staging file either
    f, cached = mipt.build_bvh_cached(lb, path, opt)
    assert not cached and os.listdir(tmp_path) == ["scene.bvh"]
    blocked = str(tmp_path / "blocked.bvh")
    os.mkdir(blocked)
    with pytest.raises(mipt.MiptError):
        mipt.build_bvh_cached(lb, blocked, opt)
    assert sorted(os.listdir(tmp_path)) == ["blocked.bvh", "scene.bvh"] and os.listdir(blocked) == []
    with pytest.raises(mipt.MiptError):
        mipt.build_bvh_cached(lb, str(tmp_path / "no_such_dir" / "x.bvh"), opt)


def test_concurrent_saves_of_one_cache_path(mipt, scenes, tmp_path):
    """Every rank of a multi-GPU run may miss the same cache file at once and save it: each save
    stages into a file of its own, all succeed, and the file that remains loads."""
    import os
    import threading
    vtx, idx, _, _ = scenes.random_soup(4000, seed=5)
    lb = mipt.mesh_leaf_bounds(vtx, idx)
    opt = mipt.bvh_options(**{"bvh.builder": "median"})
    path = str(tmp_path / "shared.bvh")
    want = mipt.build_bvh(lb, opt)
    results, errors = [], []
    start = threading.Barrier(6)

    def saver():
        try:
            start.wait()
            for _ in range(5):
                h = mipt.C.c_void_p()
                key = mipt.lib.pt_bvh_cache_key(lb.ctypes.data, lb.shape[0], mipt.C.byref(opt))
                assert mipt.lib.pt_bvh_build(lb.ctypes.data, lb.shape[0], mipt.C.byref(opt), mipt.C.byref(h)) == 0
                rc = mipt.lib.pt_bvh_save(h, path.encode(), key)
                mipt.lib.pt_bvh_destroy(h)
                results.append(rc)
        except Exception as e:                      # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=saver) for _ in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert results == [mipt.PT_OK] * 30
    assert os.listdir(tmp_path) == ["shared.bvh"]
    got, cached = mipt.build_bvh_cached(lb, path, opt)
    assert cached
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


def _sliver_scene(scenes, n=1500, seed=13):
    """Small triangles plus long thin ones whose boxes span much of the scene: what spatial
    splits are for."""
    vtx, idx, _, _ = scenes.random_soup(n, seed=seed, extent=10.0, size=0.3)
    rng = np.random.default_rng(seed)
    m = n // 10
    a = rng.uniform(-10, 10, size=(m, 3))
    d = rng.normal(size=(m, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    b = a + d * rng.uniform(8, 20, size=(m, 1))
    c = a + rng.normal(size=(m, 3)) * 0.1
    long_v = np.stack([a, b, c], axis=1).reshape(-1, 3).astype(np.float32)
    vtx = np.concatenate([vtx, long_v])
    idx = np.concatenate([idx, np.arange(3 * m, dtype=np.int32).reshape(-1, 3) + 3 * n])
    return vtx, idx


def test_spatial_splits(mipt, oracle, scenes):
    """bvh.sah.use_splits (src/bvh/bvh_builder.cpp:38-44): boxes are cut, so a triangle may sit
    in several leaves; the array stays a valid skip-link tree, the nearest hit stays that of the
    linear scan, and rays test fewer boxes on a scene with slivers."""
    vtx, idx = _sliver_scene(scenes)
    T = idx.shape[0]
    lb = mipt.mesh_leaf_bounds(vtx, idx)
    plain = mipt.build_bvh(lb, mipt.bvh_options(**{"bvh.builder": "sah"}))
    split = mipt.build_bvh(lb, mipt.bvh_options(**{"bvh.builder": "sah", "bvh.sah.use_splits": 1,
                                                   "bvh.sah.max_split_depth": 16,
                                                   "bvh.sah.min_overlap": 1e-4}))
    assert plain.shape[0] == 2 * T - 1
    leaves = (split.shape[0] + 1) // 2
    assert T < leaves <= T + int(0.5 * T)                      # extra_node_budget = 0.5
    mipt.validate_bvh(split, T)
    raw = split.view(np.float32).reshape(-1, 8)
    payload = raw[:, 3].astype(np.int64)
    assert set(payload[payload >= 0].tolist()) == set(range(T))
    # a leaf's box lies inside its triangle's inflated box
    full = oracle.inflate(lb).view(np.float32).reshape(-1, 8)
    lf = payload >= 0
    assert (raw[lf, 0:3] >= full[payload[lf], 0:3]).all() and (raw[lf, 4:7] <= full[payload[lf], 4:7]).all()
    rays = random_rays(20000, 5, dtype=oracle.ray_dtype)
    got, c_split = oracle.closest(split, vtx, idx, rays, nthreads=8, counters=True)
    want, c_plain = oracle.closest(plain, vtx, idx, rays, nthreads=8, counters=True)
    brute = oracle.brute_closest(vtx, idx, rays, nthreads=8)
    np.testing.assert_array_equal(got["uvwt"][:, 3], brute["uvwt"][:, 3])
    np.testing.assert_array_equal(got["primid"], want["primid"])
    # boxes are cut, not re-fitted to the geometry (the builder is given boxes only), so the gain
    # is modest: fewer slab tests, somewhat more triangle tests
    assert c_split.box_tests < c_plain.box_tests
    # budget 0: no splits, the plain tree
    none = mipt.build_bvh(lb, mipt.bvh_options(**{"bvh.builder": "sah", "bvh.sah.use_splits": 1,
                                                  "bvh.sah.extra_node_budget": 0}))
    assert none.shape[0] == 2 * T - 1
    mipt.validate_bvh(none, T)


def test_insertion_optimisation_keeps_leaves_and_answers(mipt, oracle, scenes):
    """Option mipt.optimize_passes (an extra of this project): the finished tree is rearranged
    to a smaller total box area; it stays a valid skip-link tree over the same leaf boxes and
    every ray finds what it found before (the nearer-child-first traversal of the GPU then
    meets about 2 % fewer boxes on the benchmark scene; the fixed-order walk does not profit)."""
    from conftest import random_rays, same_hits
    vtx, idx, _, _ = scenes.atrium(65536, seed=5)
    lb = mipt.mesh_leaf_bounds(vtx, idx)
    plain = mipt.build_bvh(lb, mipt.bvh_options(**{"bvh.builder": "sah"}))
    better = mipt.build_bvh(lb, mipt.bvh_options(**{"bvh.builder": "sah", "mipt.optimize_passes": 2}))
    assert better.shape == plain.shape
    mipt.validate_bvh(better, idx.shape[0])

    def leaves(nodes):
        raw = nodes.view(np.float32).reshape(-1, 8)
        leaf = raw[:, 3] != -1.0
        order = np.argsort(raw[leaf, 3])
        return raw[leaf][order][:, [0, 1, 2, 4, 5, 6]]

    def cost(nodes):
        raw = nodes.view(np.float32).reshape(-1, 8).astype(np.float64)
        e = raw[:, 4:7] - raw[:, 0:3]
        a = e[:, 0] * e[:, 1] + e[:, 1] * e[:, 2] + e[:, 2] * e[:, 0]
        return a[raw[:, 3] == -1.0].sum() / a[0]

    np.testing.assert_array_equal(leaves(better), leaves(plain))
    assert cost(better) < 0.99 * cost(plain)
    rays = scenes.surface_rays(vtx, idx, 30000, seed=2).astype(oracle.ray_dtype)
    a = oracle.closest(plain, vtx, idx, rays, nthreads=8)
    b = oracle.closest(better, vtx, idx, rays, nthreads=8)
    bad = same_hits(a, b)
    assert len(bad) < 30                                   # order-dependent ties only
    np.testing.assert_allclose(a["uvwt"][bad, 3], b["uvwt"][bad, 3], rtol=1e-6)
    assert (a["primid"] >= 0).mean() > 0.5
    with pytest.raises(mipt.MiptError, match="optimize_passes"):
        mipt.bvh_options(**{"mipt.optimize_passes": 1000})
    # the same pass on a tree handed in (pt_bvh_optimize): what the option does inside the build
    again = mipt.optimize_bvh(plain, 2)
    np.testing.assert_array_equal(again.view(np.uint32), better.view(np.uint32))
