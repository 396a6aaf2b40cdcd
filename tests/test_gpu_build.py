"""The GPU BVH builder (pt_scene_build) against the same bar as the host builder: the node
array it produces must be a valid skip-link tree over exactly the reference's inflated leaf
boxes, and tracing it must give the oracle's hits on that same array, bit for bit.

Topology parity is unpinned for every builder here (RadeonRays::Bvh::Build is external to the
reference); what is pinned is the format (src/bvh/bvh_builder.h:26-34), the leaf boxes
(src/main_granite.cpp:162-172 + src/bvh/bvh_builder.cpp:53-59) and the traversal results."""
import numpy as np
import pytest

from conftest import random_rays, same_hits

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[(0, 0), (8, 0), (2, 0), (0, 3), (8, 2)],
                ids=["radix_tree", "clustering", "clustering_r2", "radix_tree_optimised",
                     "clustering_optimised"], autouse=True)
def algorithm(request, gpu_ctx):
    """The GPU builders: the Morton radix tree (the default) and locally-ordered clustering,
    each also followed by rounds of parallel insertion-based optimisation with treelet
    restructuring between them (after every round on the radix tree, after the second on the
    clustered one)."""
    gpu_ctx.set_option("build.ploc_radius", request.param[0])
    gpu_ctx.set_option("build.optimize_passes", request.param[1])
    gpu_ctx.set_option("build.treelet_every", 1 if request.param[0] == 0 else 2)
    gpu_ctx.set_option("build.split_ratio", 0)           # one leaf per triangle: the tests with splits say so
    yield request.param
    gpu_ctx.set_option("build.ploc_radius", 0)
    gpu_ctx.set_option("build.optimize_passes", 6)
    gpu_ctx.set_option("build.treelet_every", 3)         # the library's defaults
    gpu_ctx.set_option("build.split_ratio", 4)


def _check_tree(mipt, oracle, nodes, vtx, idx, enc):
    T = idx.shape[0]
    assert nodes.shape[0] == 2 * T - 1
    mipt.validate_bvh(nodes, T, enc)                      # structure: pre-order, links, one leaf per triangle
    raw = nodes.view(np.float32).reshape(-1, 8)
    field = (lambda c: raw[:, c].view(np.int32)) if enc == mipt.PT_LINKS_INT_BITS else (lambda c: raw[:, c].astype(np.int64))
    payload, link = field(3), field(7)
    leaf = payload >= 0
    assert leaf.sum() == T and sorted(payload[leaf].tolist()) == list(range(T))
    want = oracle.inflate(oracle.mesh_leaf_bounds(vtx, idx)).view(np.float32).reshape(-1, 8)
    got = raw[leaf][np.argsort(payload[leaf])]
    np.testing.assert_array_equal(got[:, [0, 1, 2, 4, 5, 6]].view(np.uint32),
                                  want[:, [0, 1, 2, 4, 5, 6]].view(np.uint32))
    # every inner box is exactly the union of its children's (left = i+1, right = link of i+1)
    inner = np.nonzero(~leaf)[0]
    l, r = inner + 1, link[inner + 1]
    np.testing.assert_array_equal(raw[inner, 0:3], np.minimum(raw[l, 0:3], raw[r, 0:3]))
    np.testing.assert_array_equal(raw[inner, 4:7], np.maximum(raw[l, 4:7], raw[r, 4:7]))


@pytest.mark.parametrize("enc", ["float", "int"])
def test_device_build_soup(mipt, oracle, scenes, gpu_ctx, enc):
    enc = mipt.PT_LINKS_INT_BITS if enc == "int" else mipt.PT_LINKS_FLOAT_VALUE
    vtx, idx, _, _ = scenes.random_soup(30000, seed=21)
    sc = mipt.Scene(gpu_ctx, None, vtx, idx, link_encoding=enc)
    try:
        nodes = sc.read_nodes()
        _check_tree(mipt, oracle, nodes, vtx, idx, enc)
        rays = random_rays(80000, 4, dtype=mipt.ray_dtype)
        got = sc.trace_closest(rays)
        want = oracle.closest(nodes, vtx, idx, rays, nthreads=8, int_bits=enc == mipt.PT_LINKS_INT_BITS)
        assert len(same_hits(got, want)) == 0
        assert (got["primid"] >= 0).mean() > 0.2
        brute = oracle.brute_closest(vtx, idx, rays[:4000], nthreads=8)
        np.testing.assert_array_equal(got["uvwt"][:4000, 3], brute["uvwt"][:, 3])
        if enc == mipt.PT_LINKS_FLOAT_VALUE:
            rays["o"][:, 3] = 8.0
            np.testing.assert_array_equal(sc.trace_any(rays), oracle.any_hit(nodes, vtx, idx, rays, nthreads=8))
    finally:
        sc.close()


def test_device_build_is_deterministic_and_handles_duplicates(mipt, oracle, scenes, gpu_ctx):
    """Coincident triangles share one Morton code: ties are broken by position, every run alike."""
    vtx, idx, _, _ = scenes.random_soup(2000, seed=2)
    idx = np.concatenate([idx, idx[:500], idx[:500], idx[500:1200]])   # exact duplicates: three and two alike
    a = mipt.Scene(gpu_ctx, None, vtx, idx)
    b = mipt.Scene(gpu_ctx, None, vtx, idx)
    try:
        na, nb = a.read_nodes(), b.read_nodes()
        np.testing.assert_array_equal(na.view(np.uint32), nb.view(np.uint32))
        _check_tree(mipt, oracle, na, vtx, idx, mipt.PT_LINKS_FLOAT_VALUE)
        rays = random_rays(30000, 8, dtype=mipt.ray_dtype)
        assert len(same_hits(a.trace_closest(rays), oracle.closest(na, vtx, idx, rays, nthreads=8))) == 0
    finally:
        a.close()
        b.close()


def test_device_build_small_and_degenerate(mipt, oracle, scenes, gpu_ctx):
    # two triangles (one inner node), one triangle (host path), all centres equal, flat scene
    vtx, idx, _, _ = scenes.fallback_quad()
    for tri in (idx, idx[:1]):
        sc = mipt.Scene(gpu_ctx, None, vtx, tri)
        nodes = sc.read_nodes()
        _check_tree(mipt, oracle, nodes, vtx, tri, mipt.PT_LINKS_FLOAT_VALUE)
        rays = random_rays(5000, 1, extent=2.0, dtype=mipt.ray_dtype)
        assert len(same_hits(sc.trace_closest(rays), oracle.closest(nodes, vtx, tri, rays))) == 0
        sc.close()
    same = np.tile(np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0]]), (50, 1))
    idx50 = np.arange(150, dtype=np.int32).reshape(50, 3)
    sc = mipt.Scene(gpu_ctx, None, same, idx50)
    _check_tree(mipt, oracle, sc.read_nodes(), same, idx50, mipt.PT_LINKS_FLOAT_VALUE)
    sc.close()
    bad = vtx.copy()
    bad[0, 0] = np.nan
    with pytest.raises(mipt.MiptError, match="not finite"):
        mipt.Scene(gpu_ctx, None, bad, idx)
    with pytest.raises(mipt.MiptError, match="outside the vertex buffer"):
        mipt.Scene(gpu_ctx, None, vtx, np.int32([[0, 1, 99]]))


def test_device_build_renders_like_the_oracle(mipt, oracle, scenes, gpu_ctx):
    """Path tracing over a GPU-built tree: the film equals the CPU restatement run on the
    tree read back from the device."""
    vtx, idx, mats, mids = scenes.cornell_box()
    sc = mipt.Scene(gpu_ctx, None, vtx, idx, mats, mids)
    film = mipt.Film(gpu_ctx, 64, 64)
    try:
        cam = mipt.look_at_camera((278.0, 273.0, -800.0), (278.0, 273.0, 0.0), fovy_deg=39.3)
        p = mipt.render_params(64, 64, sample_count=4, max_depth=4, flags=mipt.PT_RENDER_NEE)
        mipt.render(gpu_ctx, sc, cam, p, film)
        want, _ = oracle.render(sc.read_nodes(), vtx, idx, mats, mids, cam, p, nthreads=8)
        np.testing.assert_array_equal(film.read().view(np.uint32), want.view(np.uint32))
    finally:
        film.close()
        sc.close()


def test_optimisation_passes_improve_the_tree(mipt, scenes, gpu_ctx, algorithm):
    """build.optimize_passes: fewer box tests per ray on the same rays with the same hits, and
    every pass count gives a valid tree (moves of one pass must not collide)."""
    if algorithm != (0, 0):
        pytest.skip("sets the options itself")
    vtx, idx, _, _ = scenes.clutter(200_000, seed=5)
    rays = scenes.surface_rays(vtx, idx, 1 << 16, seed=3).astype(mipt.ray_dtype)
    boxes, hits = [], []
    try:
        for passes in (0, 1, 6):
            gpu_ctx.set_option("build.optimize_passes", passes)
            sc = mipt.Scene(gpu_ctx, None, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS)
            try:
                mipt.validate_bvh(sc.read_nodes(), idx.shape[0], mipt.PT_LINKS_INT_BITS)
                got, st = sc.trace_closest(rays, stats=True)
                boxes.append(st.box_tests / rays.shape[0])
                hits.append(got)
            finally:
                sc.close()
    finally:
        gpu_ctx.set_option("build.optimize_passes", 6)
    assert boxes[1] < boxes[0] and boxes[2] < 0.92 * boxes[0], boxes
    for h in hits[1:]:
        # the nearest distance does not depend on the tree; which of two triangles at one distance
        # is reported may
        np.testing.assert_array_equal(h["uvwt"][:, 3], hits[0]["uvwt"][:, 3])


def test_treelet_rounds_improve_the_tree(mipt, oracle, scenes, gpu_ctx, algorithm):
    """build.treelet_every: the rounds of treelet restructuring between the insertion rounds give
    a valid tree over the same leaf boxes with a smaller surface-area sum and fewer box tests per
    ray, and the hits stay the oracle's on the tree read back."""
    if algorithm != (0, 0):
        pytest.skip("sets the options itself")
    vtx, idx, _, _ = scenes.clutter(120_000, seed=9)
    rays = scenes.surface_rays(vtx, idx, 1 << 16, seed=4).astype(mipt.ray_dtype)

    def area_sum(nodes):
        raw = nodes.view(np.float32).reshape(-1, 8)
        e = raw[:, 4:7].astype(np.float64) - raw[:, 0:3].astype(np.float64)
        a = e[:, 0] * e[:, 1] + e[:, 0] * e[:, 2] + e[:, 1] * e[:, 2]
        return float(a[raw[:, 3].view(np.int32) < 0].sum() / a[0])

    cost, boxes = {}, {}
    try:
        gpu_ctx.set_option("build.optimize_passes", 4)
        for every in (0, 1, 4):
            gpu_ctx.set_option("build.treelet_every", every)
            sc = mipt.Scene(gpu_ctx, None, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS)
            try:
                nodes = sc.read_nodes()
                _check_tree(mipt, oracle, nodes, vtx, idx, mipt.PT_LINKS_INT_BITS)
                got, st = sc.trace_closest(rays, stats=True)
                assert len(same_hits(got, oracle.closest(nodes, vtx, idx, rays, nthreads=8, int_bits=True), True)) == 0
                cost[every], boxes[every] = area_sum(nodes), st.box_tests / rays.shape[0]
            finally:
                sc.close()
    finally:
        gpu_ctx.set_option("build.optimize_passes", 6)
        gpu_ctx.set_option("build.treelet_every", 3)
    assert cost[1] < cost[4] < cost[0], cost
    assert boxes[1] < boxes[0] and boxes[4] < boxes[0], boxes


def _levels(nodes):
    raw = nodes.view(np.float32).reshape(-1, 8)
    payload, link = raw[:, 3].astype(np.int64), raw[:, 7].astype(np.int64)
    d = np.zeros(len(raw), dtype=np.int64)
    for i in range(len(raw)):
        if payload[i] < 0:
            d[i + 1] = d[i] + 1
            d[link[i + 1]] = d[i] + 1
    return int(d.max()) + 1


def test_too_deep_a_tree_falls_back_to_the_radix_tree(mipt, oracle, gpu_ctx, algorithm):
    """Optimised trees have no depth bound of their own; the level lists the passes work with do
    (256).  Past the limit the build answers with the radix tree instead of failing.  The limit
    is lowered here (option build.max_levels) so that ordinary input reaches it."""
    if algorithm != (0, 0):
        pytest.skip("sets the options itself")
    n = 400
    x = 1.08 ** np.arange(n, dtype=np.float64)
    tri = np.array([[0, -1, -1], [0, 1, -1], [0, 0, 1]], dtype=np.float64) * 0.02
    vtx = (tri[None] * x[:, None, None] + np.stack([x, 0 * x, 0 * x], 1)[:, None]).reshape(-1, 3).astype(np.float32)
    idx = np.arange(3 * n, dtype=np.int32).reshape(-1, 3)
    rays = np.zeros(2 * n, dtype=mipt.ray_dtype)
    rays["o"][:n] = [(-1, 0, 0, 1e30)] * n
    rays["d"][:n, 0] = 1
    rays["o"][n:, 0] = (x * 1.01).astype(np.float32)
    rays["o"][n:, 3] = 1e30
    rays["d"][n:, 0] = -1
    levels = {}
    try:
        for limit in (256, 36):
            gpu_ctx.set_option("build.max_levels", limit)
            for radius, passes in ((0, 0), (8, 0), (8, 2), (0, 4)):
                gpu_ctx.set_option("build.ploc_radius", radius)
                gpu_ctx.set_option("build.optimize_passes", passes)
                sc = mipt.Scene(gpu_ctx, None, vtx, idx)
                try:
                    nodes = sc.read_nodes()
                    _check_tree(mipt, oracle, nodes, vtx, idx, mipt.PT_LINKS_FLOAT_VALUE)
                    levels[limit, radius, passes] = _levels(nodes)
                    got = sc.trace_closest(rays)
                    assert len(same_hits(got, oracle.closest(nodes, vtx, idx, rays), True)) == 0
                    assert (got["primid"] >= 0).mean() > 0.9
                finally:
                    sc.close()
    finally:
        gpu_ctx.set_option("build.ploc_radius", 0)
        gpu_ctx.set_option("build.optimize_passes", 6)
        gpu_ctx.set_option("build.max_levels", 256)
    radix = levels[256, 0, 0]
    assert radix <= 36 and levels[36, 0, 0] == radix
    for radius, passes in ((8, 2), (0, 4)):
        assert levels[256, radius, passes] > 36, levels            # deeper than the lowered limit ...
        assert levels[36, radius, passes] == radix, levels         # ... so the radix tree it is
    assert levels[36, 8, 0] == levels[256, 8, 0]                   # plain clustering needs no level lists


# ---- triangles in several leaves (the reference's bvh.sah.use_splits, src/bvh/bvh_builder.cpp:38-44) ----
def _cone_fan(n=1024, seed=1):
    """A fan of n sliver triangles around the tip of a cone standing askew (so the slivers run
    diagonally through their boxes), closed by a base fan, plus a floor of ordinary triangles."""
    rng = np.random.default_rng(seed)
    phi = np.linspace(0, 2 * np.pi, n, endpoint=False)
    ring = np.stack([np.cos(phi), np.sin(phi), np.zeros(n)], 1)
    tip, centre = np.array([[0.0, 0.0, 1.5]]), np.array([[0.0, 0.0, 0.0]])
    vtx = np.concatenate([ring, tip, centre])
    nxt = (np.arange(n) + 1) % n
    side = np.stack([np.arange(n), nxt, np.full(n, n)], 1)
    base = np.stack([nxt, np.arange(n), np.full(n, n + 1)], 1)
    a = rng.normal(size=3)
    a /= np.linalg.norm(a)
    k = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    rot = np.eye(3) + np.sin(0.7) * k + (1 - np.cos(0.7)) * k @ k
    vtx = vtx @ rot.T + np.array([0.3, -0.2, 0.1])
    g = 60
    xs, ys = np.meshgrid(np.linspace(-3, 3, g), np.linspace(-3, 3, g), indexing="ij")
    floor = np.stack([xs.ravel(), ys.ravel(), np.full(g * g, -2.0)], 1)
    q = (np.arange(g - 1)[:, None] * g + np.arange(g - 1)[None, :]).ravel() + len(vtx)
    fidx = np.concatenate([np.stack([q, q + g, q + 1], 1), np.stack([q + 1, q + g, q + g + 1], 1)])
    return (np.concatenate([vtx, floor]).astype(np.float32),
            np.concatenate([side, base, fidx]).astype(np.int32))


def _fan_rays(mipt, vtx, idx, n, seed):
    """Rays towards points of the sliver triangles from all around."""
    rng = np.random.default_rng(seed)
    t = rng.integers(0, idx.shape[0], n)
    w = rng.dirichlet([1, 1, 1], n).astype(np.float64)
    p = (vtx[idx[t]].astype(np.float64) * w[:, :, None]).sum(1)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros(n, dtype=mipt.ray_dtype)
    rays["o"][:, :3] = (p - 4.0 * d).astype(np.float32)
    rays["o"][:, 3] = 1e30
    rays["d"][:, :3] = d.astype(np.float32)
    return rays


def _check_split_tree(mipt, oracle, nodes, vtx, idx, enc=None, budget=0.5):
    """A valid skip-link tree in which every triangle has one leaf or several; a leaf's box lies
    inside its triangle's (inflated) box, and the boxes of a triangle's leaves cover it."""
    enc = mipt.PT_LINKS_FLOAT_VALUE if enc is None else enc
    T = idx.shape[0]
    mipt.validate_bvh(nodes, T, enc)
    raw = nodes.view(np.float32).reshape(-1, 8)
    payload = raw[:, 3].view(np.int32) if enc == mipt.PT_LINKS_INT_BITS else raw[:, 3].astype(np.int64)
    leaf = payload >= 0
    L = int(leaf.sum())
    assert nodes.shape[0] == 2 * L - 1 and T <= L <= (1 + 1.02 * budget) * T + 64
    per_tri = np.bincount(payload[leaf], minlength=T)
    assert per_tri.min() >= 1
    whole = oracle.inflate(oracle.mesh_leaf_bounds(vtx, idx)).view(np.float32).reshape(-1, 8)
    lb, lt = raw[leaf], payload[leaf]
    slack = 4e-7 * (np.abs(whole[lt][:, [0, 1, 2, 4, 5, 6]]).max(axis=1, keepdims=True) + 1e-30)
    assert (lb[:, 0:3] >= whole[lt, 0:3] - slack).all() and (lb[:, 4:7] <= whole[lt, 4:7] + slack).all()
    one = per_tri[lt] == 1                                  # an uncut triangle keeps the reference's box, bit for bit
    np.testing.assert_array_equal(lb[one][:, [0, 1, 2, 4, 5, 6]].view(np.uint32),
                                  whole[lt[one]][:, [0, 1, 2, 4, 5, 6]].view(np.uint32))
    # points of the cut triangles (corners, edges, inside) each lie in one of their triangle's leaves
    rng = np.random.default_rng(5)
    cut = np.nonzero(per_tri > 1)[0]
    order = np.argsort(lt, kind="stable")
    first = np.concatenate([[0], np.cumsum(per_tri)])
    w = np.concatenate([np.eye(3), rng.dirichlet([1, 1, 1], 29), rng.dirichlet([0.3, 0.3, 0.3], 32)])
    for t in cut[rng.permutation(len(cut))[:400]]:
        boxes = lb[order[first[t]:first[t + 1]]].astype(np.float64)
        pts = w @ vtx[idx[t]].astype(np.float64)
        inside = ((pts[:, None, :] >= boxes[None, :, 0:3]) & (pts[:, None, :] <= boxes[None, :, 4:7])).all(2).any(1)
        assert inside.all(), (t, per_tri[t])
    return per_tri


@pytest.mark.parametrize("scene", ["cone_fan", "cone_fan_doubled", "pole_fans"])
def test_split_leaves_of_fans(mipt, oracle, scenes, gpu_ctx, algorithm, scene):
    """Long thin triangles lying askew in their boxes go into several leaves; hits stay those of the
    oracle walking the tree read back, and of a search over all triangles; a ray through the fan
    meets far fewer boxes."""
    if algorithm not in ((0, 0), (0, 3), (8, 2)):
        pytest.skip("one builder of each kind is enough here")
    if scene.startswith("cone_fan"):
        vtx, idx = _cone_fan()
        rays = _fan_rays(mipt, vtx, idx[:2048], 60000, seed=3)
        if scene == "cone_fan_doubled":
            # every third sliver twice, a few three times: rays that hit them have two or three equal
            # candidates, each in several leaves, and are decided at those leaves or by the fixed-order walk
            idx = np.concatenate([idx, idx[0:2048:3], idx[0:2048:30]])
    else:
        vtx, idx, _, _ = scenes.displaced_sphere(120_000, seed=7)
        pole = np.nonzero(np.abs(vtx[idx][:, :, 1]).max(axis=1) > 0.999 * np.abs(vtx[:, 1]).max())[0]
        rays = np.concatenate([_fan_rays(mipt, vtx, idx[pole], 30000, seed=4),
                               scenes.surface_rays(vtx, idx, 30000, seed=9).astype(mipt.ray_dtype)])
    work = {}
    try:
        gpu_ctx.set_option("build.split_budget", 400)        # most of these small scenes is fan
        for ratio in (0, 8):
            gpu_ctx.set_option("build.split_ratio", ratio)
            a = mipt.Scene(gpu_ctx, None, vtx, idx)
            b = mipt.Scene(gpu_ctx, None, vtx, idx)
            try:
                nodes = a.read_nodes()
                np.testing.assert_array_equal(nodes.view(np.uint32), b.read_nodes().view(np.uint32))
                if ratio:
                    per_tri = _check_split_tree(mipt, oracle, nodes, vtx, idx, budget=4.0)
                    assert per_tri.max() > 4 and (per_tri > 1).sum() > 500      # the fans were cut
                else:
                    _check_tree(mipt, oracle, nodes, vtx, idx, mipt.PT_LINKS_FLOAT_VALUE)
                got, st = a.trace_closest(rays, stats=True)
                work[ratio] = st.box_tests / rays.shape[0]
                want = oracle.closest(nodes, vtx, idx, rays, nthreads=8)
                assert len(same_hits(got, want)) == 0
                if scene == "cone_fan_doubled":
                    assert st.tie_rays > 2000
                brute = oracle.brute_closest(vtx, idx, rays[:3000], nthreads=8)
                np.testing.assert_array_equal(got["uvwt"][:3000, 3], brute["uvwt"][:, 3])
                assert (got["primid"] >= 0).mean() > 0.5
                rays_any = rays.copy()
                rays_any["o"][:, 3] = 3.9
                np.testing.assert_array_equal(a.trace_any(rays_any), oracle.any_hit(nodes, vtx, idx, rays_any, nthreads=8))
            finally:
                a.close()
                b.close()
    finally:
        gpu_ctx.set_option("build.split_ratio", 0)
        gpu_ctx.set_option("build.split_budget", 50)
    # (a 346-sliver fan at the pole of the small sphere is cut into at most 6 leaves per sliver
    # and is a small part of what these rays meet; the 1024-sliver cone is most of its scene)
    assert work[8] < (0.6 if scene.startswith("cone_fan") else 1.05) * work[0], work


def test_split_budget_and_limits(mipt, oracle, gpu_ctx, algorithm):
    """build.split_budget caps the extra leaves (every triangle's surplus scaled alike), and
    build.split_max the leaves of one triangle."""
    if algorithm != (0, 0):
        pytest.skip("sets the options itself")
    vtx, idx = _cone_fan()
    T = idx.shape[0]
    leaves = {}
    try:
        gpu_ctx.set_option("build.split_ratio", 4)           # the library's default
        for budget, most in ((50, 32), (10, 32), (1000, 4), (0, 32)):
            gpu_ctx.set_option("build.split_budget", budget)
            gpu_ctx.set_option("build.split_max", most)
            sc = mipt.Scene(gpu_ctx, None, vtx, idx)
            try:
                nodes = sc.read_nodes()
                raw = nodes.view(np.float32).reshape(-1, 8)
                per_tri = np.bincount(raw[:, 3].astype(np.int64)[raw[:, 3] >= 0], minlength=T)
                leaves[budget, most] = int(per_tri.sum())
                assert per_tri.max() <= most and per_tri.sum() - T <= 1.02 * budget * T / 100 + 64
                mipt.validate_bvh(nodes, T, mipt.PT_LINKS_FLOAT_VALUE)
            finally:
                sc.close()
    finally:
        gpu_ctx.set_option("build.split_ratio", 0)
        gpu_ctx.set_option("build.split_budget", 50)
        gpu_ctx.set_option("build.split_max", 64)
    assert leaves[0, 32] == T and leaves[10, 32] > T and leaves[50, 32] > leaves[10, 32], leaves
