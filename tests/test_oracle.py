"""The CPU oracle against the literals the reference text holds, and against itself
(skip-link traversal == linear scan).  No GPU, no product code."""

import numpy as np
import pytest

from conftest import random_rays, same_hits

# shaders/depth.frag:133-136: v1=(-1,-1,0), v3=(1,-1,0), v2=(0,1,0), passed as (v1, v2, v3)
UNIT_TRI = np.array([[-1, -1, 0], [0, 1, 0], [1, -1, 0]], dtype=np.float32)


def _ray(o, d, tmax=1e30):
    from oracle.oracle import ray_dtype
    r = np.zeros(1, dtype=ray_dtype)
    r["o"][0] = (*o, tmax)
    r["d"][0] = (*d, 0.0)
    return r


def test_unit_triangle_known_answers(oracle):
    idx = np.array([[0, 1, 2]], dtype=np.int32)
    # down the z axis through the origin: t = 5, weights (v1,v2,v3) = (1/4, 1/2, 1/4), all exact in fp32
    r = _ray((0, 0, 5), (0, 0, -1))
    h = oracle.brute_closest(UNIT_TRI, idx, r)[0]
    assert (h["shapeid"], h["primid"]) == (0, 0)
    assert tuple(h["padding"]) == (667, 0)                      # bvh.glslh:429-430
    np.testing.assert_array_equal(h["uvwt"], np.float32([0.25, 0.5, 0.25, 5.0]))
    # a ray outside the triangle, and one pointing away: miss convention (bvh.glslh:423-425)
    for r in (_ray((2, 2, 5), (0, 0, -1), tmax=1000.0), _ray((0, 0, 5), (0, 0, 1), tmax=1000.0)):
        h = oracle.brute_closest(UNIT_TRI, idx, r)[0]
        assert (h["shapeid"], h["primid"]) == (-1, -1)
        assert tuple(h["padding"]) == (667, 0)
        np.testing.assert_array_equal(h["uvwt"], np.float32([0, 0, 0, 1000.0]))
    # tmax shorter than the hit distance rejects it (range test in the scaled domain, :268-273)
    h = oracle.brute_closest(UNIT_TRI, idx, _ray((0, 0, 5), (0, 0, -1), tmax=4.999))[0]
    assert h["primid"] == -1
    # no back-face culling: the same triangle is hit from behind
    h = oracle.brute_closest(UNIT_TRI, idx, _ray((0, 0, -5), (0, 0, 1)))[0]
    assert h["primid"] == 0 and h["uvwt"][3] == np.float32(5.0)


def test_unit_triangle_moller_trumbore(oracle):
    """IntersectTriangle (bvh.glslh:181-202) writes (b1, b2, 0, t): weights of v2 and v3."""
    from oracle.oracle import isect_dtype
    lib = oracle.lib()
    r = _ray((0, 0, 5), (0, 0, -1))
    isect = np.zeros(1, dtype=isect_dtype)
    isect["uvwt"][0, 3] = 1000.0
    v = UNIT_TRI
    ok = lib.o_intersect_triangle_mt(r.ctypes.data, v[0].ctypes.data, v[1].ctypes.data,
                                     v[2].ctypes.data, isect.ctypes.data)
    assert ok == 1
    np.testing.assert_array_equal(isect["uvwt"][0], np.float32([0.5, 0.25, 0.0, 5.0]))


def test_precompute_ray_axis_rules(oracle):
    """Dominant-axis permutation with the >= tie rules of bvh.glslh:55-59."""
    lib = oracle.lib()
    out = np.zeros(14, dtype=np.float32)   # o_ray (8) + s (3) + d (3)
    cases = {
        (1.0, 1.0, 1.0): (1.0, 1.0, 1.0),       # x wins ties -> d.yzx
        (0.5, 1.0, 1.0): (1.0, 0.5, 1.0),       # y wins over z -> d.zxy
        (0.5, 0.25, -2.0): (0.5, 0.25, -2.0),   # z dominant -> unchanged
        (-3.0, 1.0, 2.0): (1.0, 2.0, -3.0),
    }
    for d, perm in cases.items():
        r = _ray((0, 0, 0), d)
        lib.o_precompute_ray(r.ctypes.data, out.ctypes.data)
        np.testing.assert_array_equal(out[11:14], np.float32(perm))
        px, py, pz = np.float32(perm)
        np.testing.assert_array_equal(out[8:11], np.float32([-px / pz, -py / pz, np.float32(1) / pz]))


def test_fallback_quad(oracle, scenes):
    """The 2-triangle mesh of src/main.cpp:918-921: interior points of each triangle are hit,
    through the BVH and by linear scan alike."""
    vtx, idx, _, _ = scenes.fallback_quad()
    nodes = oracle.build_bvh(oracle.mesh_leaf_bounds(vtx, idx))
    assert nodes.shape[0] == 3
    for tri in range(2):
        p = vtx[idx[tri]].astype(np.float64).mean(axis=0)
        o = p + np.array([3.0, 0.2, 0.1])
        d = (p - o) / np.linalg.norm(p - o)
        r = _ray(o, d)
        hb = oracle.brute_closest(vtx, idx, r)[0]
        ht = oracle.closest(nodes, vtx, idx, r)[0]
        assert hb["primid"] == tri and ht["primid"] == tri
        assert ht["shapeid"] == 0 and tuple(ht["padding"]) == (667, 0)
        assert abs(float(ht["uvwt"][3]) - np.linalg.norm(p - o)) < 1e-5
        assert len(same_hits(hb[None], ht[None])) == 0


def test_shared_edge_is_watertight_and_later_triangle_wins(oracle):
    """A ray through the shared diagonal of a quad: both edge functions are exactly zero, the
    double-precision fallback (bvh.glslh:242-252) runs, both triangles are hit at the same
    distance and the one tested last is kept (equal t passes the range test, :268-273)."""
    vtx = np.array([[0, 0, 0], [2, 0, 0], [2, 2, 0], [0, 2, 0]], dtype=np.float32)
    idx = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int32)
    r = _ray((1, 1, 4), (0, 0, -1))
    h = oracle.brute_closest(vtx, idx, r)[0]
    assert h["primid"] == 1 and h["uvwt"][3] == np.float32(4.0)
    h = oracle.brute_closest(vtx, idx[::-1].copy(), r)[0]
    assert h["primid"] == 1   # order reversed: now index 1 is the first quad half, still the last tested


def test_raygen_reference_known_answers(oracle):
    """depth.frag:47-83 re-derived independently in numpy float32 for a few pixels."""
    f = np.float32
    cam = oracle.camera_from(eye=(0.5, -1.0, -10.0), c0=(1, 0, 0), c1=(0, 1, 0), c2=(0, 0, 1),
                             sx=0.8, sy=-0.6)
    W, H = 8, 6
    rays = oracle.raygen_reference(cam, W, H)
    for (px, py) in [(0, 0), (7, 5), (3, 2), (4, 3)]:
        u = (f(px) + f(0.5)) / f(W)
        v = (f(py) + f(0.5)) / f(H)
        dx = (u - f(0.5)) * f(0.8)
        dy = (v - f(0.5)) * f(-0.6)
        d = np.array([dx, dy, f(-1.0)], dtype=np.float32)
        ln = np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2], dtype=np.float32)
        d = d / ln
        scale = f(1.0) / f(20.0)
        o = np.array([f(0.5), f(-1.0) + f(2.0), f(-10.0), f(1000.0)], dtype=np.float32) / scale
        r = rays[py * W + px]
        np.testing.assert_array_equal(r["o"], o)
        np.testing.assert_array_equal(r["d"][:3], d)
        assert r["d"][3] == 0.0
    # the debug divide is a vec4 divide: tmax becomes 1000/0.05 (depth.frag:83)
    assert rays["o"][0, 3] == np.float32(1000.0) / (np.float32(1.0) / np.float32(20.0))


def test_depth_shade(oracle):
    from oracle.oracle import isect_dtype
    h = np.zeros(3, dtype=isect_dtype)
    h["shapeid"] = [-1, 0, 0]
    h["uvwt"][:, 3] = [1000.0, 100.0, 20.0]
    out = oracle.depth_shade(h)
    np.testing.assert_array_equal(out[0], [0, 0, 0, 1])
    # t=100 -> x=5 -> smoothstep(5, 0, 3): ((3-5)/(0-5))=0.4 -> 0.16*(3-0.8)
    assert abs(out[1, 0] - 0.352) < 1e-6 and out[1, 0] == out[1, 1] == out[1, 2] and out[1, 3] == 1
    # t=20 -> x=1 -> (3-1)/(0-1) = -2 -> clamped to 0
    np.testing.assert_array_equal(out[2], [0, 0, 0, 1])


@pytest.mark.parametrize("use_sah", [False, True])
@pytest.mark.parametrize("n_tri,seed", [(1, 0), (2, 1), (3, 2), (64, 3), (2000, 4)])
def test_bvh_equals_linear_scan(oracle, scenes, n_tri, seed, use_sah):
    from oracle.oracle import ray_dtype
    vtx, idx, _, _ = scenes.random_soup(n_tri, seed=seed)
    nodes = oracle.build_bvh(oracle.mesh_leaf_bounds(vtx, idx), use_sah=use_sah)
    assert nodes.shape[0] == 2 * n_tri - 1
    rays = random_rays(4000, seed + 100, dtype=ray_dtype)
    a = oracle.closest(nodes, vtx, idx, rays, nthreads=4)
    b = oracle.brute_closest(vtx, idx, rays, nthreads=4)
    assert len(same_hits(a, b)) == 0
    assert (a["primid"] >= 0).sum() > 0 or n_tri < 3
    occl = oracle.any_hit(nodes, vtx, idx, rays, nthreads=4)
    # any-hit uses Moller-Trumbore with an epsilon slack, so it may only add hits near edges
    assert ((a["primid"] >= 0) & (occl == 0)).sum() <= 2


def test_hits_agree_with_double_precision_geometry(oracle, scenes):
    """Independent of the reference's text: the nearest intersection computed by a plain
    double-precision Moller-Trumbore over all triangles.  Wherever that answer is clear-cut
    (not within 1e-5 of an edge, no other triangle within 1e-5 of the same distance) the
    restated watertight traversal must name the same triangle, distance and barycentrics."""
    from oracle.oracle import ray_dtype
    vtx, idx, _, _ = scenes.random_soup(1500, seed=9)
    nodes = oracle.build_bvh(oracle.mesh_leaf_bounds(vtx, idx))
    rays = random_rays(6000, 77, dtype=ray_dtype)
    got = oracle.closest(nodes, vtx, idx, rays, nthreads=8)
    v = vtx.astype(np.float64)[idx]                       # [T,3,3]
    e1, e2 = v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
    checked = 0
    for k in range(rays.shape[0]):
        o, d = rays["o"][k, :3].astype(np.float64), rays["d"][k, :3].astype(np.float64)
        pv = np.cross(d, e2)
        det = np.einsum("ij,ij->i", e1, pv)
        ok = np.abs(det) > 1e-14
        inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
        tv = o - v[:, 0]
        u = np.einsum("ij,ij->i", tv, pv) * inv
        qv = np.cross(tv, e1)
        w = np.einsum("j,ij->i", d, qv) * inv
        t = np.einsum("ij,ij->i", e2, qv) * inv
        inside = ok & (u >= 0) & (w >= 0) & (u + w <= 1) & (t > 0)
        if not inside.any():
            margin = ok & (u > -1e-5) & (w > -1e-5) & (u + w < 1 + 1e-5) & (t > -1e-5)
            if not margin.any():
                assert got["primid"][k] == -1
                checked += 1
            continue
        ts = np.where(inside, t, np.inf)
        best = int(np.argmin(ts))
        others = np.delete(ts, best)
        clear = (min(u[best], w[best], 1 - u[best] - w[best]) > 1e-5 and
                 others.min() > ts[best] * (1 + 1e-5) + 1e-5)
        near_edge_closer = (ok & (t > 0) & (t < ts[best]) & (u > -1e-5) & (w > -1e-5) & (u + w < 1 + 1e-5)).any()
        if not clear or near_edge_closer:
            continue
        checked += 1
        assert got["primid"][k] == best
        b0, b1, b2, tt = got["uvwt"][k]
        # the watertight test reports (b0, b1, b2) for (v0, v1, v2); u, w weigh v1, v2
        np.testing.assert_allclose([b1, b2, tt], [u[best], w[best], ts[best]], rtol=2e-4, atol=2e-5)
        assert abs(b0 + b1 + b2 - 1) < 1e-5
    assert checked > 4000


def test_cornell_primary_hits_vs_linear_scan(oracle, scenes):
    """Config #1 of the baseline: Cornell box, 256x256 pixel-centre primary rays, on the CPU.
    BVH order and index order may disagree only where two triangles are hit at distances
    within 2 ulp (shared edges); those are counted, not hidden."""
    vtx, idx, _, _ = scenes.cornell_box()
    assert idx.shape[0] == 32
    nodes = oracle.build_bvh(oracle.mesh_leaf_bounds(vtx, idx))
    rays = cornell_rays(256, 256)
    a = oracle.closest(nodes, vtx, idx, rays, nthreads=8)
    b = oracle.brute_closest(vtx, idx, rays, nthreads=8)
    bad = same_hits(a, b)
    ta, tb = a["uvwt"][bad, 3].view(np.int32), b["uvwt"][bad, 3].view(np.int32)
    assert (np.abs(ta.astype(np.int64) - tb) <= 2).all(), "non-tie mismatch"
    assert (a["primid"][bad] >= 0).all() and (b["primid"][bad] >= 0).all()
    assert len(bad) < 600
    assert (a["primid"] >= 0).mean() > 0.9   # the frame is slightly wider than the box opening
    assert len(np.unique(a["primid"])) >= 20


def cornell_rays(W, H):
    """Pixel-centre pinhole rays of the original Cornell camera, computed in float64 then
    rounded (inputs only: any deterministic ray set would do)."""
    from oracle.oracle import ray_dtype
    eye = np.array([278.0, 273.0, -800.0])
    t = np.tan(np.radians(39.3) / 2)
    px, py = np.meshgrid(np.arange(W) + 0.5, np.arange(H) + 0.5)
    x = (2 * px / W - 1) * t * (W / H)
    y = (1 - 2 * py / H) * t
    d = np.stack([-x, y, np.ones_like(x)], -1).reshape(-1, 3)   # looking down +z, x to the left
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros(W * H, dtype=ray_dtype)
    rays["o"][:, :3] = eye.astype(np.float32)
    rays["o"][:, 3] = 1e30
    rays["d"][:, :3] = d.astype(np.float32)
    return rays


def test_linearizer_format(oracle, scenes):
    """Skip-link format of src/bvh/bvh_builder.h:26-34 and the structural asserts of
    plain_bvh_translator.cpp:47,55,85."""
    vtx, idx, _, _ = scenes.random_soup(500, seed=9)
    nodes = oracle.build_bvh(oracle.mesh_leaf_bounds(vtx, idx))
    n = nodes.shape[0]
    assert n == 2 * 500 - 1
    payload, link = nodes["pmin"][:, 3], nodes["pmax"][:, 3]
    assert link[0] == -1.0
    leaf = payload != -1.0
    assert leaf.sum() == 500
    assert sorted(payload[leaf].astype(int)) == list(range(500))     # each triangle in one leaf
    for i in range(n):
        assert link[i] == -1.0 or (link[i] > i and link[i] < n)
        if not leaf[i]:
            r = int(link[i + 1])            # the left child sits at i+1 and links to the right child
            assert r > i + 1 and link[r] == link[i]
    # leaf boxes are the triangle boxes inflated by 1+1e-4 about the centre (bvh_builder.cpp:53-59)
    lb = oracle.inflate(oracle.mesh_leaf_bounds(vtx, idx))
    for i in np.nonzero(leaf)[0][:50]:
        t = int(payload[i])
        np.testing.assert_array_equal(nodes["pmin"][i, :3], lb["pmin"][t, :3])
        np.testing.assert_array_equal(nodes["pmax"][i, :3], lb["pmax"][t, :3])
