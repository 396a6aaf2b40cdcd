"""The CPU restatement of the shading specification, checked against analytic results.
The reference has no shading (parity unpinned), so these known answers are what pins it."""
import numpy as np
import pytest


def _box_scene(scenes, kd, ke, ks=(0, 0, 0), rough=1.0):
    """Closed unit cube seen from inside: 12 triangles, one material."""
    v = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], dtype=np.float32)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    idx = np.array([t for a, b, c, d in quads for t in ((a, b, c), (a, c, d))], dtype=np.int32)
    mats = scenes.make_materials([(kd, ks, ke, rough)])
    return v, idx, mats, np.zeros(12, dtype=np.int32)


def _render(oracle, mipt, scene, cam, **kw):
    vtx, idx, mats, mids = scene
    nodes = oracle.build_bvh(oracle.mesh_leaf_bounds(vtx, idx))
    p = mipt.render_params(kw.pop("width", 16), kw.pop("height", 16), **kw)
    film, rays = oracle.render(nodes, vtx, idx, mats, mids, cam, p, nthreads=8)
    return film, rays, p


def test_sincos_polynomials(oracle):
    u = np.linspace(0, 1, 4001, endpoint=False).astype(np.float32)
    sc = np.array([oracle.sincos_2pi(float(x)) for x in u])
    ref = np.stack([np.sin(2 * np.pi * u.astype(np.float64)), np.cos(2 * np.pi * u.astype(np.float64))], 1)
    assert np.abs(sc - ref).max() < 4e-7
    assert oracle.sincos_2pi(0.0) == (0.0, 1.0)
    s, c = oracle.sincos_2pi(0.25)
    assert s == 1.0 and abs(c) < 1e-7


def test_rng_is_uniform_and_keyed(oracle):
    r = np.array([oracle.rnd(1, p, s, d) for p in range(40) for s in range(10) for d in range(10)])
    assert r.min() >= 0.0 and r.max() < 1.0
    assert abs(r.mean() - 0.5) < 0.02 and abs(r.var() - 1 / 12) < 0.01
    assert len(np.unique(r)) > 0.99 * len(r)
    assert oracle.rnd(1, 5, 3, 2) == oracle.rnd(1, 5, 3, 2)
    assert oracle.rnd(1, 5, 3, 2) != oracle.rnd(2, 5, 3, 2)


@pytest.mark.parametrize("depth", [1, 2, 5])
def test_lambert_furnace(oracle, mipt, scenes, depth):
    """Closed box, albedo 1/2, every surface emitting 1: with cosine-weighted sampling the
    Lambert weight is exactly the albedo, so every path returns sum_{k<depth} 2^-k."""
    cam = mipt.look_at_camera((0.1, 0.2, -0.3), (0.5, 0.3, 1.0), fovy_deg=70.0)
    film, rays, p = _render(oracle, mipt, _box_scene(scenes, (0.5, 0.5, 0.5), (1, 1, 1)), cam,
                            sample_count=4, max_depth=depth, rr_start_depth=99)
    want = sum(0.5 ** k for k in range(depth))
    np.testing.assert_allclose(film[..., :3] / film[..., 3:], want, rtol=1e-6)
    assert (film[..., 3] == 4).all()
    assert rays == 16 * 16 * 4 * depth


def test_russian_roulette_is_unbiased(oracle, mipt, scenes):
    cam = mipt.look_at_camera((0.1, 0.2, -0.3), (0.5, 0.3, 1.0), fovy_deg=70.0)
    scene = _box_scene(scenes, (0.5, 0.5, 0.5), (1, 1, 1))
    film, rays, _ = _render(oracle, mipt, scene, cam, width=32, height=32, sample_count=64,
                            max_depth=12, rr_start_depth=1)
    want = sum(0.5 ** k for k in range(12))
    mean = (film[..., :3] / film[..., 3:]).mean()
    assert abs(mean - want) / want < 0.01
    assert rays < 32 * 32 * 64 * 6          # and it does terminate paths early


def test_ggx_white_furnace_loses_little_energy(oracle, mipt, scenes):
    """Perfect mirror-coloured GGX lobe (ks = 1, so F = 1) in a uniformly lit box: single
    scattering must return at most the incoming radiance, and most of it."""
    cam = mipt.look_at_camera((0.1, 0.2, -0.3), (0.5, 0.3, 1.0), fovy_deg=70.0)
    for rough, lo in ((0.2, 0.95), (0.5, 0.85), (1.0, 0.25)):
        scene = _box_scene(scenes, (0, 0, 0), (1, 1, 1), ks=(1, 1, 1), rough=rough)
        film, _, _ = _render(oracle, mipt, scene, cam, width=24, height=24, sample_count=64,
                             max_depth=2, rr_start_depth=99)
        second = (film[..., 0] / film[..., 3]).mean() - 1.0      # minus the directly seen emission
        assert lo < second <= 1.005, (rough, second)


def test_background_and_miss(oracle, mipt, scenes):
    vtx, idx, mats, mids = scenes.fallback_quad()
    cam = mipt.look_at_camera((5, 0, 0), (0, 0, 0), fovy_deg=30.0)
    film, _, _ = _render(oracle, mipt, (vtx, idx, mats, mids), cam, sample_count=2, max_depth=3,
                         background=(0.25, 0.5, 1.0), flags=mipt.PT_RENDER_PIXEL_CENTRE)
    img = film[..., :3] / film[..., 3:]
    np.testing.assert_allclose(img[0, 0], [0.25, 0.5, 1.0])      # corner pixel sees only sky
    centre = img[8, 8]
    assert (centre <= np.array([0.25, 0.5, 1.0]) + 1e-6).all() and centre[2] > 0.1


def test_cornell_energy_is_sane(oracle, mipt, scenes):
    cam = mipt.look_at_camera((278.0, 273.0, -800.0), (278.0, 273.0, 0.0), fovy_deg=39.3)
    film, _, _ = _render(oracle, mipt, scenes.cornell_box(), cam, width=48, height=48,
                         sample_count=16, max_depth=4)
    img = film[..., :3] / film[..., 3:]
    assert np.isfinite(img).all() and img.min() >= 0
    assert 0.02 < img.mean() < 2.0
    # the left wall is red, the right wall green (as seen from the original camera)
    left, right = img[16:32, 5:10].mean(axis=(0, 1)), img[16:32, 38:43].mean(axis=(0, 1))
    assert left[0] > 2 * left[1] and right[1] > 2 * right[0]


def test_next_event_estimation_is_unbiased_and_quieter(oracle, mipt, scenes):
    """Light sampling + MIS estimates the same image as plain path tracing, with less noise:
    Cornell box, two seeds each way."""
    cam = mipt.look_at_camera((278.0, 273.0, -800.0), (278.0, 273.0, 0.0), fovy_deg=39.3)
    out = {}
    for flags in (0, mipt.PT_RENDER_NEE):
        imgs = []
        for seed in (1, 2):
            film, rays, _ = _render(oracle, mipt, scenes.cornell_box(), cam, width=40, height=40,
                                    sample_count=48, max_depth=4, seed=seed, flags=flags)
            imgs.append(film[..., :3] / film[..., 3:])
        out[flags] = (np.mean(imgs), imgs, rays)
    plain, nee = out[0], out[mipt.PT_RENDER_NEE]
    assert abs(nee[0] - plain[0]) / plain[0] < 0.03
    # noise = difference between the two seeds, away from the directly seen lamp (whose
    # jittered edge is equally noisy both ways)
    dim = np.max(plain[1] + nee[1], axis=(0, 3)) < 1.0
    noise = [float(np.sqrt(np.mean((i[0] - i[1])[dim] ** 2))) for i in (plain[1], nee[1])]
    assert noise[1] < 0.5 * noise[0], noise
    assert nee[2] > plain[2]                 # shadow rays are counted


def test_next_event_estimation_furnace(oracle, mipt, scenes):
    """Every surface emits 1 with albedo 1/2: light sampling and MIS weights must still add
    up to sum_{k<depth} 2^-k in expectation."""
    cam = mipt.look_at_camera((0.1, 0.2, -0.3), (0.5, 0.3, 1.0), fovy_deg=70.0)
    film, _, _ = _render(oracle, mipt, _box_scene(scenes, (0.5, 0.5, 0.5), (1, 1, 1)), cam,
                         width=32, height=32, sample_count=32, max_depth=4, rr_start_depth=99,
                         flags=mipt.PT_RENDER_NEE)
    want = sum(0.5 ** k for k in range(4))
    mean = (film[..., :3] / film[..., 3:]).mean()
    assert abs(mean - want) / want < 0.01
