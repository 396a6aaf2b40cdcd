"""The committed regression vectors (tests/golden/vectors.npz, written by
tests/golden/make_vectors.py): the CPU restatement still produces them, and the GPU path
produces them too.  They pin the project's results across changes; the reference itself holds
no outputs to pin against (parity unpinned)."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _load():
    return np.load(os.path.join(HERE, "golden", "vectors.npz"), allow_pickle=False)


def _same_float_bits(a, b):
    fa, fb = a.view(np.float32), b.view(np.float32)
    return ((a == b) | (np.isnan(fa) & np.isnan(fb))).all()


def test_cpu_restatement_reproduces_the_vectors(oracle):
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_vectors import vectors
    want = _load()
    got = vectors()
    assert sorted(got) == sorted(want.files)
    for k in want.files:
        if k.endswith("_bits"):
            assert _same_float_bits(got[k], want[k]), k
        else:
            np.testing.assert_array_equal(got[k], want[k], err_msg=k)


@pytest.mark.gpu
def test_gpu_reproduces_the_vectors(mipt, scenes, gpu_ctx):
    """No CPU restatement in the loop: the library alone against the committed data."""
    from test_oracle import cornell_rays
    import two_level_util as tl
    want = _load()
    vtx, idx, mats, mids = scenes.cornell_box()
    nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx))
    sc = mipt.Scene(gpu_ctx, nodes, vtx, idx, mats, mids)
    film = mipt.Film(gpu_ctx, 48, 48)
    try:
        hits = sc.trace_closest(cornell_rays(256, 256).astype(mipt.ray_dtype))
        np.testing.assert_array_equal(hits["primid"].astype(np.int16), want["cornell256_primid"])
        assert _same_float_bits(hits["uvwt"][:, 3].view(np.uint32), want["cornell256_t_bits"])
        cam = mipt.look_at_camera((278.0, 273.0, -800.0), (278.0, 273.0, 0.0), fovy_deg=39.3, aspect=1.0)
        for name, flags in (("plain", 0), ("nee", mipt.PT_RENDER_NEE)):
            film.clear()
            mipt.render(gpu_ctx, sc, cam, mipt.render_params(48, 48, sample_count=8, max_depth=4, seed=1,
                                                             flags=flags), film)
            assert _same_float_bits(film.read().view(np.uint32), want[f"cornell_film_{name}_bits"]), name
    finally:
        film.close()
        sc.close()
    s2 = tl.scene()
    meshes = [(mipt.build_bvh(mipt.mesh_leaf_bounds(v, i)), v, i, m) for v, i, m in s2["meshes"]]
    top = mipt.build_bvh(mipt.instance_bounds(meshes, s2["transforms"], s2["mesh_of"]))
    two = mipt.Scene.instanced(gpu_ctx, meshes, s2["transforms"], s2["mesh_of"], top,
                               materials=tl.materials(mipt), material_offsets=s2["material_offsets"])
    try:
        h = two.trace_closest(tl.rays_for(8192, 9, mipt.ray_dtype)[4:])
        np.testing.assert_array_equal(h["shapeid"].astype(np.int16), want["two_level_shapeid"])
        np.testing.assert_array_equal(h["primid"].astype(np.int16), want["two_level_primid"])
        assert _same_float_bits(h["uvwt"][:, 3].view(np.uint32), want["two_level_t_bits"])
    finally:
        two.close()
