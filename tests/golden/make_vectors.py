"""Writes the committed regression vectors under tests/golden/ from the CPU restatement.

The reference holds no golden outputs for this path (its only literals -- the fallback quad,
the unit triangle, the 667 marker -- are checked directly in tests/test_oracle.py), so these
vectors do not pin the restatement to the reference: they pin it, the shading specification
and the GPU kernels to themselves across changes.  Regenerate only after a deliberate change of
specification:   python tests/golden/make_vectors.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(REPO, "gpu-pathtracer_amd"), REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def vectors():
    """name -> array, computed by the CPU restatement only (no GPU)."""
    import mipt
    from mipt import scenes
    from oracle import oracle
    from test_oracle import cornell_rays
    import two_level_util as tl

    out = {}
    # baseline config 1: Cornell box, 256x256 pixel-centre primary rays -> triangle id, distance
    vtx, idx, mats, mids = scenes.cornell_box()
    nodes = oracle.build_bvh(oracle.mesh_leaf_bounds(vtx, idx))
    hits = oracle.closest(nodes, vtx, idx, cornell_rays(256, 256), nthreads=8)
    out["cornell256_primid"] = hits["primid"].astype(np.int16)
    out["cornell256_t_bits"] = hits["uvwt"][:, 3].view(np.uint32).copy()
    # the path tracer on it: 48x48, 8 spp, depth 4, seed 1, without and with light sampling
    cam = mipt.look_at_camera((278.0, 273.0, -800.0), (278.0, 273.0, 0.0), fovy_deg=39.3, aspect=1.0)
    for name, flags in (("plain", 0), ("nee", mipt.PT_RENDER_NEE)):
        p = mipt.render_params(48, 48, sample_count=8, max_depth=4, seed=1, flags=flags)
        film, _ = oracle.render(nodes, vtx, idx, mats, mids, cam, p, nthreads=8)
        out[f"cornell_film_{name}_bits"] = film.view(np.uint32).copy()
    # a two-level scene: instance, triangle and distance of 8188 rays
    sc = tl.scene()
    meshes = [(oracle.build_bvh(oracle.mesh_leaf_bounds(v, i)), v, i) for v, i, _ in sc["meshes"]]
    t = oracle.TwoLevel(meshes, sc["transforms"], sc["mesh_of"])
    t.top = oracle.build_bvh(t.bounds())
    # (without the generator's four degenerate rays: what a NaN ray returns depends on the tree)
    h = t.closest(tl.rays_for(8192, 9, oracle.ray_dtype)[4:], nthreads=8)
    out["two_level_shapeid"] = h["shapeid"].astype(np.int16)
    out["two_level_primid"] = h["primid"].astype(np.int16)
    out["two_level_t_bits"] = h["uvwt"][:, 3].view(np.uint32).copy()
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "vectors.npz"), **vectors())
    print("wrote", os.path.join(HERE, "vectors.npz"), os.path.getsize(os.path.join(HERE, "vectors.npz")), "bytes")
