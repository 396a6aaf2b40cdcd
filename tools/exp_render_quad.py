import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-pathtracer_amd"))
import numpy as np, torch, mipt
from mipt import scenes
ctx = mipt.Context(0); ctx.set_kernel_timing(True)
vtx, idx, mats, mids = scenes.clutter(10_000_000, seed=11)
cam = mipt.look_at_camera((-17.0, 10.5, -16.0), (3.0, 1.0, 2.0), fovy_deg=50.0, aspect=16 / 9)
nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx), mipt.bvh_options(**{"bvh.builder": "sah", "mipt.link_encoding": "int"}))
sc = mipt.Scene(ctx, nodes, vtx, idx, mats, mids, link_encoding=mipt.PT_LINKS_INT_BITS)
W, H = 3840, 2160
film = mipt.Film(ctx, W, H)
for depth in (2, 8):
    p = mipt.render_params(W, H, sample_count=4, max_depth=depth, seed=5, rr_start_depth=3, background=(0.5, 0.6, 0.8))
    for q in (0, 1, -1, 0, 1, -1):
        ctx.set_option("extend.quad", q)
        mipt.render(ctx, sc, cam, p, film, stats=True)
        st = mipt.render(ctx, sc, cam, p, film, stats=True)
        print(json.dumps({"depth": depth, "quad": q, "extend_ms": round(st.extend_ms, 2), "shade_ms": round(st.shade_ms, 2),
                          "rays": st.rays_primary + st.rays_bounce, "launches": st.extend_launches}), flush=True)
