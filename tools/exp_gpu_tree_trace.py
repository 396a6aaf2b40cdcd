"""Closest-hit rate and work per ray on GPU-built trees (radix tree and clustering) beside the
host SAH tree, on the 9 M-triangle scene: where does a lower tree cost not turn into speed?"""
import os, sys, json, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "gpu-pathtracer_amd"))
import numpy as np
import torch
import mipt
from mipt import scenes

n_tri = int(sys.argv[1]) if len(sys.argv) > 1 else 9_000_000
ctx = mipt.Context(0)
vtx, idx, _, _ = scenes.clutter(n_tri, seed=11)
n = 1 << 22
rays = scenes.surface_rays(vtx, idx, n, seed=7).astype(mipt.ray_dtype)
d_r = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
d_h = torch.empty_like(d_r)
ctx.set_option("trace.sort_rays", 0)


def measure(sc, label):
    st = sc.trace_closest_device_counted(d_r.data_ptr(), n, d_h.data_ptr())
    sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr()); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr())
    ctx.synchronize()
    print(json.dumps({"tree": label, "mrays_s": round(3 * n / (time.perf_counter() - t0) / 1e6, 1),
                      "boxes_per_ray": round(st.box_tests / n, 2), "tris_per_ray": round(st.tri_tests / n, 3),
                      "retraced_rays": int(st.tie_rays)}), flush=True)


passes_list = [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["0"])]
for r in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "8"])]:
    for passes in passes_list:
        ctx.set_option("build.ploc_radius", r)
        ctx.set_option("build.optimize_passes", passes)
        ctx.synchronize()
        t0 = time.perf_counter()
        sc = mipt.Scene(ctx, None, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS)
        dt = time.perf_counter() - t0
        measure(sc, ("gpu radix tree" if r == 0 else f"gpu clustering r={r}") + f", {passes} optimisation passes, built in {dt * 1e3:.0f} ms")
        sc.close()
nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx), mipt.bvh_options(**{"bvh.builder": "sah", "mipt.link_encoding": "int"}))
sc = mipt.Scene(ctx, nodes, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS)
measure(sc, "host SAH")
