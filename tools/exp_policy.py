"""Which split of the persistent workgroups' LDS (stack entries : cached nodes @ workgroups per
CU) suits which scene and workload: trace calls with incoherent secondary rays (queue order and
origin-cell order) and render passes (primary + one bounce), on scenes beyond the caches."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "gpu-pathtracer_amd"))
import numpy as np
import torch
import mipt
from mipt import scenes

ctx = mipt.Context(0)
ctx.set_kernel_timing(True)
splits = os.environ.get("SPLITS", "-1:-1:-1 16:224:5 24:96:5 30:128:4 30:128:5").split()
which = [a for a in sys.argv[1:] if "=" not in a] or ["clutter10m", "sphere28m"]
for a in sys.argv[1:]:                      # key=int: pt_context_set_option first
    if "=" in a:
        k_, _, v_ = a.partition("=")
        ctx.set_option(k_, int(v_))
for name in which:
    if name == "clutter10m":
        vtx, idx, mats, mids = scenes.clutter(10_000_000, seed=11)
        cam = mipt.look_at_camera((-17.0, 10.5, -16.0), (3.0, 1.0, 2.0), fovy_deg=50.0, aspect=16 / 9)
    else:
        vtx, idx, mats, mids = scenes.displaced_sphere(28_000_000, seed=7)
        cam = mipt.look_at_camera((0.0, 0.5, -3.2), (0.0, 0.0, 0.0), fovy_deg=40.0, aspect=16 / 9)
    host = os.environ.get("HOST", "1") == "1"
    if host:
        nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx),
                               mipt.bvh_options(**{"bvh.builder": "sah", "mipt.link_encoding": "int"}))
        sc = mipt.Scene(ctx, nodes, vtx, idx, mats, mids, link_encoding=mipt.PT_LINKS_INT_BITS)
    else:
        ctx.set_option("build.optimize_passes", 8)
        sc = mipt.Scene(ctx, None, vtx, idx, mats, mids, link_encoding=mipt.PT_LINKS_INT_BITS)
    n = 1 << 24
    rays = scenes.surface_rays(vtx, idx, n, seed=7)
    d_r = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
    d_h = torch.empty_like(d_r)
    W, H = 3840, 2160
    film = mipt.Film(ctx, W, H)
    p = mipt.render_params(W, H, sample_count=4, max_depth=2, seed=5, background=(0.5, 0.6, 0.8))
    for split in splits:
        S, C, B = (int(x) for x in split.split(":"))
        ctx.set_option("extend.stack_entries", S)
        ctx.set_option("extend.cache_nodes", C)
        ctx.set_option("extend.blocks_per_cu", B)
        row = {"scene": name, "host_tree": host, "depth": sc.tree_depth(), "split": split}
        for label, sort in (("trace_sorted_ms", 1), ("trace_queue_order_ms", 0)):
            ctx.set_option("trace.sort_rays", sort)
            sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr()); ctx.synchronize()
            k = 0.0
            for _ in range(3):
                sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr())
                k += ctx.last_trace_ms()
            row[label] = round(k / 3, 2)
        mipt.render(ctx, sc, cam, p, film, stats=True)
        st = mipt.render(ctx, sc, cam, p, film, stats=True)
        row["render_extend_ms"] = round(st.extend_ms, 2)
        row["render_rays"] = st.rays_primary + st.rays_bounce
        print(json.dumps(row), flush=True)
    film.close(); sc.close()
