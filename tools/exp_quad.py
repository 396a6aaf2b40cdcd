"""Four-child nodes with quantised boxes against the two-child nodes: same hits, kernel time,
slab tests per ray.  Scenes: the bench scene (in the caches), 10 M and 28 M triangles (beyond)."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "gpu-pathtracer_amd"))
import numpy as np
import torch
import mipt
from mipt import scenes

ctx = mipt.Context(0)
ctx.set_kernel_timing(True)
which = [a for a in sys.argv[1:] if "=" not in a] or ["atrium", "clutter10m", "sphere28m"]
for a in sys.argv[1:]:
    if "=" in a:
        k_, _, v_ = a.partition("=")
        ctx.set_option(k_, int(v_))
splits = os.environ.get("SPLITS", "-1:-1").split()
for name in which:
    if name == "atrium":
        vtx, idx, _, _ = scenes.atrium(262144, seed=42)
    elif name == "clutter10m":
        vtx, idx, _, _ = scenes.clutter(10_000_000, seed=11)
    else:
        vtx, idx, _, _ = scenes.displaced_sphere(28_000_000, seed=7)
    if os.environ.get("HOST", "1") == "1":
        nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx),
                               mipt.bvh_options(**{"bvh.builder": "sah", "mipt.link_encoding": "int"}))
        sc = mipt.Scene(ctx, nodes, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS)
    else:
        ctx.set_option("build.optimize_passes", int(os.environ.get("PASSES", "8")))
        ctx.set_option("build.ploc_radius", int(os.environ.get("PLOC", "0")))
        sc = mipt.Scene(ctx, None, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS)
    n = 1 << 24 if name != "atrium" else 1 << 22
    rays = scenes.surface_rays(vtx, idx, n, seed=7)
    d_r = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
    d_h = torch.empty_like(d_r)
    ref = None
    for quad in [int(q) for q in os.environ.get('QUADS', '0 1').split()]:
        ctx.set_option("extend.quad", quad)
        for split in splits:
            S, C = split.split(":")[:2]
            ctx.set_option("extend.stack_entries", int(S))
            ctx.set_option("extend.cache_nodes", int(C))
            ctx.set_option("extend.blocks_per_cu", int(split.split(":")[2]) if split.count(":") > 1 else -1)
            row = {"scene": name, "depth": sc.tree_depth(), "quad": quad, "split": split}
            stc = sc.trace_closest_device_counted(d_r.data_ptr(), n, d_h.data_ptr())
            row["boxes_per_ray"] = round(stc.box_tests / n, 1)
            row["tris_per_ray"] = round(stc.tri_tests / n, 2)
            row["ties"] = int(stc.tie_rays)
            for label, sort in (("sorted_ms", 1), ("queue_order_ms", 0)):
                ctx.set_option("trace.sort_rays", sort)
                sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr()); ctx.synchronize()
                k = 0.0
                for _ in range(3):
                    sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr())
                    k += ctx.last_trace_ms()
                row[label] = round(k / 3, 3)
            if ref is None:
                ref = d_h.clone()
            row["same_hits"] = bool(torch.equal(ref.view(torch.int32), d_h.view(torch.int32)))
            print(json.dumps(row), flush=True)
    ctx.set_option("extend.quad", -1)
    sc.close()
