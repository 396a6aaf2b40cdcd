"""Closest-hit rate on the 28 M-triangle scene for different GPU builder settings (and the host SAH
tree with HOST=1): build time, slab tests per ray, kernel time in origin-cell order."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "gpu-pathtracer_amd"))
import numpy as np
import torch
import mipt
from mipt import scenes

ctx = mipt.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
vtx, idx, _, _ = scenes.displaced_sphere(28_000_000, seed=7)
n = 1 << 24
rays = scenes.surface_rays(vtx, idx, n, seed=7)
d_r = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
d_h = torch.empty_like(d_r)
ctx.set_kernel_timing(True)
configs = [("radix+8", 0, 8), ("radix+16", 0, 16)]
if os.environ.get("ONLY"):
    configs = [c for c in configs if c[0] == os.environ["ONLY"]]
for o in sys.argv[1:]:
    k_, _, v_ = o.partition("=")
    ctx.set_option(k_, int(v_))
if os.environ.get("HOST"):
    configs = [("host sah", -1, 0)]
for name, radius, passes in configs:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if radius < 0:
        nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx), mipt.bvh_options(**{"bvh.builder": "sah", "mipt.link_encoding": "int"}))
        sc = mipt.Scene(ctx, nodes, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS)
    else:
        ctx.set_option("build.ploc_radius", radius)
        ctx.set_option("build.optimize_passes", passes)
        sc = mipt.Scene(ctx, None, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS)
    ctx.synchronize()
    build_s = time.perf_counter() - t0
    stc = sc.trace_closest_device_counted(d_r.data_ptr(), n, d_h.data_ptr())
    # SPLITS: "stack:nodes ..." of the persistent workgroups' LDS; or SETS: "key=int,key=int ..."
    # option sets, each applied in turn (options stay set: give every key in every set)
    variants = os.environ.get("SETS", "").split() or os.environ.get("SPLITS", "-1:-1").split()
    for split in variants:
        if "=" in split:
            for kv in split.split(","):
                k_, _, v_ = kv.partition("=")
                ctx.set_option(k_, int(v_))
        else:
            S, C = split.split(":")
            ctx.set_option("extend.stack_entries", int(S))
            ctx.set_option("extend.cache_nodes", int(C))
        sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr()); ctx.synchronize()
        k = 0.0
        for _ in range(3):
            sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr())
            k += ctx.last_trace_ms()
        print(json.dumps({"split": split, "kernel_ms": round(k / 3, 2), "mrays_s": round(n / (k / 3) / 1e3)}), flush=True)
    print(json.dumps({"tree": name, "build_s": round(build_s, 2), "depth": sc.tree_depth(),
                      "boxes_per_ray": round(stc.box_tests / n, 1), "tris_per_ray": round(stc.tri_tests / n, 2),
                      "kernel_ms": round(k / 3, 2), "mrays_s": round(n / (k / 3) / 1e3),
                      "re_resolved_rays": int(stc.tie_rays)}), flush=True)
    sc.close()
