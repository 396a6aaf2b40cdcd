"""Does the order of the two-child nodes in memory matter beyond the caches?  The same host-built
tree of a scene larger than the Infinity Cache, laid out in pre-order (the default), breadth
first, in treelets of 2 (a 128-byte line), 64 (a 4 KB page) and 32768 nodes (2 MB), and shuffled;
2^24 incoherent secondary rays in origin-cell order and in queue order."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "gpu-pathtracer_amd"))
import numpy as np
import torch
import mipt
from mipt import scenes

ctx = mipt.Context(0)
ctx.set_kernel_timing(True)
which = sys.argv[1:] or ["sphere28m"]
for name in which:
    if name == "clutter10m":
        vtx, idx, _, _ = scenes.clutter(10_000_000, seed=11)
    else:
        vtx, idx, _, _ = scenes.displaced_sphere(28_000_000, seed=7)
    nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx),
                           mipt.bvh_options(**{"bvh.builder": "sah", "mipt.link_encoding": "int"}))
    n = 1 << 24
    rays = scenes.surface_rays(vtx, idx, n, seed=7)
    d_r = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
    d_h = torch.empty_like(d_r)
    ref = None
    for order in ("", "bfs", "treelet:2", "treelet:64", "treelet:32768", "random"):
        if order:
            os.environ["MIPT_WIDE_ORDER"] = order
        else:
            os.environ.pop("MIPT_WIDE_ORDER", None)
        sc = mipt.Scene(ctx, nodes, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS)
        row = {"scene": name, "order": order or "pre-order"}
        for label, sort in (("sorted_ms", 1), ("queue_order_ms", 0)):
            ctx.set_option("trace.sort_rays", sort)
            sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr()); ctx.synchronize()
            k = 0.0
            for _ in range(3):
                sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr())
                k += ctx.last_trace_ms()
            row[label] = round(k / 3, 2)
        if ref is None:
            ref = d_h.clone()
        row["same_hits"] = bool(torch.equal(ref.view(torch.int32), d_h.view(torch.int32)))
        print(json.dumps(row), flush=True)
        sc.close()
    os.environ.pop("MIPT_WIDE_ORDER", None)
