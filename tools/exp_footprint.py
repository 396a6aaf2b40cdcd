"""Closest-hit kernel time against the size of the scene's records, in steps across the 4 GB mark
where the rate of random fetches falls (tools/micro/gather_knee.hip): displaced spheres of
16..36 M triangles, GPU-built trees, 2^24 secondary rays.  MIPT_ALLOC_POW2=1: scene arrays in
allocations rounded up to powers of two."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "gpu-pathtracer_amd"))
import numpy as np
import torch
import mipt
from mipt import scenes

ctx = mipt.Context(0)
ctx.set_kernel_timing(True)
sizes = [int(a) for a in sys.argv[1:] if "=" not in a] or [16, 20, 22, 24, 26, 28, 32]
for a in sys.argv[1:]:
    if "=" in a:
        k_, _, v_ = a.partition("=")
        ctx.set_option(k_, int(v_))
n = 1 << 24
for m in sizes:
    vtx, idx, _, _ = scenes.displaced_sphere(m * 1_000_000, seed=7)
    ctx.set_option("build.optimize_passes", 8)
    sc = mipt.Scene(ctx, None, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS)
    rays = scenes.surface_rays(vtx, idx, n, seed=7)
    d_r = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
    d_h = torch.empty_like(d_r)
    row = {"tris_M": idx.shape[0] / 1e6, "records_GB": round(128 * idx.shape[0] / 2**30, 2), "depth": sc.tree_depth()}
    stc = sc.trace_closest_device_counted(d_r.data_ptr(), n, d_h.data_ptr())
    row["boxes_per_ray"] = round(stc.box_tests / n, 1)
    for label, sort in (("sorted_ms", 1), ("queue_order_ms", 0)):
        ctx.set_option("trace.sort_rays", sort)
        sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr()); ctx.synchronize()
        k = 0.0
        for _ in range(3):
            sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr())
            k += ctx.last_trace_ms()
        row[label] = round(k / 3, 3)
    print(json.dumps(row), flush=True)
    sc.close()
    del d_r, d_h
    torch.cuda.empty_cache()
