"""Closest-hit rate on the 28 M-triangle scene (tree built on the GPU) for different sort keys of
the rays: bits per axis of the origin cell, bits per octahedral coordinate of the direction."""
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
ctx.set_option("build.optimize_passes", 8)
sc = mipt.Scene(ctx, None, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS)
n = 1 << 24
rays = scenes.surface_rays(vtx, idx, n, seed=7)
d_r = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
d_h = torch.empty_like(d_r)
ctx.set_kernel_timing(True)
ctx.set_option("trace.sort_rays", 1)
ref = None
for ob, db in [(10, 0), (10, 1), (9, 2), (8, 3), (8, 4), (7, 4), (7, 5), (6, 5), (6, 7), (5, 8), (4, 8)]:
    ctx.set_option("trace.sort_origin_bits", ob)
    ctx.set_option("trace.sort_dir_bits", db)
    sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr()); ctx.synchronize()
    if ref is None:
        ref = d_h.clone()
    assert torch.equal(ref.view(torch.int32), d_h.view(torch.int32))
    k = 0.0
    t0 = time.perf_counter()
    for _ in range(3):
        sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr())
        k += ctx.last_trace_ms()
    ctx.synchronize()
    call = (time.perf_counter() - t0) / 3 * 1e3
    print(json.dumps({"origin_bits": ob, "dir_bits": db, "kernel_ms": round(k / 3, 2), "call_ms": round(call, 2),
                      "mrays_s_call": round(n / call / 1e3)}), flush=True)
