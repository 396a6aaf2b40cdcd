"""One closest-hit call over 2^24 secondary rays on the 28 M-triangle stand-in (tree built on
the GPU), for profiling the traversal beyond the caches."""
import os, sys, json, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "gpu-pathtracer_amd"))
import numpy as np
import torch
import mipt
from mipt import scenes

ctx = mipt.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
for o in sys.argv[1:]:
    k, _, v = o.partition("=")
    ctx.set_option(k, int(v))
vtx, idx, mats, mids = scenes.displaced_sphere(28_000_000, seed=7)
sc = mipt.Scene(ctx, None, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS)
n = 1 << 24
rays = scenes.surface_rays(vtx, idx, n, seed=7)
d_r = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
d_h = torch.empty_like(d_r)
sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr()); ctx.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr())
ctx.synchronize()
print(json.dumps({"mrays_s": 3 * n / (time.perf_counter() - t0) / 1e6}))
