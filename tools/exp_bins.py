"""Experiment: how much the SAH bin count changes the work per ray on the bench scene."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "gpu-pathtracer_amd"))
import numpy as np
import torch
import mipt
from mipt import scenes

ctx = mipt.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
vtx, idx, mats, mids = scenes.atrium(262144, seed=42)
n = 1 << 22
rays = scenes.surface_rays(vtx, idx, n, seed=7)
d_r = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
d_h = torch.empty_like(d_r)
lb = mipt.mesh_leaf_bounds(vtx, idx)
for opts in ({"bvh.builder": "median"}, {"bvh.builder": "sah", "bvh.sah.num_bins": "16"},
             {"bvh.builder": "sah"}, {"bvh.builder": "sah", "bvh.sah.num_bins": "256"},
             {"bvh.builder": "sah", "bvh.sah.num_bins": "2048"}):
    t0 = time.time()
    nodes = mipt.build_bvh(lb, mipt.bvh_options(**opts))
    bs = time.time() - t0
    sc = mipt.Scene(ctx, nodes, vtx, idx)
    stc = sc.trace_closest_device_counted(d_r.data_ptr(), n, d_h.data_ptr())
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr())
    ctx.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(json.dumps({"opts": opts, "build_s": bs, "box_per_ray": stc.box_tests / n,
                      "tri_per_ray": stc.tri_tests / n, "mrays_s": n / dt / 1e6}), flush=True)
    sc.close()
