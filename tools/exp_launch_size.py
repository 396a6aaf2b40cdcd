"""Closest-hit kernel time against the number of rays of the launch (the fixed part of a launch:
the persistent grid ends on its slowest waves), 28 M-triangle displaced sphere, GPU-built tree."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "gpu-pathtracer_amd"))
import numpy as np
import torch
import mipt
from mipt import scenes

ctx = mipt.Context(0)
ctx.set_kernel_timing(True)
for a in sys.argv[1:]:
    if "=" in a:
        k_, _, v_ = a.partition("=")
        ctx.set_option(k_, int(v_))
gen = scenes.displaced_cube_sphere if os.environ.get("SCENE", "uv") == "cube" else scenes.displaced_sphere
vtx, idx, _, _ = gen(int(os.environ.get("TRIS", "28000000")), seed=7)
ctx.set_option("build.optimize_passes", int(os.environ.get("PASSES", "8")))
if "TREELET" in os.environ:
    ctx.set_option("build.treelet_every", int(os.environ["TREELET"]))
import time
ctx.synchronize()
_t0 = time.perf_counter()
sc = mipt.Scene(ctx, None, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS)
ctx.synchronize()
build_s = time.perf_counter() - _t0
nmax = 1 << 25
FLAGS = int(os.environ.get("FLAGS", "0"))       # 4: PT_TRACE_NO_TIE_FIXUP
rays = scenes.surface_rays(vtx, idx, nmax, seed=7)
d_r = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
d_h = torch.empty_like(d_r)
ctx.set_option("extend.quad", 0)
_st = sc.trace_closest_device_counted(d_r.data_ptr(), 1 << 22, d_h.data_ptr())
ctx.set_option("extend.quad", -1)
print(json.dumps({"scene": os.environ.get("SCENE", "uv"), "tris": int(idx.shape[0]), "depth": sc.tree_depth(),
                  "build_s": round(build_s, 3), "boxes_per_ray": round(_st.box_tests / (1 << 22), 2),
                  "tris_per_ray": round(_st.tri_tests / (1 << 22), 2),
                  "leaves": (int(mipt.lib.pt_scene_node_count(sc._h)) + 1) // 2}), flush=True)
for lg in [int(x) for x in os.environ.get('LGS', '20 21 22 23 24 25').split()]:
    n = 1 << lg
    row = {"rays": n}
    for label, sort in (("sorted_ms", 1), ("queue_order_ms", 0)):
        ctx.set_option("trace.sort_rays", sort)
        sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr(), FLAGS); ctx.synchronize()
        k = 0.0
        for _ in range(3):
            sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr(), FLAGS)
            k += ctx.last_trace_ms()
        row[label] = round(k / 3, 3)
        row[label.replace("_ms", "_ns_per_ray")] = round(k / 3 * 1e6 / n, 3)
    print(json.dumps(row), flush=True)
