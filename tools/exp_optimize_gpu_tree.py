import os, sys, json, time
sys.path.insert(0, "gpu-pathtracer_amd")
import numpy as np, torch, mipt
from mipt import scenes
n_tri = 1_000_000
ctx = mipt.Context(0)
vtx, idx, _, _ = scenes.clutter(n_tri, seed=11)
n = 1 << 21
rays = scenes.surface_rays(vtx, idx, n, seed=7).astype(mipt.ray_dtype)
d_r = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda(); d_h = torch.empty_like(d_r)
def measure(sc, label):
    st = sc.trace_closest_device_counted(d_r.data_ptr(), n, d_h.data_ptr())
    sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr()); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr())
    ctx.synchronize()
    print(json.dumps({"tree": label, "mrays_s": round(3*n/(time.perf_counter()-t0)/1e6,1), "boxes_per_ray": round(st.box_tests/n,2), "retraced": int(st.tie_rays)}), flush=True)
sc = mipt.Scene(ctx, None, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS); measure(sc, "gpu radix")
lb = sc.read_nodes(); sc.close()
for passes in (1, 3):
    t0 = time.time(); opt = mipt.optimize_bvh(lb, passes, mipt.PT_LINKS_INT_BITS); dt = time.time() - t0
    sc = mipt.Scene(ctx, opt, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS); measure(sc, f"gpu radix + {passes} host passes ({dt:.1f} s)"); sc.close()
nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx), mipt.bvh_options(**{"bvh.builder": "sah", "mipt.link_encoding": "int"}))
sc = mipt.Scene(ctx, nodes, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS); measure(sc, "host SAH"); sc.close()
opt = mipt.optimize_bvh(nodes, 2, mipt.PT_LINKS_INT_BITS)
sc = mipt.Scene(ctx, opt, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS); measure(sc, "host SAH + 2 passes"); sc.close()
