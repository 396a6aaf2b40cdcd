"""Experiment: closest-hit rate on one set of secondary rays in generation order, in random
order, and sorted by (direction octant, Morton code of the origin)."""
import json
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "gpu-pathtracer_amd"))
import time
import numpy as np
import torch
import mipt
from mipt import scenes


def morton3(q):
    def spread(v):
        v = v.astype(np.uint64) & 0x3FF
        v = (v | (v << 16)) & 0x30000FF
        v = (v | (v << 8)) & 0x300F00F
        v = (v | (v << 4)) & 0x30C30C3
        v = (v | (v << 2)) & 0x9249249
        return v
    return (spread(q[:, 0]) << 2) | (spread(q[:, 1]) << 1) | spread(q[:, 2])


def main():
    ctx = mipt.Context(0)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
    for name, (vtx, idx, mats, mids) in (("atrium", scenes.atrium(262144, seed=42)),
                                         ("clutter10M", scenes.clutter(10_000_000, seed=11))):
        enc = mipt.PT_LINKS_INT_BITS
        nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx), mipt.bvh_options(**{"bvh.builder": "sah", "mipt.link_encoding": "int"}))
        sc = mipt.Scene(ctx, nodes, vtx, idx, link_encoding=enc)
        n = 1 << 22
        rays = scenes.surface_rays(vtx, idx, n, seed=7)
        o = rays["o"][:, :3]; d = rays["d"][:, :3]
        lo, hi = o.min(0), o.max(0)
        q = np.clip(((o - lo) / (hi - lo) * 1023).astype(np.int64), 0, 1023)
        octant = ((d[:, 0] < 0).astype(np.uint64) << 2) | ((d[:, 1] < 0).astype(np.uint64) << 1) | (d[:, 2] < 0).astype(np.uint64)
        orders = {"generated": np.arange(n), "random": np.random.default_rng(1).permutation(n),
                  "morton_origin": np.argsort(morton3(q), kind="stable"),
                  "octant_then_morton": np.argsort((octant << np.uint64(30)) | morton3(q), kind="stable")}
        for k, perm in orders.items():
            r = np.ascontiguousarray(rays[perm])
            d_r = torch.from_numpy(r.view(np.float32).reshape(-1, 8)).cuda()
            d_h = torch.empty_like(d_r)
            sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr()); ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr())
            ctx.synchronize()
            dt = (time.perf_counter() - t0) / 5
            print(json.dumps({"scene": name, "order": k, "mrays_s": n / dt / 1e6}), flush=True)
        sc.close()


main()
