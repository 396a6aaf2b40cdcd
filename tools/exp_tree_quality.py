"""SAH cost (sum of inner-node surface areas over the root's) of the trees the builders make."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "gpu-pathtracer_amd"))
import numpy as np
import torch
import mipt
from mipt import scenes


def sah(nodes):
    raw = nodes.view(np.float32).reshape(-1, 8)
    e = raw[:, 4:7].astype(np.float64) - raw[:, 0:3].astype(np.float64)
    area = e[:, 0] * e[:, 1] + e[:, 0] * e[:, 2] + e[:, 1] * e[:, 2]
    inner = raw[:, 3].view(np.int32) == -1
    return float(area[inner].sum() / area[0]), float(area[~inner].sum() / area[0])


ctx = mipt.Context(0)
for name, (vtx, idx, _, _) in (("atrium", scenes.atrium(262144, seed=42)), ("soup", scenes.random_soup(200000, seed=3))):
    lb = mipt.mesh_leaf_bounds(vtx, idx)
    out = {"scene": name}
    out["host_sah"] = sah(mipt.build_bvh(lb, mipt.bvh_options(**{"bvh.builder": "sah", "mipt.link_encoding": "int"})))
    out["host_median"] = sah(mipt.build_bvh(lb, mipt.bvh_options(**{"bvh.builder": "median", "mipt.link_encoding": "int"})))
    for r in (0, 2, 8, 16, 32):
        ctx.set_option("build.ploc_radius", r)
        sc = mipt.Scene(ctx, None, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS)
        out["gpu_r%d" % r] = sah(sc.read_nodes())
        sc.close()
    print(json.dumps(out), flush=True)
