"""Where a tree's surface-area cost sits: the sum of inner-node areas (over the root's) by the
number of leaves under the node, for the host SAH tree and the GPU-built tree of one scene.
env SCENE=cube|uv, TRIS, PASSES."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "gpu-pathtracer_amd"))
import numpy as np
import torch
import mipt
from mipt import scenes


def by_scale(nodes):
    raw = nodes.view(np.float32).reshape(-1, 8)
    n = raw.shape[0]
    e = raw[:, 4:7].astype(np.float64) - raw[:, 0:3].astype(np.float64)
    area = e[:, 0] * e[:, 1] + e[:, 0] * e[:, 2] + e[:, 1] * e[:, 2]
    inner = raw[:, 3].view(np.int32) == -1
    skip = raw[:, 7].view(np.int32).astype(np.int64)
    skip[skip < 0] = n
    leaves = (skip - np.arange(n) + 1) // 2
    lg = np.floor(np.log2(np.maximum(leaves, 1))).astype(np.int64)
    out = np.zeros(int(lg.max()) + 1)
    np.add.at(out, lg[inner], area[inner] / area[0])
    leaf_cost = float(area[~inner].sum() / area[0])
    return [round(float(x), 3) for x in out], round(float(out.sum()), 2), round(leaf_cost, 2)


ctx = mipt.Context(0)
SCENE = os.environ.get("SCENE", "cube")
if SCENE == "atrium":                         # the headline benchmark's scene
    vtx, idx, _, _ = scenes.atrium(int(os.environ.get("TRIS", "262144")), seed=42)
else:
    gen = scenes.displaced_cube_sphere if SCENE == "cube" else scenes.displaced_sphere
    vtx, idx, _, _ = gen(int(os.environ.get("TRIS", "2000000")), seed=7)
rays = scenes.surface_rays(vtx, idx, 1 << 20, seed=7)
d_r = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
d_h = torch.empty_like(d_r)


def tests_per_ray(sc):
    ctx.set_option("extend.quad", 0)
    st = sc.trace_closest_device_counted(d_r.data_ptr(), 1 << 20, d_h.data_ptr())
    ctx.set_option("extend.quad", -1)
    return round(st.box_tests / (1 << 20), 2), round(st.tri_tests / (1 << 20), 2)


lb = mipt.mesh_leaf_bounds(vtx, idx)
for passes in ([] if os.environ.get("NOHOST") else [0, 2]):
    nodes = mipt.build_bvh(lb, mipt.bvh_options(**{"bvh.builder": "sah", "mipt.link_encoding": "int",
                                                    "mipt.optimize_passes": passes}))
    sc = mipt.Scene(ctx, nodes, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS)
    h, tot, lc = by_scale(nodes)
    print(json.dumps({"tree": "host sah + %d passes" % passes, "inner": tot, "leaf": lc, "per_ray": tests_per_ray(sc),
                      "depth": sc.tree_depth(), "by_log2_leaves": h}), flush=True)
    sc.close()
ctx.set_option("build.split_ratio", int(os.environ.get("SPLIT", "0")))
ctx.set_option("build.ploc_radius", int(os.environ.get("PLOC", "0")))
for passes in os.environ.get("PASSES", "0 8 8/0 24").split():      # passes[/treelet_every]
    ctx.set_option("build.optimize_passes", int(passes.split("/")[0]))
    ctx.set_option("build.treelet_every", int(passes.split("/")[1]) if "/" in passes else 4)
    ctx.synchronize(); t0 = time.perf_counter()
    sc = mipt.Scene(ctx, None, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS)
    ctx.synchronize(); build_s = round(time.perf_counter() - t0, 3)
    h, tot, lc = by_scale(sc.read_nodes())
    print(json.dumps({"tree": "gpu + %s" % passes, "build_s": build_s, "inner": tot, "leaf": lc, "per_ray": tests_per_ray(sc),
                      "depth": sc.tree_depth(), "by_log2_leaves": h}), flush=True)
    sc.close()
