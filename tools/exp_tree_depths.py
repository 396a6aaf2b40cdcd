import sys, os
sys.path.insert(0, "gpu-pathtracer_amd")
import torch, mipt
from mipt import scenes
ctx = mipt.Context(0)
v, i, m, d = scenes.atrium(262144, seed=42)
n = mipt.build_bvh(mipt.mesh_leaf_bounds(v, i), mipt.bvh_options(**{"bvh.builder": "sah", "mipt.optimize_passes": "2"}))
print("atrium sah+opt2 depth", mipt.Scene(ctx, n, v, i).tree_depth())
print("atrium gpu depth", mipt.Scene(ctx, None, v, i).tree_depth())
v, i, m, d = scenes.cornell_box()
n = mipt.build_bvh(mipt.mesh_leaf_bounds(v, i), mipt.bvh_options(**{"bvh.builder": "sah"}))
print("cornell depth", mipt.Scene(ctx, n, v, i).tree_depth())
v, i, m, d = scenes.clutter(10_000_000, seed=11)
print("clutter10M gpu depth", mipt.Scene(ctx, None, v, i, link_encoding=1).tree_depth())
v, i, m, d = scenes.displaced_sphere(28_000_000, seed=7)
print("sphere28M gpu depth", mipt.Scene(ctx, None, v, i, link_encoding=1).tree_depth())
ctx.set_option("build.optimize_passes", 8)
print("sphere28M gpu opt8 depth", mipt.Scene(ctx, None, v, i, link_encoding=1).tree_depth())
