"""Does running two half-size renders concurrently (two contexts, two streams) beat one
full-size render?  The closest-hit kernel's tail (waves finishing at different times) and the
shade kernel (HBM-bound) could fill each other's gaps."""
import os, sys, time, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "gpu-pathtracer_amd"))
import numpy as np
import torch
import mipt
from mipt import scenes

vtx, idx, mats, mids = scenes.atrium(262144, seed=42)
nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx), mipt.bvh_options(**{"bvh.builder": "sah"}))
W, H = 1920, 1080
cam = scenes.atrium_camera(W / H)
sky = (0.55, 0.65, 0.8)
ctxs = [mipt.Context(0) for _ in range(2)]
scs = [mipt.Scene(c, nodes, vtx, idx, mats, mids) for c in ctxs]
films = [mipt.Film(c, W, H) for c in ctxs]


def params(begin, count):
    return mipt.render_params(W, H, sample_begin=begin, sample_count=count, max_depth=8, seed=42,
                              rr_start_depth=3, background=sky)


def run(split, steps=16):
    for c in ctxs:
        c.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        if split:
            mipt.render(ctxs[0], scs[0], cam, params(4 * k, 2), films[0])
            mipt.render(ctxs[1], scs[1], cam, params(4 * k + 2, 2), films[1])
        else:
            mipt.render(ctxs[0], scs[0], cam, params(4 * k, 4), films[0])
    for c in ctxs:
        c.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


run(False, 2); run(True, 2)
print(json.dumps({"one_stream_ms_per_step": run(False), "two_streams_ms_per_step": run(True),
                  "one_stream_again": run(False), "two_streams_again": run(True)}))
