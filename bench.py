"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]): Sponza-class scene (262,144-triangle procedural
stand-in, seed 42), 1920x1080, Lambert+GGX, depth 8.  A step is one pass of the hot path
over one batch of camera samples (``--spp-per-step`` samples per pixel through all
bounces).  Inputs are resident in HBM before the timed region.  One process per GPU; ranks
render disjoint sample ranges of the same frame (weak scaling) and the film is reduced
over RCCL after the timed steps.

Prints one JSON line (rank 0).  `value` = rays traced by all ranks / wall time (Mrays/s).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(REPO, "gpu-pathtracer_amd"), REPO):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--tris", type=int, default=262144)
    ap.add_argument("--cpu-baseline-rays", type=int, default=400000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    import mipt
    from mipt import scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # ---- scene (host), BVH (host), upload ---------------------------------------------
    vtx, idx, mats, mids = scenes.atrium(args.tris, seed=42)
    t0 = time.time()
    nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx), mipt.bvh_options(**{"bvh.builder": "sah"}))
    build_s = time.time() - t0
    ctx = mipt.Context(local_rank)
    stream = torch.cuda.Stream()          # the library's work and the timing events share it
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    scene = mipt.Scene(ctx, nodes, vtx, idx, mats, mids)

    W, H = args.width, args.height
    n_px = W * H
    cam = scenes.atrium_camera(W / H)
    rays_p = torch.empty((n_px, 8), dtype=torch.float32, device="cuda")
    mipt._check(mipt.lib.pt_raygen_device(ctx._h, cam, W, H, rays_p.data_ptr()))
    sec_host = scenes.surface_rays(vtx, idx, n_px, seed=7 + rank)
    rays_s = torch.from_numpy(sec_host.view(np.float32).reshape(-1, 8)).cuda()
    hits = torch.empty((n_px, 8), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()

    def step():
        scene.trace_closest_device(rays_p.data_ptr(), n_px, hits.data_ptr())
        scene.trace_closest_device(rays_s.data_ptr(), n_px, hits.data_ptr())
        return 2 * n_px

    # ---- work per ray (not timed): counted variant of the same kernels ------------------
    st_p = scene.trace_closest_device_counted(rays_p.data_ptr(), n_px, hits.data_ptr())
    st_s = scene.trace_closest_device_counted(rays_s.data_ptr(), n_px, hits.data_ptr())
    n_box = (st_p.box_tests + st_s.box_tests) / (2 * n_px)
    n_tri = (st_p.tri_tests + st_s.tri_tests) / (2 * n_px)
    bytes_per_ray = 32.0 * n_box + 48.0 * n_tri + 64.0

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    rays_done = 0
    for _ in range(args.steps):
        rays_done += step()
    ev1.record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1)

    t_all = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    r_all = torch.tensor([float(rays_done)], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        dist.all_reduce(r_all, op=dist.ReduceOp.SUM)
    elapsed = float(t_all.item())
    total_rays = float(r_all.item())

    if rank == 0:
        launches = 2 * args.steps
        avg_launch_s = kernel_ms * 1e-3 / launches
        achieved = bytes_per_ray * n_px / avg_launch_s / 1e9
        out = {
            "metric": "Mrays/s (primary+bounce), Sponza-class 1080p",
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "atrium stand-in for Crytek Sponza, %d tris, %dx%d, closest-hit "
                                   "of primary rays + equal count of incoherent surface rays"
                                   % (args.tris, W, H),
                       "rays_per_step_per_gpu": 2 * n_px, "bvh_build_s": round(build_s, 3)},
            "nodes_per_ray": n_box, "tris_per_ray": n_tri, "bytes_per_ray": bytes_per_ray,
            "tie_rays_per_step": int(st_p.tie_rays + st_s.tie_rays),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": None,
                         "kernel": "k_trace_ordered", "avg_launch_ms": avg_launch_s * 1e3},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nodes, vtx, idx, rays_p, rays_s, args.cpu_baseline_rays)
        print(json.dumps(out))
    scene.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(nodes, vtx, idx, rays_p, rays_s, n_sample):
    """The oracle's IntersectSceneClosest (reference skip-link order) over the product's BVH,
    on every host core this process may use, on an evenly strided sample of the step's rays."""
    from oracle import oracle
    oracle.build()
    cores = len(os.sched_getaffinity(0))
    cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    half = n_sample // 2
    n = rays_p.shape[0]
    sel = np.linspace(0, n - 1, half).astype(np.int64)
    rp = rays_p.cpu().numpy()[sel]
    rs = rays_s.cpu().numpy()[sel]
    rays = np.concatenate([rp, rs]).view(oracle.ray_dtype).reshape(-1)
    oracle.closest(nodes, vtx, idx, rays[:2000], nthreads=cores)
    t0 = time.perf_counter()
    _, cnt = oracle.closest(nodes, vtx, idx, rays, nthreads=cores, counters=True)
    dt = time.perf_counter() - t0
    return {"value": rays.shape[0] / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d rays (every %d-th primary + every %d-th surface ray of one step), "
                      "%.1f s" % (rays.shape[0], n // half, n // half, dt),
            "nodes_per_ray": cnt.box_tests / rays.shape[0],
            "tris_per_ray": cnt.tri_tests / rays.shape[0]}


if __name__ == "__main__":
    main()
