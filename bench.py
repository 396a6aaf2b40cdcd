"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (the configuration the headline metric is quoted on): Sponza-class
scene (262,144-triangle procedural stand-in for Crytek Sponza, seed 42 -- no assets exist
offline), 1920x1080, Lambert+GGX materials, max depth 8, Russian roulette from bounce 3.
A step is one pass of the whole hot path -- raygen, (extend, shade) x depth, accumulate --
over one batch of --spp-per-step samples per pixel; 256 spp is 256/spp-per-step steps.
Scene, BVH and film are resident in HBM before the timed region.

Multi-GPU: one process per GPU, scene replicated, ranks render disjoint sample ranges of the
same frame (weak scaling: work per rank fixed) with no collective on the data path; the film
is summed once over RCCL (reduce to rank 0) at the end of the timed region.

Prints one JSON line (rank 0): value = rays traced by all ranks / wall time, in Mrays/s.
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
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--tris", type=int, default=262144)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--spp-per-step", type=int, default=4)
    ap.add_argument("--sharding", choices=["samples", "tiles"], default="samples",
                    help="how N ranks split the frame (N > 1)")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=INT",
                    help="pt_context_set_option tuning knob, e.g. extend.persistent=0")
    ap.add_argument("--bvh-option", action="append", default=[], metavar="KEY=VALUE",
                    help="extra pt_bvh_options_set pairs for the host BVH build (e.g. mipt.optimize_passes=2)")
    ap.add_argument("--nee", action="store_true",
                    help="next-event estimation: one shadow ray per path vertex (counted in value)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-scale", type=int, default=1,
                    help="the CPU baseline renders the frame at 1/scale resolution in x and y")
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
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         "python -m torch.distributed.run --nproc-per-node N")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # ---- host side: scene, BVH; then upload ------------------------------------------------
    vtx, idx, mats, mids = scenes.atrium(args.tris, seed=42)
    t0 = time.time()
    # binned SAH as the reference configures its builder (src/main_granite.cpp:176), then two
    # rounds of insertion-based optimisation of the finished tree (untimed, like the build)
    bvh_opts = {"bvh.builder": "sah", "mipt.optimize_passes": "2"}
    bvh_opts.update(dict(o.split("=", 1) for o in args.bvh_option))
    nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx), mipt.bvh_options(**bvh_opts))
    build_s = time.time() - t0
    ctx = mipt.Context(local_rank)
    stream = torch.cuda.Stream()          # the library's kernels and the timing share this stream
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    for opt in args.option:
        key, _, val = opt.partition("=")
        ctx.set_option(key, int(val))
    scene = mipt.Scene(ctx, nodes, vtx, idx, mats, mids)
    W, H, spp = args.width, args.height, args.spp_per_step
    cam = scenes.atrium_camera(W / H)
    film_t = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    film = mipt.Film(ctx, W, H, device_ptr=film_t.data_ptr())
    sky = (0.55, 0.65, 0.8)

    from mipt import distributed as D

    base_flags = mipt.PT_RENDER_NEE if args.nee else 0

    def params(step, flags=0):
        flags |= base_flags
        common = dict(max_depth=args.depth, seed=42, rr_start_depth=3, background=sky, flags=flags)
        if args.sharding == "tiles":
            # rank r renders the tiles it owns with world*spp samples: same work per rank
            return mipt.render_params(W, H, sample_begin=step * world * spp, sample_count=world * spp,
                                      **D.tile_shard_kwargs(rank, world, 64), **common)
        # rank r renders samples [(step*world + r)*spp, +spp) of every pixel
        return mipt.render_params(W, H, sample_begin=(step * world + rank) * spp, sample_count=spp,
                                  **common)

    def assemble():
        if world == 1:
            return
        if args.sharding == "tiles":
            sizes = [film.packed_size(64, r, world) for r in range(world)]
            D.assemble_tiles(sizes, lambda r, buf: film.pack_tiles(64, r, world, buf.data_ptr()),
                             lambda r, buf: film.unpack_tiles(64, r, world, buf.data_ptr()), "cuda")
        else:
            D.assemble_samples(film_t, dst=0)

    # ---- work per ray (untimed): the same kernels with their test counters compiled in ------
    st = mipt.render(ctx, scene, cam, params(0, mipt.PT_RENDER_COUNT_WORK), film, stats=True)
    rays0 = st.rays_primary + st.rays_bounce          # the counters cover the closest-hit kernel only
    n_box, n_tri = st.box_tests / rays0, st.tri_tests / rays0
    bytes_per_ray = 32.0 * n_box + 48.0 * n_tri + 64.0        # reference-layout bytes per ray
    film.clear()

    ctx.set_kernel_timing(True)
    for k in range(args.warmup):
        mipt.render(ctx, scene, cam, params(1000 + k), film, stats=True)
    if args.warmup:
        assemble()                # the collective's first call sets up its channels: not timed
    film.clear()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rays = paths = ties = launches = shadow = 0
    extend_ms = 0.0
    for k in range(args.steps):
        st = mipt.render(ctx, scene, cam, params(k), film, stats=True)
        rays += st.rays_primary + st.rays_bounce
        shadow += st.rays_shadow
        paths += st.paths
        ties += st.tie_rays
        launches += st.extend_launches
        extend_ms += st.extend_ms
    assemble()                                                 # frame assembly over RCCL
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    agg = torch.tensor([elapsed, float(rays + shadow), float(paths)], dtype=torch.float64, device="cuda")
    if world > 1:
        mx = agg.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0].item())
    total_rays, total_paths = float(agg[1].item()), float(agg[2].item())

    if rank == 0:
        img = film_t[..., :3] / film_t[..., 3:].clamp(min=1)
        achieved = bytes_per_ray * rays / (extend_ms * 1e-3) / 1e9
        out = {
            "metric": "Mrays/s (primary+bounce%s), Sponza 1080p depth 8" % ("+shadow" if args.nee else ""),
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "msamples_per_s": total_paths / elapsed / 1e6,
            "config": {"workload": "Sponza-class atrium stand-in (%d tris, seed 42), %dx%d, "
                                   "Lambert+GGX, max depth %d, %d spp per step per GPU; full "
                                   "wavefront path tracer (raygen, extend, shade, accumulate)"
                                   % (args.tris, W, H, args.depth, spp),
                       "rays_per_step_per_gpu": (rays + shadow) / args.steps,
                       "shadow_rays_per_step_per_gpu": shadow / args.steps,
                       "next_event_estimation": bool(args.nee),
                       "samples_per_pixel_total": args.steps * spp * world,
                       "bvh": ", ".join("%s=%s" % kv for kv in sorted(bvh_opts.items())),
                       "bvh_build_s": round(build_s, 3),
                       "sharding": ("sample ranges per rank, RCCL reduce of the film at the end"
                                    if args.sharding == "samples" else
                                    "64x64 tiles round-robin over ranks, RCCL all-gather of packed tiles")},
            "nodes_per_ray": n_box, "tris_per_ray": n_tri, "bytes_per_ray": bytes_per_ray,
            "tie_rays": int(ties), "mean_radiance": float(img.mean().item()),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": None,
                         "kernel": "k_pt_extend_persistent<false>", "launches": int(launches),
                         "avg_launch_ms": extend_ms / max(launches, 1),
                         "kernel_time_share": extend_ms * 1e-3 / elapsed,
                         "note": "algorithmic reference-layout bytes; the scene's 46 MB of traversal "
                                 "data are cache resident, so this is not HBM traffic and the "
                                 "fraction can exceed 1",
                         # what the kernel really does: dependent gathers of 64-byte node and
                         # 48-byte triangle records; beside it the rate of a bare gather kernel
                         # (tools/micro/gather_nodes.hip, every lane of 5 resident workgroups
                         # per CU chasing random 64-byte records) measured on this GPU model
                         "records_per_s": (n_box / 2 + n_tri) * rays / (extend_ms * 1e-3),
                         "gather_microbenchmark_records_per_s": {"16 MB table": 133e9, "2 MB table": 212e9}},
        }
        # HBM-side bytes per launch of that kernel come from hardware counters, which a
        # benchmark cannot read on itself: they are taken from the committed rocprofv3 --pmc
        # passes over this same command (profiles/collect_traffic.sh) when the workload matches.
        tpath = os.path.join(REPO, "profiles", "r01_traffic.json")
        default_workload = (args.tris, W, H, args.depth, spp, args.nee) == (262144, 1920, 1080, 8, 4, False)
        if os.path.exists(tpath) and default_workload and world == 1:
            with open(tpath) as f:
                tr = json.load(f)
            out["roofline"]["traffic"] = tr["traffic_bytes_per_launch"] / 1e9
            out["roofline"]["traffic_unit"] = "GB per launch (2 x FETCH_SIZE + WRITE_SIZE, profiles/r01_traffic.json)"
            out["roofline"]["algorithmic_gb_per_launch"] = bytes_per_ray * rays / max(launches, 1) / 1e9
            if "counters" in tr:
                out["roofline"]["counters"] = tr["counters"]
        if not args.no_cpu_baseline and world == 1:      # the host-core baseline belongs to the 1-GPU line
            out["cpu_baseline"] = cpu_baseline(args, nodes, vtx, idx, mats, mids, sky)
            out["traversal_counts_same_rays"] = traversal_counts(scene, nodes, vtx, idx)
        print(json.dumps(out), flush=True)
    film.close()
    scene.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, nodes, vtx, idx, mats, mids, sky):
    """The CPU restatement of the same step -- the reference's IntersectSceneClosest in
    skip-link order plus this project's shading spec -- over the product's BVH, on the host
    cores this process may use, on the same frame at reduced resolution."""
    import mipt
    from mipt import scenes
    from oracle import oracle
    oracle.build()
    cores = len(os.sched_getaffinity(0))
    cores = max(1, min(cores, int(os.environ.get("OMP_NUM_THREADS", cores))))
    s = max(1, args.cpu_baseline_scale)
    w, h = args.width // s, args.height // s
    cam = scenes.atrium_camera(w / h)
    steps = 3                                   # about 10 s of work on 16 cores
    p = mipt.render_params(w, h, sample_begin=0, sample_count=steps * args.spp_per_step,
                           max_depth=args.depth, seed=42, rr_start_depth=3, background=sky,
                           flags=mipt.PT_RENDER_NEE if args.nee else 0)
    t0 = time.perf_counter()
    _, rays = oracle.render(nodes, vtx, idx, mats, mids, cam, p, nthreads=cores)
    dt = time.perf_counter() - t0
    # one core, on a quarter of that sample again
    w1, h1 = max(w // 2, 8), max(h // 2, 8)
    p1 = mipt.render_params(w1, h1, sample_begin=0, sample_count=args.spp_per_step, max_depth=args.depth,
                            seed=42, rr_start_depth=3, background=sky,
                            flags=mipt.PT_RENDER_NEE if args.nee else 0)
    t1 = time.perf_counter()
    _, rays1 = oracle.render(nodes, vtx, idx, mats, mids, scenes.atrium_camera(w1 / h1), p1, nthreads=1)
    dt1 = time.perf_counter() - t1
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model,
            "single_core_value": rays1 / dt1 / 1e6,
            "single_core_sample": "one step at %dx%d: %d rays in %.1f s" % (w1, h1, rays1, dt1),
            "sample": "%d steps of the same frame at %dx%d (1/%d of the pixels), %d spp each, depth %d: "
                      "%d rays in %.1f s" % (steps, w, h, s * s, args.spp_per_step, args.depth, rays, dt)}


def traversal_counts(scene, nodes, vtx, idx, n=1 << 18):
    """Slab and triangle tests per ray of the product's ordered traversal and of the
    reference's skip-link order (CPU restatement), on one set of secondary rays."""
    import torch
    from mipt import scenes
    from oracle import oracle
    rays = scenes.surface_rays(vtx, idx, n, seed=7)
    d_r = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
    d_h = torch.empty_like(d_r)
    st = scene.trace_closest_device_counted(d_r.data_ptr(), n, d_h.data_ptr())
    cores = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", 64))))
    _, cnt = oracle.closest(nodes, vtx, idx, rays, nthreads=cores, counters=True)
    return {"rays": n, "ordered": {"box_tests_per_ray": st.box_tests / n, "tri_tests_per_ray": st.tri_tests / n},
            "skip_link_oracle": {"box_tests_per_ray": cnt.box_tests / n, "tri_tests_per_ray": cnt.tri_tests / n}}


if __name__ == "__main__":
    main()
