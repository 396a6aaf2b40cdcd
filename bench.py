"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (the configuration the headline metric is quoted on): Sponza-class scene
(262,144-triangle procedural stand-in for Crytek Sponza, seed 42 -- no assets exist offline;
`--scene FILE` or `PT_SCENE_DIR` holding sponza.obj / sponza.glb renders the real asset
instead), 1920x1080, Lambert+GGX materials, max depth 8, Russian roulette from bounce 3.
A step is one pass of the whole hot path -- raygen, (extend, shade) x depth, accumulate -- over
--spp-per-step samples per pixel of the frame; the default step is 64 spp (a quarter of the
256-spp frame the metric is quoted on; the default 40 steps are ten such frames, about 5 s), rendered in passes of
at most 32 M paths.
Scene, BVH and film are resident in HBM before the timed region.

Multi-GPU: one process per GPU, scene replicated, no collective on the data path.  Started
without a launcher (`python bench.py --gpus N`, no WORLD_SIZE in the environment), the script
starts its N ranks itself -- a child `python -m torch.distributed.run` of this same file, before
anything here has touched a GPU -- and relays rank 0's JSON line and the exit code; when fewer
than N GPUs are visible it prints a one-line JSON error and exits non-zero.  The host BVH build
runs once (rank 0 saves the tree, the others load it).  Default is
strong scaling: the same frame (same total samples per step) is split over the ranks, by
sample ranges or by 64x64 tiles, and assembled once over RCCL at the end of the timed region
through the library's C ABI (pt_film_reduce / pt_film_gather_tiles), so value = rays of the
frame / time to the assembled frame.  --scaling weak keeps the work per rank fixed instead.

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

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--tris", type=int, default=262144)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--spp-per-step", type=int, default=64,
                    help="samples per pixel of one step (of the whole frame with --scaling strong, "
                         "of each rank with --scaling weak)")
    ap.add_argument("--samples-per-pass", type=int, default=0,
                    help="pt_render_params.samples_per_pass (0: library default)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--sharding", choices=["samples", "tiles"], default="samples",
                    help="how N ranks split the frame (N > 1)")
    ap.add_argument("--assembly", choices=["abi", "torch"], default="abi",
                    help="frame assembly through the library's RCCL entry points or torch.distributed")
    ap.add_argument("--scene", default=None, help="OBJ / glTF / GLB file to render instead of the stand-in")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=INT",
                    help="pt_context_set_option tuning knob, e.g. extend.persistent=0")
    ap.add_argument("--bvh", choices=("host", "gpu"), default="host",
                    help="host: bvh::build_bvh's counterpart pt_bvh_build (binned SAH + 2 insertion rounds) and "
                         "pt_scene_create; gpu: pt_scene_build (radix tree + insertion and treelet rounds)")
    ap.add_argument("--bvh-option", action="append", default=[], metavar="KEY=VALUE",
                    help="extra pt_bvh_options_set pairs for the host BVH build (e.g. mipt.optimize_passes=2)")
    ap.add_argument("--nee", action="store_true",
                    help="next-event estimation: one shadow ray per path vertex (counted in value)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="run the --gpus N ranks on GPU 0 all together, with gloo collectives staged through host "
                         "memory: a rehearsal of the multi-rank control flow (sharding, single BVH build, "
                         "barriers, frame assembly, the line) on a one-GPU box; its value is no measurement")
    ap.add_argument("--force-distributed", action="store_true",
                    help="with one rank: still bring up torch.distributed, the communicator and the frame "
                         "assembly as for several (rehearses the multi-GPU branches on a one-GPU box)")
    ap.add_argument("--comm-timeout", type=float, default=120.0,
                    help="seconds a rank waits for the library's RCCL communicator to come up before every "
                         "rank falls back to torch.distributed assembly")
    ap.add_argument("--no-live-counters", action="store_true",
                    help="do not measure HBM-side traffic with rocprofv3 child passes (the committed "
                         "profile's figure is copied instead, labelled so)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-record", action="store_true",
                    help="skip the second roofline record (28 M-triangle scene, HBM-resident traversal)")
    ap.add_argument("--no-hbm-companion", "--no-hbm-uniform", dest="no_hbm_companion", action="store_true",
                    help="skip the second record's companion measurement on the other tessellation of the surface")
    ap.add_argument("--hbm-scene", choices=["cube", "uv"], default="cube",
                    help="tessellation of the second record's closed surface: a cube's faces pushed out to "
                         "the sphere (triangles of nearly uniform size, as a scanned mesh has them), or latitude "
                         "/ longitude (two pole fans of sliver triangles); the other one is the companion")
    ap.add_argument("--hbm-launches", type=int, default=7,
                    help="timed launches per ray order of the second record (median reported, all listed)")
    ap.add_argument("--hbm-tris", type=int, default=28_000_000)
    ap.add_argument("--hbm-rays", type=int, default=1 << 24)
    ap.add_argument("--cpu-baseline-scale", type=int, default=1,
                    help="the CPU baseline renders the frame at 1/scale resolution in x and y")
    return ap.parse_args()


def find_scene_file(args):
    """--scene, or the Sponza asset in PT_SCENE_DIR if there is one."""
    if args.scene:
        return args.scene
    d = os.environ.get("PT_SCENE_DIR")
    if d:
        for name in ("sponza.obj", "sponza.glb", "sponza.gltf", "Sponza.obj", "Sponza.glb", "Sponza.gltf"):
            p = os.path.join(d, name)
            if os.path.exists(p):
                return p
    return None


def load_workload(args, mipt, scenes):
    """(vtx, idx, materials, material_ids, camera factory, description, data)."""
    path = find_scene_file(args)
    if path is None:
        vtx, idx, mats, mids = scenes.atrium(args.tris, seed=42)
        return (vtx, idx, mats, mids, scenes.atrium_camera,
                "Sponza-class atrium stand-in (%d tris, seed 42)" % idx.shape[0], "synthetic")
    # the fields the reference takes from its loader (src/main.cpp:852-899,
    # src/main_granite.cpp:265-298), through pt_obj_load / pt_gltf_load
    m = mipt.Mesh(path)
    lo, hi = m.vtx.min(axis=0), m.vtx.max(axis=0)
    centre, ext = (lo + hi) / 2, (hi - lo)
    axis = int(np.argmax(ext[[0, 2]])) * 2                    # look along the longer floor axis
    eye = centre.copy()
    eye[axis] -= 0.35 * ext[axis]
    eye[1] = lo[1] + 0.3 * ext[1]
    target = centre.copy()
    target[1] = lo[1] + 0.35 * ext[1]

    def camera(aspect):
        return mipt.look_at_camera(tuple(eye), tuple(target), fovy_deg=55.0, aspect=aspect)
    mats, mids = (m.materials, m.material_ids) if len(m.materials) else (None, None)
    return (m.vtx, m.idx, mats, mids, camera,
            "%s (%d tris)" % (os.path.basename(path), m.idx.shape[0]), "file")


def error_line(args, message, **extra):
    """One JSON line that says why there is no measurement (stdout, like the result line)."""
    out = {"metric": "Mrays/s (primary+bounce), Sponza 1080p depth 8", "value": None, "unit": "Mrays/s",
           "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "error": message}
    out.update(extra)
    print(json.dumps(out), flush=True)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child
    `python -m torch.distributed.run` of this file and relay its output and exit code.  Nothing in
    this process touches a GPU before or after (counting devices does not initialise one), so the
    child is an ordinary subprocess, never a replacement of a process that holds the GPU."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < (1 if args.rehearse_on_one_gpu else args.gpus):
        error_line(args, "--gpus %d but %d GPU%s visible" % (args.gpus, have, "" if have == 1 else "s"),
                   gpus_visible=have)
        return 2
    with socket.socket() as s:                       # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    # the launcher sets OMP_NUM_THREADS=1 for its workers when it is unset; the ranks size their
    # host threads from MIPT_HOST_THREADS instead (this process's CPU share, divided among them)
    env.setdefault("MIPT_HOST_THREADS", str(max(1, host_cores() // max(1, args.gpus))))
    env["MIPT_BENCH_LAUNCHED_BY"] = "bench.py itself (child torch.distributed.run, %d ranks)" % args.gpus
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    got_line = False
    for line in proc.stdout:
        got_line = got_line or line.startswith("{")
        sys.stdout.write(line)
        sys.stdout.flush()
    rc = proc.wait()
    if rc != 0 and not got_line:
        error_line(args, "the %d-rank child run ended with exit code %d (its stderr is above)" % (args.gpus, rc))
    return rc if rc >= 0 else 1


def host_cores():
    """Host threads this process may use: its CPU affinity, capped by OMP_NUM_THREADS and
    MIPT_HOST_THREADS when set (a GPU box shows every CPU of the machine, not the share of one GPU)."""
    n = len(os.sched_getaffinity(0))
    for var in ("MIPT_HOST_THREADS", "OMP_NUM_THREADS"):      # the first one set decides
        v = os.environ.get(var, "")
        if v.isdigit() and int(v) > 0:
            return max(1, min(n, int(v)))
    return max(1, n)


def main():
    args = parse_args()
    if args.gpus < 1:
        error_line(args, "--gpus must be at least 1")
        return 2
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.force_distributed):
        return self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            error_line(args, "--gpus %d but the launcher started %d rank%s" % (args.gpus, world, "" if world == 1 else "s"))
        return 2
    import torch
    import torch.distributed as dist
    import mipt
    from mipt import scenes
    from mipt import distributed as D

    rehearsal = args.rehearse_on_one_gpu
    if rehearsal:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        if rank == 0:
            error_line(args, "rank %d has no GPU: %d visible" % (local_rank, torch.cuda.device_count()),
                       gpus_visible=torch.cuda.device_count())
        return 2
    torch.cuda.set_device(local_rank)
    multi = world > 1 or args.force_distributed      # collectives and frame assembly in play
    if multi and rehearsal:
        dist.init_process_group("gloo")
    elif multi:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # ---- host side: scene, BVH; then upload ------------------------------------------------
    vtx, idx, mats, mids, make_camera, scene_name, data_kind = load_workload(args, mipt, scenes)
    t0 = time.time()
    # binned SAH as the reference configures its builder (src/main_granite.cpp:176), then two
    # rounds of insertion-based optimisation of the finished tree (untimed, like the build)
    bvh_opts = {"bvh.builder": "sah", "mipt.optimize_passes": "2", "mipt.num_threads": str(host_cores())}
    if 2 * idx.shape[0] - 1 > (1 << 24):
        bvh_opts["mipt.link_encoding"] = "int"
    bvh_opts.update(dict(o.split("=", 1) for o in args.bvh_option))
    link_enc = mipt.PT_LINKS_INT_BITS if bvh_opts.get("mipt.link_encoding") == "int" else mipt.PT_LINKS_FLOAT_VALUE
    leaf_bounds = mipt.mesh_leaf_bounds(vtx, idx)
    if args.bvh == "gpu":
        nodes, bvh_source = None, "built on the GPU by this process (pt_scene_build)"
    elif world > 1:
        nodes, bvh_source = D.build_bvh_once(leaf_bounds, mipt.bvh_options(**bvh_opts), rank)
    else:
        nodes = mipt.build_bvh(leaf_bounds, mipt.bvh_options(**bvh_opts))
        bvh_source = "built by this process"
    build_s = time.time() - t0
    ctx = mipt.Context(local_rank)
    stream = torch.cuda.Stream()          # the library's kernels and the timing share this stream
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    for opt in args.option:
        key, _, val = opt.partition("=")
        ctx.set_option(key, int(val))
    if nodes is None:
        ctx.set_option("build.optimize_passes", 8)
        ctx.synchronize()
        t0 = time.time()
    scene = mipt.Scene(ctx, nodes, vtx, idx, mats, mids, link_encoding=link_enc)
    if nodes is None:
        ctx.synchronize()
        build_s = time.time() - t0
        ctx.set_option("build.optimize_passes", 6)
        nodes = scene.read_nodes()          # what the CPU legs walk
    W, H, spp = args.width, args.height, args.spp_per_step
    cam = make_camera(W / H)
    film_t = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    film = mipt.Film(ctx, W, H, device_ptr=film_t.data_ptr())
    sky = (0.55, 0.65, 0.8)
    base_flags = mipt.PT_RENDER_NEE if args.nee else 0
    strong = args.scaling == "strong"

    # ---- the communicator of the frame assembly ---------------------------------------------
    comm, assembly = None, "none"
    if multi:
        comm, assembly = D.make_comm(ctx, rank, world, prefer_abi=args.assembly == "abi",
                                     timeout_s=args.comm_timeout)

    def params(step, flags=0):
        """pt_render_params of this rank for one step, or None if it has nothing to render."""
        shard = D.step_shard_kwargs(rank, world, step, spp, args.scaling, args.sharding, 64)
        if shard is None:
            return None
        return mipt.render_params(W, H, max_depth=args.depth, seed=42, rr_start_depth=3, background=sky,
                                  flags=flags | base_flags, samples_per_pass=args.samples_per_pass, **shard)

    def assemble():
        if not multi:
            return
        if comm is not None:
            if args.sharding == "tiles":
                comm.gather_tiles(film, 64, root=0)
            else:
                comm.reduce_film(film, root=0)
            ctx.synchronize()
        elif args.sharding == "tiles" and rehearsal:
            sizes = [film.packed_size(64, r, world) for r in range(world)]

            def pack_host(r, buf):                  # through a device buffer: gloo moves host memory
                tmp = torch.empty(buf.shape, dtype=torch.float32, device="cuda")
                torch.cuda.current_stream().synchronize()
                film.pack_tiles(64, r, world, tmp.data_ptr())
                ctx.synchronize()
                buf.copy_(tmp)

            def unpack_host(r, buf):
                tmp = buf.cuda()
                torch.cuda.current_stream().synchronize()
                film.unpack_tiles(64, r, world, tmp.data_ptr())
                ctx.synchronize()

            D.assemble_tiles(sizes, pack_host, unpack_host, "cpu")
        elif args.sharding == "tiles":
            sizes = [film.packed_size(64, r, world) for r in range(world)]
            D.assemble_tiles(sizes, lambda r, buf: film.pack_tiles(64, r, world, buf.data_ptr()),
                             lambda r, buf: film.unpack_tiles(64, r, world, buf.data_ptr()), "cuda",
                             sync=ctx.synchronize)
        elif rehearsal:
            ctx.synchronize()
            host = film_t.cpu()
            D.assemble_samples(host, dst=0)
            film_t.copy_(host)
            torch.cuda.synchronize()
        else:
            ctx.synchronize()
            D.assemble_samples(film_t, dst=0)

    def render_step(k, flags=0):
        p = params(k, flags)
        return mipt.render(ctx, scene, cam, p, film, stats=True) if p is not None else mipt.RenderStats()

    # ---- work per ray (untimed): the same kernels with their test counters compiled in ------
    st = render_step(0, mipt.PT_RENDER_COUNT_WORK)
    rays0 = max(st.rays_primary + st.rays_bounce, 1)  # the counters cover the closest-hit kernel only
    n_box, n_tri = st.box_tests / rays0, st.tri_tests / rays0
    n_box_lds = st.lds_box_tests / rays0             # of n_box: on nodes served from the LDS copy of the top levels
    bytes_per_ray = 32.0 * n_box + 48.0 * n_tri + 64.0        # reference-layout bytes per ray
    film.clear()

    # ---- what this GPU can gather (untimed): dependent random 64-byte fetches, the traversal's
    # access pattern, from tables sized to each level of the memory hierarchy -----------------
    ceilings = {}
    if rank == 0:
        scene_bytes = max(64 * (idx.shape[0] - 1) + 48 * idx.shape[0], 1 << 16)
        for name, size in (("l1_resident_16KB", 16 << 10), ("l2_resident_2MB", 2 << 20),
                           ("scene_sized_%dMB" % max(1, scene_bytes >> 20), scene_bytes),
                           ("hbm_resident_4GB", 4 << 30)):
            ceilings[name] = ctx.measure_gather(size, 64, 400) / 1e9

    ctx.set_kernel_timing(True)
    for k in range(args.warmup):
        render_step(1000 + k)
    if args.warmup:
        assemble()                # the collective's first call sets up its channels: not timed
    film.clear()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rays = paths = ties = launches = shadow = 0
    kernel_ms = {"extend": 0.0, "raygen": 0.0, "shade": 0.0, "occlude": 0.0, "accumulate": 0.0}
    for k in range(args.steps):
        st = render_step(k)
        rays += st.rays_primary + st.rays_bounce
        shadow += st.rays_shadow
        paths += st.paths
        ties += st.tie_rays
        launches += st.extend_launches
        for name in kernel_ms:
            kernel_ms[name] += getattr(st, name + "_ms")
    assemble()                                                 # frame assembly over RCCL
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    extend_ms = kernel_ms["extend"]

    agg = torch.tensor([elapsed, float(rays + shadow), float(paths)], dtype=torch.float64,
                       device="cpu" if rehearsal else "cuda")
    if multi:
        mx = agg.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0].item())
    total_rays, total_paths = float(agg[1].item()), float(agg[2].item())

    if rank == 0:
        img = film_t[..., :3] / film_t[..., 3:].clamp(min=1)
        frame_spp = float(film_t[..., 3].mean().item())
        ext_s = max(extend_ms, 1e-9) * 1e-3
        # 64-byte nodes + 48-byte triangle records fetched through the vector memory pipeline (the
        # nodes of the top levels that the workgroups keep in LDS are counted apart)
        records = ((n_box - n_box_lds) / 2 + n_tri) * rays
        records_lds = n_box_lds / 2 * rays
        out = {
            "metric": "Mrays/s (primary+bounce%s), Sponza 1080p depth 8" % ("+shadow" if args.nee else ""),
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32", "data": data_kind,
            "msamples_per_s": total_paths / elapsed / 1e6,
            "config": {"workload": "%s, %dx%d, Lambert+GGX, max depth %d, %d spp per step%s; full "
                                   "wavefront path tracer (raygen, extend, shade, accumulate)"
                                   % (scene_name, W, H, args.depth, spp,
                                      "" if world == 1 else (" of the frame, split over the ranks" if strong
                                                             else " per GPU")),
                       "rays_per_step_per_gpu": (rays + shadow) / args.steps,
                       "shadow_rays_per_step_per_gpu": shadow / args.steps,
                       "next_event_estimation": bool(args.nee),
                       "samples_per_pixel_total": frame_spp,
                       "bvh": ", ".join("%s=%s" % kv for kv in sorted(bvh_opts.items())),
                       "bvh_build_s": round(build_s, 3), "bvh_source": bvh_source,
                       "host_threads_per_rank": host_cores(),
                       "rehearsal_on_one_gpu": bool(rehearsal),
                       "launched_by": os.environ.get("MIPT_BENCH_LAUNCHED_BY", "a launcher (WORLD_SIZE was set)"
                                                     if "WORLD_SIZE" in os.environ else "directly, one process"),
                       "sharding": "none (one GPU)" if world == 1 else
                                   ("sample ranges per rank, film summed on rank 0 at the end"
                                    if args.sharding == "samples" else
                                    "64x64 tiles round-robin over ranks, packed tiles sent to rank 0 at the end"),
                       "assembly": assembly, "ranks_in_communicator": comm.size if comm is not None else world},
            "nodes_per_ray": n_box, "tris_per_ray": n_tri, "bytes_per_ray": bytes_per_ray,
            "nodes_per_ray_from_lds": n_box_lds, "tree_depth": scene.tree_depth(),
            "tie_rays": int(ties), "mean_radiance": float(img.mean().item()),
            "kernel_ms_per_step": {k: v / args.steps for k, v in kernel_ms.items()},
            "kernel_time_share_of_step": sum(kernel_ms.values()) * 1e-3 / elapsed,
        }
        # The dominant kernel.  Its scene data (tens of MB) are cache resident, so HBM is not what
        # bounds it: it is a dependent gather of 64-byte records through the L1 address/data path.
        # achieved = records fetched per second; peak = the best rate at which a bare gather
        # kernel (pt_measure_gather: same grid, same loads, no arithmetic) fetched records from a
        # cache-resident table on this GPU a moment ago.  The algorithmic reference-layout bytes
        # and their HBM-peak ratio are kept beside it as context, not as the fraction.
        peak_rec = max(v for k, v in ceilings.items() if not k.startswith("hbm"))
        out["roofline"] = {
            "bound": "l1-gather (scene cache resident; dependent 64 B record fetches)",
            "kernel": "k_pt_extend_persistent<false>",
            "achieved": records / ext_s / 1e9, "peak": peak_rec, "unit": "Grecords/s",
            "frac": records / ext_s / 1e9 / peak_rec,
            "traffic": None,
            "records_served_from_lds_grecords_s": records_lds / ext_s / 1e9,
            "launches": int(launches), "avg_launch_ms": extend_ms / max(launches, 1),
            "kernel_time_share": extend_ms * 1e-3 / elapsed,
            "gather_ceilings_measured_in_this_run_grecords_s": ceilings,
            # uniform random gathers over a table of the scene's size: the kernel's fetches are
            # not uniform (upper levels are hot), so this ratio may exceed 1 and is not the fraction
            "achieved_over_scene_sized_gather": records / ext_s / 1e9 / max(
                [v for k, v in ceilings.items() if k.startswith("scene_sized")] or [float("inf")]),
            "algorithmic_gb_per_launch": bytes_per_ray * rays / max(launches, 1) / 1e9,
            "algorithmic_gbs": bytes_per_ray * rays / ext_s / 1e9,
            "algorithmic_gbs_over_hbm_peak": bytes_per_ray * rays / ext_s / 1e9 / HBM_PEAK_GBS,
            "note": "algorithmic bytes are reference-layout bytes (32 B per box test, 48 B per "
                    "triangle test, 64 B per ray) served from caches here; their ratio to HBM peak "
                    "is not a roofline fraction and may exceed 1",
        }
        # HBM-side bytes per launch of that kernel come from hardware counters, which a process
        # cannot read on itself: they are measured by child passes of this script under
        # rocprofv3 --pmc (live_traffic), or, failing that, copied from the committed passes over
        # the same command when the workload matches, and labelled either way.
        live = {}
        if world == 1 and not multi and not args.no_live_counters:
            live = live_traffic(args, want_hbm=not args.no_hbm_record)
        rays_per_launch = rays / max(launches, 1)
        if live.get("extend", {}).get("FETCH_SIZE") is not None and "WRITE_SIZE" in live["extend"]:
            gb = traffic_bytes(live["extend"], rays_per_launch) / 1e9
            out["roofline"]["traffic"] = gb
            out["roofline"]["traffic_source"] = (
                "GB per launch, measured by this run: two child passes of this script under rocprofv3 --pmc "
                "(FETCH_SIZE, WRITE_SIZE; %d launches each) on the same workload, two steps"
                % live["extend"]["launches_counted"])
            out["roofline"]["hbm_frac_from_traffic"] = gb / (extend_ms / max(launches, 1) * 1e-3) / HBM_PEAK_GBS
        elif live.get("error"):
            out["roofline"]["live_counters_failed"] = live["error"]
        tpath = os.path.join(REPO, "profiles", "r03_traffic.json")
        if os.path.exists(tpath) and world == 1:
            with open(tpath) as f:
                tr = json.load(f)
            same = tr.get("workload") == [args.tris, W, H, args.depth, spp, args.samples_per_pass,
                                          bool(args.nee), data_kind]
            if same and out["roofline"]["traffic"] is None:
                gb = tr["traffic_bytes_per_launch"] / 1e9
                out["roofline"]["traffic"] = gb
                out["roofline"]["traffic_source"] = ("GB per launch, copied from profiles/r03_traffic.json (rocprofv3 "
                                                     "--pmc passes over this command; FETCH_SIZE calibrated there "
                                                     "on 64-byte gathers of known size), not measured by this run")
                out["roofline"]["hbm_frac_from_traffic"] = gb / (extend_ms / max(launches, 1) * 1e-3) / HBM_PEAK_GBS
            if same and "counters" in tr:
                out["roofline"]["counters_copied_from_profiles"] = tr["counters"]
        if not args.no_hbm_record and world == 1:
            out["roofline_hbm"] = hbm_record(args, ctx, torch, mipt, scenes, live=live.get("trace"))
            if not args.no_hbm_companion:
                # The same surface in the other tessellation (latitude / longitude: at each pole
                # thousands of sliver triangles meet in one vertex; the GPU builder puts each
                # sliver into several leaves, without which a few rays through a fan make
                # thousands of tests each and hold every launch for milliseconds).
                other = "uv" if args.hbm_scene == "cube" else "cube"
                u = hbm_record(args, ctx, torch, mipt, scenes, kind=other)
                out["roofline_hbm"]["same_surface_other_tessellation"] = {
                    k: u[k] for k in ("workload", "avg_launch_ms", "launch_ms", "mrays_per_s_kernel",
                                      "mrays_per_s_call_incl_sort", "nodes_per_ray", "tris_per_ray", "bytes_per_ray",
                                      "four_child_walk", "queue_order", "hit_fraction", "tie_rays", "tree_depth",
                                      "leaves", "algorithmic_gbs_over_hbm_peak")}
        if link_enc != mipt.PT_LINKS_FLOAT_VALUE:
            out["cpu_baseline_skipped"] = "the CPU restatement of the renderer reads float-valued links only"
        elif not args.no_cpu_baseline and world == 1:    # the host-core baseline belongs to the 1-GPU line
            out["cpu_baseline"] = cpu_baseline(args, nodes, vtx, idx, mats, mids, sky, make_camera)
            out["traversal_counts_same_rays"] = traversal_counts(scene, nodes, vtx, idx)
        print(json.dumps(out), flush=True)
    film.close()
    scene.close()
    if comm is not None:
        comm.close()
    ctx.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if D.abandoned_threads():
        # a communicator that never came up left its thread inside RCCL: do not wait for it
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    return 0


def live_traffic(args, want_hbm):
    """HBM-side bytes per launch of the two dominant kernels, measured now: this script run again
    as a child under `rocprofv3 --pmc`, one counter per pass as the profiler requires (FETCH_SIZE,
    WRITE_SIZE), on the same workload with two timed steps.  Returns {"extend": {...}, "trace":
    {...}} with the mean counter values per launch, or {} with a reason when the profiler is not
    there or a pass fails -- the caller then falls back to the committed profile's figure."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return {"error": "rocprofv3 not found"}
    child = [sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
             "--no-live-counters", "--no-hbm-companion", "--width", str(args.width), "--height", str(args.height),
             "--tris", str(args.tris), "--depth", str(args.depth), "--spp-per-step", str(args.spp_per_step),
             "--samples-per-pass", str(args.samples_per_pass), "--hbm-tris", str(args.hbm_tris),
             "--hbm-rays", str(args.hbm_rays), "--hbm-scene", args.hbm_scene, "--hbm-launches", "3"]
    if not want_hbm:
        child.append("--no-hbm-record")
    if args.nee:
        child.append("--nee")
    if args.scene:
        child += ["--scene", os.path.abspath(args.scene)]
    for o in args.option:
        child += ["--option", o]
    for o in args.bvh_option:
        child += ["--bvh-option", o]
    out = {"extend": {}, "trace": {}}
    work = tempfile.mkdtemp(prefix="mipt_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(work, counter)
            cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "c", "--"] + child
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            except subprocess.TimeoutExpired:
                return {"error": "rocprofv3 --pmc %s timed out" % counter}
            if r.returncode != 0:
                return {"error": "rocprofv3 --pmc %s: exit %d: %s" % (counter, r.returncode, r.stderr[-300:])}
            rows = []
            for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as f:
                    rows += [x for x in csv.DictReader(f) if x.get("Counter_Name") == counter]
            rows.sort(key=lambda x: int(x.get("Dispatch_Id", 0)))
            ext = [float(x["Counter_Value"]) for x in rows if "k_pt_extend_persistent<false" in x["Kernel_Name"]]
            # the uncounted trace kernel on the rays in origin-cell order: the first four launches
            # of the variant without the LDS copy of the top (four-child nodes), or of the only
            # variant there is (two-child nodes)
            names = sorted({x["Kernel_Name"] for x in rows if "k_trace_ordered_persistent<false" in x["Kernel_Name"]})
            pick = [n for n in names if "<false, false, false, true>" in n] or names
            trc = [float(x["Counter_Value"]) for x in rows if pick and x["Kernel_Name"] == pick[0]][:4]
            if not ext:
                return {"error": "no launch of the extend kernel in the %s pass" % counter}
            out["extend"][counter] = sum(ext) / len(ext)
            out["extend"]["launches_counted"] = len(ext)
            if trc:
                out["trace"][counter] = sum(trc) / len(trc)
                out["trace"]["launches_counted"] = len(trc)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return out


def traffic_bytes(c, rays_per_launch):
    """FETCH_SIZE and WRITE_SIZE count kilobytes.  For dependent 64-byte gathers FETCH_SIZE is the
    byte count (checked against a gather of known size and against L2 misses x 64 B); the ray
    queue is a wide streaming read of 32 B per ray, of which it shows half."""
    return c["FETCH_SIZE"] * 1024.0 + 16.0 * rays_per_launch + c["WRITE_SIZE"] * 1024.0


def hbm_record(args, ctx, torch, mipt, scenes, kind=None, live=None):
    """The regime beyond the caches: a 28 M-triangle closed surface (3.6 GB of node and triangle
    records, tree built on the GPU) and 2^24 incoherent secondary rays through
    pt_trace_closest_device.  What bounds the kernel there is the rate at which the memory system
    serves dependent 64-byte fetches scattered over gigabytes, so that is what the record prices:
    achieved = the launch's fetches from memory (FETCH_SIZE of this run's counter passes / 64 B)
    per second, peak = what a bare gather kernel (pt_measure_gather: same grid, same loads, no
    arithmetic) fetches per second from a table of the scene's own footprint, measured in this
    run.  The HBM-peak fraction of the counted traffic and the reference-layout ("algorithmic")
    bytes are reported beside it as context.  live: {"FETCH_SIZE": KB, "WRITE_SIZE": KB,
    "launches_counted": n} per launch of this kernel, or None."""
    t0 = time.time()
    kind = kind or args.hbm_scene
    gen = scenes.displaced_cube_sphere if kind == "cube" else scenes.displaced_sphere
    vtx, idx, _, _ = gen(args.hbm_tris, seed=7)
    n = args.hbm_rays
    rays = scenes.surface_rays(vtx, idx, n, seed=7)
    gen_s = time.time() - t0
    ctx.set_option("build.optimize_passes", 6)      # with the default treelet round after every third
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sc = mipt.Scene(ctx, None, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS)
    ctx.synchronize()
    build_s = time.perf_counter() - t0
    leaves = (int(mipt.lib.pt_scene_node_count(sc._h)) + 1) // 2
    d_r = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
    d_h = torch.empty_like(d_r)
    # The reference-layout bytes are priced with the slab and triangle tests of the two-child walk
    # of this tree (the fewer of the two forms); the four-child walk that is timed tests more
    # child boxes (its coarse boxes admit more) and is not credited for them.
    ctx.set_option("extend.quad", 0)
    stc = sc.trace_closest_device_counted(d_r.data_ptr(), n, d_h.data_ptr())
    ctx.set_option("extend.quad", -1)
    stq = sc.trace_closest_device_counted(d_r.data_ptr(), n, d_h.data_ptr())
    n_box, n_tri = stc.box_tests / n, stc.tri_tests / n
    bpr = 32.0 * n_box + 48.0 * n_tri + 64.0
    ctx.set_kernel_timing(True)
    res = {}
    reps = max(1, args.hbm_launches)
    for name, sort in (("origin_cell_order", 1), ("queue_order", 0)):
        ctx.set_option("trace.sort_rays", sort)
        sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr())
        first_ms = ctx.last_trace_ms()
        ctx.synchronize()
        k_ms = []
        t0 = time.perf_counter()
        for _ in range(reps):
            sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr())
            k_ms.append(ctx.last_trace_ms())
        ctx.synchronize()
        res[name] = {"kernel_ms": float(np.median(k_ms)), "min_ms": min(k_ms), "max_ms": max(k_ms),
                     "first_untimed_launch_ms": first_ms, "each_ms": [round(x, 3) for x in k_ms],
                     "call_ms": (time.perf_counter() - t0) / reps * 1e3}
    ctx.set_option("trace.sort_rays", -1)
    hit_frac = float((d_h.view(torch.int32)[:, 1] >= 0).float().mean().item())
    depth = sc.tree_depth()
    sc.close()
    # what this GPU fetches per second from a table as large as the scene's records, 64 bytes at a
    # time, with nothing else to do
    footprint = 64 * (leaves - 1) + 64 * leaves
    gather = ctx.measure_gather(footprint, 64, 400) / 1e9
    best = res["origin_cell_order"]
    sec = best["kernel_ms"] * 1e-3
    alg = bpr * n / sec / 1e9
    rec = {"bound": "hbm (dependent 64 B fetches scattered over gigabytes: fetch rate, not bandwidth)",
           "kernel": "k_trace_ordered_persistent (four-child nodes with quantised boxes: the form "
                     "scenes beyond the caches are walked in); nodes_per_ray / tris_per_ray / bytes_per_ray "
                     "are those of the two-child walk of the same tree on the same rays",
           "workload": "Lucy-class displaced closed surface (%s), %d tris in %d leaves (%.2f GB of 64-byte node "
                       "and triangle records), %d secondary rays (origins on the surface, cosine directions, "
                       "seed 7), tree built on the GPU with 6 insertion and 2 treelet passes in %.2f s" %
                       ("latitude / longitude tessellation: two pole fans of thousands of sliver triangles, each "
                        "sliver cut into several leaves" if kind == "uv"
                        else "cube faces pushed out to the sphere: triangles of nearly uniform size",
                        idx.shape[0], leaves, footprint / 1e9, n, build_s),
           "achieved": None, "peak": gather, "unit": "Gfetches/s", "frac": None, "traffic": None,
           "peak_source": "pt_measure_gather on a %.2f GB table (the scene's footprint), this run" % (footprint / 1e9),
           "launches": reps, "avg_launch_ms": best["kernel_ms"],
           "launch_ms": {k: best[k] for k in ("min_ms", "max_ms", "first_untimed_launch_ms", "each_ms")},
           "nodes_per_ray": n_box, "tris_per_ray": n_tri, "bytes_per_ray": bpr,
           "four_child_walk": {"child_boxes_tested_per_ray": stq.box_tests / n,
                               "tris_per_ray": stq.tri_tests / n},
           "algorithmic_gb_per_launch": bpr * n / 1e9, "algorithmic_gbs": alg,
           "algorithmic_gbs_over_hbm_peak": alg / HBM_PEAK_GBS,
           "note": "algorithmic bytes are reference-layout bytes (32 B per box test, 48 B per triangle test, 64 B "
                   "per ray); the caches serve the upper levels of the tree, so their ratio to HBM peak is context, "
                   "not the fraction",
           "mrays_per_s_kernel": n / best["kernel_ms"] / 1e3,
           "mrays_per_s_call_incl_sort": n / best["call_ms"] / 1e3,
           "queue_order": {"avg_launch_ms": res["queue_order"]["kernel_ms"],
                           "launch_ms": {k: res["queue_order"][k] for k in ("min_ms", "max_ms",
                                                                            "first_untimed_launch_ms", "each_ms")},
                           "mrays_per_s": n / res["queue_order"]["kernel_ms"] / 1e3},
           "hit_fraction": hit_frac, "tie_rays": int(stc.tie_rays), "tree_depth": depth, "leaves": leaves,
           "scene_and_rays_generated_s": round(gen_s, 1)}
    if live and live.get("FETCH_SIZE") is not None and "WRITE_SIZE" in live:
        fetches = live["FETCH_SIZE"] * 1024.0 / 64.0
        rec["achieved"] = fetches / sec / 1e9
        rec["frac"] = rec["achieved"] / gather
        rec["achieved_source"] = ("FETCH_SIZE of this run's rocprofv3 --pmc child pass over the same kernel (%d launches), "
                                  "in 64-byte fetches, / the median launch time" % live["launches_counted"])
        rec["traffic"] = traffic_bytes(live, float(n)) / 1e9
        rec["traffic_source"] = ("GB per launch, measured by this run: child passes of this script under "
                                 "rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE; %d launches each)" % live["launches_counted"])
        rec["hbm_frac_from_traffic"] = rec["traffic"] / sec / HBM_PEAK_GBS
    else:
        rec["achieved_source"] = ("no counter pass in this run (rocprofv3 missing, --no-live-counters, or a companion "
                                  "measurement): fetch rate not measured; see profiles/ for the committed passes")
    return rec


def cpu_baseline(args, nodes, vtx, idx, mats, mids, sky, make_camera):
    """The CPU restatement of the same step -- the reference's IntersectSceneClosest in
    skip-link order plus this project's shading spec -- over the product's BVH, on the host
    cores this process may use, on the same frame at reduced resolution."""
    import mipt
    from oracle import oracle
    oracle.build()
    cores = host_cores()
    s = max(1, args.cpu_baseline_scale)
    w, h = args.width // s, args.height // s
    cam = make_camera(w / h)
    n_spp = 12                                  # about 10 s of work on 16 cores
    p = mipt.render_params(w, h, sample_begin=0, sample_count=n_spp,
                           max_depth=args.depth, seed=42, rr_start_depth=3, background=sky,
                           flags=mipt.PT_RENDER_NEE if args.nee else 0)
    t0 = time.perf_counter()
    _, rays = oracle.render(nodes, vtx, idx, mats, mids, cam, p, nthreads=cores)
    dt = time.perf_counter() - t0
    # one core, on a quarter of the pixels and a third of the samples
    w1, h1 = max(w // 2, 8), max(h // 2, 8)
    p1 = mipt.render_params(w1, h1, sample_begin=0, sample_count=4, max_depth=args.depth,
                            seed=42, rr_start_depth=3, background=sky,
                            flags=mipt.PT_RENDER_NEE if args.nee else 0)
    t1 = time.perf_counter()
    _, rays1 = oracle.render(nodes, vtx, idx, mats, mids, make_camera(w1 / h1), p1, nthreads=1)
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
            "single_core_sample": "4 spp at %dx%d: %d rays in %.1f s" % (w1, h1, rays1, dt1),
            "sample": "the same frame at %dx%d (1/%d of the pixels), %d spp, depth %d: "
                      "%d rays in %.1f s" % (w, h, s * s, n_spp, args.depth, rays, dt)}


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
    cores = host_cores()
    _, cnt = oracle.closest(nodes, vtx, idx, rays, nthreads=cores, counters=True)
    return {"rays": n, "ordered": {"box_tests_per_ray": st.box_tests / n, "tri_tests_per_ray": st.tri_tests / n},
            "skip_link_oracle": {"box_tests_per_ray": cnt.box_tests / n, "tri_tests_per_ray": cnt.tri_tests / n}}


if __name__ == "__main__":
    sys.exit(main())
