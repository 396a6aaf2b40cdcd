"""The N>1 path on the CPU: two gloo ranks shard a frame by tiles and by sample ranges with
the product's own sharding rules (pt_shard_layout, mipt.distributed) and assemble it; the
pixels themselves come from the CPU oracle, since there is no GPU here.  The assembled frame
must equal the single-rank frame (bit-exact for tiles, to rounding for sample ranges)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, TS, SPP, DEPTH = 72, 40, 16, 5, 4


def _scene_and_camera():
    import mipt
    from mipt import scenes
    from oracle import oracle
    vtx, idx, mats, mids = scenes.cornell_box()
    nodes = oracle.build_bvh(oracle.mesh_leaf_bounds(vtx, idx))
    cam = mipt.look_at_camera((278.0, 273.0, -800.0), (278.0, 273.0, 0.0), fovy_deg=39.3, aspect=W / H)
    return mipt, oracle, (nodes, vtx, idx, mats, mids), cam


def _render_shard(mipt, oracle, scene, cam, **kw):
    """What pt_render does for one shard, with the oracle producing the pixels."""
    nodes, vtx, idx, mats, mids = scene
    p = mipt.render_params(W, H, max_depth=DEPTH, seed=3, **kw)
    film = np.zeros((H, W, 4), dtype=np.float32)
    xy = mipt.shard_layout(W, H, p.tile_size, p.shard_index, p.shard_count)
    per_tile = p.tile_size * p.tile_size
    for k in range(xy.shape[0] // per_tile):
        t = xy[k * per_tile:(k + 1) * per_tile]
        t = t[t[:, 0] >= 0]
        if len(t):
            rect = (int(t[:, 0].min()), int(t[:, 1].min()), int(t[:, 0].max()) + 1, int(t[:, 1].max()) + 1)
            oracle.render(nodes, vtx, idx, mats, mids, cam, p, film=film, rect=rect)
    return film, xy


def _worker(rank, world, port, out_dir):
    for p in (os.path.join(REPO, "gpu-pathtracer_amd"), REPO):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mipt, oracle, scene, cam = _scene_and_camera()
        from mipt import distributed as D

        # ---- tiles -----------------------------------------------------------------------
        film, _ = _render_shard(mipt, oracle, scene, cam, sample_count=SPP, **D.tile_shard_kwargs(rank, world, TS))
        layouts = [mipt.shard_layout(W, H, TS, r, world) for r in range(world)]
        ft = torch.from_numpy(film)

        def pack(r, buf):
            xy = layouts[r]
            ok = xy[:, 0] >= 0
            buf.zero_()
            buf[torch.from_numpy(ok)] = ft[xy[ok, 1], xy[ok, 0]]

        def unpack(r, buf):
            xy = layouts[r]
            ok = xy[:, 0] >= 0
            ft[xy[ok, 1], xy[ok, 0]] = buf[torch.from_numpy(ok)]

        # without GPUs the ranks must agree on the torch.distributed fallback, not hang
        comm, how = D.make_comm(None, rank, world)
        assert comm is None and "torch" in how
        calls = []
        D.assemble_tiles([len(l) for l in layouts], pack, unpack, device="cpu",
                         sync=lambda: calls.append(1))
        assert len(calls) == 2
        np.save(os.path.join(out_dir, f"tiles_{rank}.npy"), ft.numpy())

        # ---- one BVH build for all ranks ------------------------------------------------------
        lb = mipt.mesh_leaf_bounds(scene[1], scene[2])
        opt = mipt.bvh_options(**{"bvh.builder": "sah", "mipt.optimize_passes": "1"})
        cache = os.path.join(out_dir, "once.bvh")
        nodes_once, how = D.build_bvh_once(lb, opt, rank, cache_path=cache)
        assert "rank 0" in how and "could not" not in how, how
        dist.barrier()
        assert not os.path.exists(cache)              # rank 0 removed it once every rank had loaded it
        np.save(os.path.join(out_dir, f"nodes_{rank}.npy"), nodes_once.view(np.uint32))

        # ---- sample ranges ------------------------------------------------------------------
        kw = D.sample_shard_kwargs(rank, world, SPP)
        film, _ = _render_shard(mipt, oracle, scene, cam, tile_size=TS, **kw)
        fs = torch.from_numpy(film)
        D.assemble_samples(fs, dst=None)
        np.save(os.path.join(out_dir, f"samples_{rank}.npy"), fs.numpy())
    finally:
        dist.destroy_process_group()


def test_two_ranks_assemble_one_frame(tmp_path):
    world, port = 2, 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for p in (os.path.join(REPO, "gpu-pathtracer_amd"), REPO):
        if p not in sys.path:
            sys.path.insert(0, p)
    mipt, oracle, scene, cam = _scene_and_camera()
    full, _ = _render_shard(mipt, oracle, scene, cam, sample_count=SPP, tile_size=TS)
    assert (full[..., 3] == SPP).all()
    want_nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(scene[1], scene[2]),
                                mipt.bvh_options(**{"bvh.builder": "sah", "mipt.optimize_passes": "1"}))
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"nodes_{r}.npy"), want_nodes.view(np.uint32))
        tiles = np.load(tmp_path / f"tiles_{r}.npy")
        np.testing.assert_array_equal(tiles.view(np.uint32), full.view(np.uint32))
        samples = np.load(tmp_path / f"samples_{r}.npy")
        np.testing.assert_array_equal(samples[..., 3], full[..., 3])
        np.testing.assert_allclose(samples[..., :3], full[..., :3], rtol=2e-6, atol=1e-7)


def test_shard_layout_partitions_the_image():
    for p in (os.path.join(REPO, "gpu-pathtracer_amd"), REPO):
        if p not in sys.path:
            sys.path.insert(0, p)
    import mipt
    from mipt import distributed as D
    for (w, h, ts, world) in [(72, 40, 16, 2), (100, 76, 32, 3), (1920, 1080, 64, 8), (8, 8, 64, 4)]:
        seen = np.zeros((h, w), dtype=np.int32)
        for r in range(world):
            xy = mipt.shard_layout(w, h, ts, r, world)
            assert xy.shape[0] % (ts * ts) == 0
            ok = xy[:, 0] >= 0
            np.add.at(seen, (xy[ok, 1], xy[ok, 0]), 1)
            # consecutive groups of 64 slots are 8x8 pixel blocks (one wavefront's camera rays)
            if ok.all() and len(xy):
                blk = xy[:64]
                assert blk[:, 0].max() - blk[:, 0].min() == 7 and blk[:, 1].max() - blk[:, 1].min() == 7
        assert (seen == 1).all(), "tiles must cover every pixel exactly once"
    parts = [D.sample_shard_kwargs(r, 3, 10) for r in range(3)]
    assert [p["sample_count"] for p in parts] == [4, 3, 3]
    assert [p["sample_begin"] for p in parts] == [0, 4, 7]
    # the benchmark's split of a frame over ranks and steps: every (tile, sample index) is
    # rendered exactly once, whatever the world size, under strong and weak scaling alike
    for world in (1, 2, 3, 8):
        for scaling in ("strong", "weak"):
            for sharding in ("samples", "tiles"):
                spp, steps = 5, 3
                per_step = spp if scaling == "strong" else spp * world
                seen = np.zeros((world if sharding == "tiles" else 1, steps * per_step), dtype=np.int32)
                for r in range(world):
                    for k in range(steps):
                        kw = D.step_shard_kwargs(r, world, k, spp, scaling, sharding, 64)
                        if kw is None:
                            assert sharding == "samples" and spp < world
                            continue
                        row = kw["shard_index"] if sharding == "tiles" else 0
                        if sharding == "tiles":
                            assert kw["shard_count"] == world and kw["tile_size"] == 64
                        seen[row, kw["sample_begin"]:kw["sample_begin"] + kw["sample_count"]] += 1
                assert (seen == 1).all(), (world, scaling, sharding)
    with pytest.raises(mipt.MiptError):
        mipt.shard_layout(64, 64, 12, 0, 1)
    with pytest.raises(mipt.MiptError):
        mipt.shard_layout(64, 64, 16, 2, 2)


def _run_bench(args, env=None):
    import subprocess
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True,
                          text=True, timeout=120, env=e)


def test_bench_without_enough_gpus_prints_one_json_error():
    """`python bench.py --gpus N` as the round driver starts it, on a box with fewer GPUs: one JSON
    line with the reason and a non-zero exit code, no traceback."""
    out = _run_bench(["--gpus", "2"])
    assert out.returncode == 2 and "Traceback" not in out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    import json
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 2 and "GPU" in d["error"] and d["gpus_visible"] == 0
    # started under a launcher with another rank count than --gpus says
    out = _run_bench(["--gpus", "1"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode == 2 and "Traceback" not in out.stderr
    assert "started 2 ranks" in json.loads(out.stdout.splitlines()[-1])["error"]


def test_bench_starts_its_own_ranks_and_relays_their_line(monkeypatch, capsys):
    """The self-launch path with two ranks on the CPU: bench.py, told that two GPUs are visible,
    starts `python -m torch.distributed.run --nproc-per-node 2` of itself; the ranks find no GPU,
    rank 0 says so in one JSON line, and the parent relays that line and a non-zero exit code."""
    import importlib
    import json
    monkeypatch.syspath_prepend(REPO)
    bench = importlib.import_module("bench")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    rc = bench.main()
    assert rc != 0
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 2 and "has no GPU" in d["error"]
