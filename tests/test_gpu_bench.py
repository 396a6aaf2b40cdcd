"""bench.py keeps its contract: one JSON line with the metric, the roofline records (fractions
at most 1, measured ceilings from the same run) and the CPU baseline.  Run small, as a child
process, exactly as the benchmark is started for one GPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_small(tmp_path):
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1",
           "--tris", "65536", "--width", "320", "--height", "180", "--spp-per-step", "4",
           "--hbm-tris", "200000", "--hbm-rays", "262144", "--cpu-baseline-scale", "4", "--no-live-counters"]
    env = dict(os.environ, OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "roofline_hbm", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(d["config"]["samples_per_pixel_total"] - 8.0) < 1e-6      # 2 steps x 4 spp reached the film
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert 0 < r["frac"] <= 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None          # counter traffic is only copied for the workload it was collected on
    ceil = r["gather_ceilings_measured_in_this_run_grecords_s"]
    assert len(ceil) == 4 and all(v > 0 for v in ceil.values())
    h = d["roofline_hbm"]
    assert h["bound"].startswith("hbm") and h["unit"] == "Gfetches/s" and h["peak"] > 0
    # the fetch rate needs the counter passes, which this run was told to skip
    assert h["achieved"] is None and h["frac"] is None and h["traffic"] is None
    assert abs(h["bytes_per_ray"] - (32 * h["nodes_per_ray"] + 48 * h["tris_per_ray"] + 64)) < 1e-6
    assert abs(h["algorithmic_gbs_over_hbm_peak"] - h["algorithmic_gbs"] / 8000.0) < 1e-9
    lm = h["launch_ms"]
    assert len(lm["each_ms"]) == h["launches"] == 7 and lm["min_ms"] <= h["avg_launch_ms"] <= lm["max_ms"]
    o = h["same_surface_other_tessellation"]
    assert "latitude" in o["workload"] and o["leaves"] >= 200000 * 0.9 and o["avg_launch_ms"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mrays/s" and c["sample"]
    k = d["kernel_ms_per_step"]
    assert k["extend"] > 0 and k["shade"] > 0 and k["raygen"] > 0 and k["accumulate"] > 0
    assert sum(k.values()) * 1e-3 * d["steps"] <= d["ms_per_step"] * 1e-3 * d["steps"] * 1.001


@pytest.mark.parametrize("sharding,assembly", [("samples", "abi"), ("tiles", "abi"), ("samples", "torch"),
                                               ("tiles", "torch")])
def test_bench_multi_gpu_branches_with_one_rank(tmp_path, sharding, assembly):
    """The branches that only several ranks take -- torch.distributed over RCCL, the library's
    communicator made from a broadcast id, the strong-scaling split, frame assembly inside the
    timed region, the reductions of the result -- run here with one rank, started by the launcher
    exactly as a multi-GPU run is."""
    port = 29600 + (os.getpid() + hash((sharding, assembly))) % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(REPO, "bench.py"), "--gpus", "1", "--force-distributed", "--steps", "2", "--warmup", "1",
           "--tris", "65536", "--width", "320", "--height", "180", "--spp-per-step", "4",
           "--sharding", sharding, "--assembly", assembly, "--no-hbm-record", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["config"]["ranks_in_communicator"] == 1
    if assembly == "abi":
        assert "C ABI" in d["config"]["assembly"], d["config"]["assembly"]
    else:
        assert d["config"]["assembly"].startswith("torch.distributed")
    assert abs(d["config"]["samples_per_pixel_total"] - 8.0) < 1e-6      # every sample reached the film once


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` with no launcher around it, as the round driver starts it: the
    script starts its ranks itself (a child torch.distributed.run) before touching a GPU and relays
    rank 0's line.  Rehearsed with one rank (--force-distributed takes the same path); asking for
    more GPUs than the box has gives one JSON line with the reason and a non-zero exit code."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR",
                                                              "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--force-distributed", "--steps", "2",
           "--warmup", "1", "--tris", "65536", "--width", "320", "--height", "180", "--spp-per-step", "4",
           "--no-hbm-record", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=str(tmp_path), env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["config"]["launched_by"].startswith("bench.py itself")
    assert "C ABI" in d["config"]["assembly"] and d["config"]["ranks_in_communicator"] == 1
    assert abs(d["config"]["samples_per_pixel_total"] - 8.0) < 1e-6

    import torch
    n = torch.cuda.device_count() + 1
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n)], capture_output=True,
                         text=True, timeout=120, cwd=str(tmp_path), env=env)
    assert out.returncode != 0 and "Traceback" not in out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == n and "visible" in d["error"]


@pytest.mark.parametrize("kind", ["obj", "glb", "scene_dir"])
def test_bench_renders_a_scene_file(tmp_path, kind):
    """--scene FILE (or PT_SCENE_DIR holding sponza.*): the benchmark renders the asset through
    pt_obj_load / pt_gltf_load, the loaders standing in for tinyobj (src/main.cpp:852-899) and
    Granite's scene loader (src/main_granite.cpp:265-298), and says so in its line."""
    import shutil
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("PT_SCENE_DIR", None)
    args = []
    if kind == "obj":
        path = os.path.join(REPO, "tests", "golden", "cornell_box.obj")
        args = ["--scene", path]
    elif kind == "glb":
        import test_gltf
        path = str(test_gltf._document(tmp_path, "glb")[0])
        args = ["--scene", path]
    else:
        d = tmp_path / "assets"
        d.mkdir()
        for ext in ("obj", "mtl"):
            shutil.copy(os.path.join(REPO, "tests", "golden", "cornell_box." + ext), d / ("sponza." + ext if ext == "obj" else "cornell_box.mtl"))
        path = str(d / "sponza.obj")
        env["PT_SCENE_DIR"] = str(d)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--width", "160",
           "--height", "90", "--spp-per-step", "4", "--no-hbm-record", "--no-cpu-baseline", "--no-live-counters"] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=str(tmp_path), env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["data"] == "file" and os.path.basename(path) in d["config"]["workload"]
    assert d["value"] > 0 and abs(d["config"]["samples_per_pixel_total"] - 8.0) < 1e-6
    assert d["mean_radiance"] > 0


@pytest.mark.parametrize("sharding,ranks", [("samples", 2), ("tiles", 3)])
def test_bench_rehearses_several_ranks_on_one_gpu(tmp_path, sharding, ranks):
    """--gpus N --rehearse-on-one-gpu: the N ranks the script starts itself all render on GPU 0 and
    exchange over gloo -- the multi-rank control flow (one BVH build per node, shards per rank,
    barriers, the frame assembled on rank 0, maxima over ranks, one line from rank 0) end to end
    on the one GPU a test box has.  The assembled frame is the one-rank frame."""
    common = ["--steps", "3", "--warmup", "1", "--tris", "65536", "--width", "320", "--height", "192",
              "--spp-per-step", "6", "--no-hbm-record", "--no-cpu-baseline", "--no-live-counters"]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    one = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + common, capture_output=True,
                         text=True, timeout=280, cwd=str(tmp_path), env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    ref = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(ranks),
                          "--rehearse-on-one-gpu", "--assembly", "torch", "--sharding", sharding] + common,
                         capture_output=True, text=True, timeout=280, cwd=str(tmp_path), env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["config"]["rehearsal_on_one_gpu"] is True
    assert "bench.py itself" in d["config"]["launched_by"] and d["config"]["assembly"].startswith("torch")
    assert d["scaling"] == "strong" and d["value"] > 0
    # the same frame: every sample of every step reached rank 0's film
    assert abs(d["config"]["samples_per_pixel_total"] - ref["config"]["samples_per_pixel_total"]) < 1e-6
    assert abs(d["mean_radiance"] - ref["mean_radiance"]) <= 1e-5 * ref["mean_radiance"]
    assert "built by rank 0" in d["config"]["bvh_source"] or "rank" in d["config"]["bvh_source"]


def test_bench_rehearsal_falls_back_when_the_communicator_cannot_come_up(tmp_path):
    """Two ranks on one GPU asking for the library's own communicator: the 128-byte id travels from
    rank 0 to rank 1 and both call ncclCommInitRank, which RCCL turns down (two ranks, one
    device).  Both ranks must learn of it, agree, and assemble with torch.distributed instead;
    the line says which way the frame went."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu",
                          "--assembly", "abi", "--comm-timeout", "30", "--steps", "2", "--warmup", "1",
                          "--tris", "65536", "--width", "320", "--height", "192", "--spp-per-step", "4",
                          "--no-hbm-record", "--no-cpu-baseline", "--no-live-counters"],
                         capture_output=True, text=True, timeout=280, cwd=str(tmp_path), env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert d["config"]["assembly"].startswith("torch.distributed") and "did not come up" in d["config"]["assembly"]
    assert abs(d["config"]["samples_per_pixel_total"] - 8.0) < 1e-6


def test_bench_on_a_tree_built_on_the_gpu(tmp_path):
    """--bvh gpu: the scene comes from pt_scene_build instead of pt_bvh_build + pt_scene_create;
    the line says so, and the CPU baseline walks the tree read back from the scene (the same
    frame, so the same mean radiance up to the order of equal-distance hits)."""
    common = ["--steps", "2", "--warmup", "1", "--tris", "65536", "--width", "320", "--height", "180",
              "--spp-per-step", "4", "--no-hbm-record", "--cpu-baseline-scale", "4", "--no-live-counters"]
    lines = {}
    for bvh in ("host", "gpu"):
        out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--bvh", bvh] + common,
                             capture_output=True, text=True, timeout=280, cwd=str(tmp_path))
        assert out.returncode == 0, out.stderr[-2000:]
        got = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(got) == 1, out.stdout[-2000:]
        lines[bvh] = json.loads(got[0])
    g, h = lines["gpu"], lines["host"]
    assert "GPU" in g["config"]["bvh_source"] and "GPU" not in h["config"]["bvh_source"]
    assert g["value"] > 0 and g["cpu_baseline"]["value"] > 0
    assert abs(g["mean_radiance"] - h["mean_radiance"]) <= 1e-3 * h["mean_radiance"]
    assert abs(g["nodes_per_ray"] - h["nodes_per_ray"]) < 0.15 * h["nodes_per_ray"]


def test_bench_measures_hbm_traffic_itself(tmp_path):
    """roofline.traffic of a default run is not copied from a file: the benchmark runs itself again
    under rocprofv3 --pmc (one counter per pass) and reports the bytes those passes counted."""
    import shutil
    if not (shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3")):
        pytest.skip("rocprofv3 is not installed")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1",
           "--tris", "65536", "--width", "320", "--height", "180", "--spp-per-step", "4",
           "--hbm-tris", "200000", "--hbm-rays", "262144", "--no-cpu-baseline", "--no-hbm-companion"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    r, h = d["roofline"], d["roofline_hbm"]
    assert "live_counters_failed" not in r, r.get("live_counters_failed")
    assert r["traffic"] is not None and r["traffic"] > 0 and "measured by this run" in r["traffic_source"]
    assert h["traffic"] is not None and h["traffic"] > 0 and "measured by this run" in h["traffic_source"]
    # at least the rays read and the hits written cross the memory interface
    assert r["hbm_frac_from_traffic"] > 0 and h["hbm_frac_from_traffic"] > 0
    # the second record's fraction: fetches from memory per second over this run's own gather rate
    assert h["achieved"] > 0 and abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-9
