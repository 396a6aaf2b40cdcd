"""Closest-hit rate on a two-level scene (one mesh placed many times) beside the same
triangles flattened into one tree built on the GPU.  Primary rays from a camera over the
field and their diffuse bounce rays.
  python tools/bench_instanced.py [n_side=16] [mesh_tris=65536] [option=value ...]"""
import os, sys, json, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "gpu-pathtracer_amd"))
sys.path.insert(0, REPO)
import numpy as np
import torch
import mipt
from mipt import scenes

n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mesh_tris = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
ctx = mipt.Context(0)
for o in sys.argv[3:]:                      # context options, key=value
    k, _, v = o.partition("=")
    ctx.set_option(k, int(v))
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)

# MESH=cube: the uniformly tessellated surface instead of the latitude / longitude one, whose
# pole fans give a few rays hundreds of triangle tests each
gen = scenes.displaced_cube_sphere if os.environ.get("MESH", "uv") == "cube" else scenes.displaced_sphere
vtx, idx, mats, mids = gen(mesh_tris, seed=3)
ext = float(np.abs(vtx).max())
t0 = time.perf_counter()
# BUILD=host (default): the mesh's tree from the host SAH builder; host_splits: the same with
# bvh.sah.use_splits; gpu: no tree from the caller, the library builds it on the GPU
BUILD = os.environ.get("BUILD", "host")
host_opts = {"bvh.builder": "sah"}
if BUILD == "host_splits":
    host_opts["bvh.sah.use_splits"] = "1"
nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx), mipt.bvh_options(**host_opts))
t_mesh = time.perf_counter() - t0
rng = np.random.default_rng(1)
tr = np.zeros((n_side * n_side, 3, 4), dtype=np.float32)
for k in range(n_side * n_side):
    a = rng.uniform(0, 2 * np.pi)
    s = rng.uniform(0.7, 1.2)
    tr[k, :, :3] = s * np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    tr[k, :, 3] = ((k % n_side) * 2.6 * ext, rng.uniform(-0.3, 0.3) * ext, (k // n_side) * 2.6 * ext)
mesh_of = [0] * (n_side * n_side)
meshes = [(nodes, vtx, idx, None)]
top = mipt.build_bvh(mipt.instance_bounds(meshes, tr, mesh_of))
if BUILD == "gpu":
    meshes = [(None, vtx, idx, None)]
t0 = time.perf_counter()
two = mipt.Scene.instanced(ctx, meshes, tr, mesh_of, top)
t_two = time.perf_counter() - t0

W, H = (int(os.environ.get("BENCH_W", 1920)), int(os.environ.get("BENCH_H", 1080)))
span = n_side * 2.6 * ext
cam = mipt.look_at_camera((span * 0.5, span * 0.35, -span * 0.25), (span * 0.5, 0.0, span * 0.45),
                          fovy_deg=50.0, aspect=W / H)
rays = ctx.raygen(cam, W, H)
n = rays.shape[0]
d_r = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
d_h = torch.empty_like(d_r)


def rate(scene, d_rays, reps=3):
    scene.trace_closest_device(d_rays.data_ptr(), n, d_h.data_ptr()); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        scene.trace_closest_device(d_rays.data_ptr(), n, d_h.data_ptr())
    ctx.synchronize()
    return reps * n / (time.perf_counter() - t0) / 1e6


out = {"instances": n_side * n_side, "mesh_triangles": mesh_tris, "mesh_tree": BUILD,
       "mesh_leaves": (len(two.read_mesh_nodes(0)) + 1) // 2,
       "instanced_triangles": n_side * n_side * mesh_tris, "mesh_build_s": round(t_mesh, 2),
       "two_level_create_s": round(t_two, 2)}
out["two_level_primary_mrays_s"] = rate(two, d_r)
stp = two.trace_closest_device_counted(d_r.data_ptr(), n, d_h.data_ptr())
out["primary_boxes_per_ray"] = stp.box_tests / n
out["primary_triangles_per_ray"] = stp.tri_tests / n
out["primary_gbox_tests_s"] = out["two_level_primary_mrays_s"] * out["primary_boxes_per_ray"] / 1e3
two.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr()); ctx.synchronize()
hits = d_h.cpu().numpy().view(mipt.hit_dtype).reshape(-1)
out["primary_hit_fraction"] = float((hits["primid"] >= 0).mean())
# bounce rays: from the hit points, cosine-ish directions about +y jittered (scene is a field)
t = hits["uvwt"][:, 3:4]
o = rays["o"][:, :3] + rays["d"][:, :3] * np.where(hits["primid"][:, None] >= 0, t, 0.0) * 0.999
d = rng.normal(size=(n, 3)).astype(np.float32); d[:, 1] = np.abs(d[:, 1]) * 0.5
d /= np.linalg.norm(d, axis=1, keepdims=True)
b = rays.copy(); b["o"][:, :3] = o; b["d"][:, :3] = d
d_b = torch.from_numpy(b.view(np.float32).reshape(-1, 8)).cuda()
out["two_level_bounce_mrays_s"] = rate(two, d_b)
st = two.trace_closest_device_counted(d_b.data_ptr(), n, d_h.data_ptr())
out["bounce_boxes_per_ray"] = st.box_tests / n
out["bounce_triangles_per_ray"] = st.tri_tests / n
out["bounce_rays_retraced_in_fixed_order"] = st.tie_rays / n
h2 = d_h.cpu().numpy().view(mipt.hit_dtype).reshape(-1).copy()

if n_side * n_side * mesh_tris <= 80_000_000:
    wv = np.concatenate([(vtx[idx.reshape(-1)] @ tr[k, :, :3].T + tr[k, :, 3]).astype(np.float32)
                         for k in range(n_side * n_side)])
    wi = np.arange(wv.shape[0], dtype=np.int32).reshape(-1, 3)
    t0 = time.perf_counter()
    flat = mipt.Scene(ctx, None, wv, wi, link_encoding=mipt.PT_LINKS_INT_BITS)
    out["flat_gpu_build_s"] = round(time.perf_counter() - t0, 2)
    out["flat_primary_mrays_s"] = rate(flat, d_r)
    out["flat_bounce_mrays_s"] = rate(flat, d_b)
    h3 = d_h.cpu().numpy().view(mipt.hit_dtype).reshape(-1)
    both = (h2["primid"] >= 0) & (h3["primid"] >= 0)
    out["bounce_distance_agreement"] = float(np.mean(np.abs(h2["uvwt"][both, 3] - h3["uvwt"][both, 3])
                                                     <= 1e-3 * np.abs(h3["uvwt"][both, 3]) + 1e-3))
print(json.dumps(out))
