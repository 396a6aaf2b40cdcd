"""Occlusion queries over two-child and four-child nodes on scenes beyond the caches: 2^24 shadow
rays (origins on the surface, towards random points of the scene's box, reach = that distance)
through pt_trace_any_device, and the occlude kernel of a render pass with next-event estimation.
  SCENE=clutter|cube  TRIS=10000000  python tools/exp_any_quad.py [option=value ...]"""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "gpu-pathtracer_amd"))
import numpy as np
import torch
import mipt
from mipt import scenes

ctx = mipt.Context(0)
for a in sys.argv[1:]:
    k_, _, v_ = a.partition("=")
    ctx.set_option(k_, int(v_))
kind = os.environ.get("SCENE", "clutter")
tris = int(os.environ.get("TRIS", "10000000"))
gen = {"clutter": scenes.clutter, "cube": scenes.displaced_cube_sphere, "uv": scenes.displaced_sphere}[kind]
vtx, idx, mats, mids = gen(tris, seed=11 if kind == "clutter" else 7)
ctx.set_option("build.optimize_passes", 8)
sc = mipt.Scene(ctx, None, vtx, idx, mats, mids, link_encoding=mipt.PT_LINKS_INT_BITS)
ctx.set_option("build.optimize_passes", 0)
n = 1 << 24
rays = scenes.surface_rays(vtx, idx, n, seed=7)
rng = np.random.default_rng(5)
lo, hi = vtx.min(axis=0), vtx.max(axis=0)
target = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
d = target - rays["o"][:, :3]
dist = np.linalg.norm(d, axis=1).astype(np.float32)
rays["d"][:, :3] = d / dist[:, None]
rays["o"][:, 3] = 0.999 * dist
d_r = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
d_o = torch.empty(n, dtype=torch.int32, device="cuda")
out = {"scene": kind, "tris": int(idx.shape[0]), "depth": sc.tree_depth()}
ref = None
for quad in (0, 1):
    ctx.set_option("extend.quad", quad)
    sc.trace_any_device(d_r.data_ptr(), n, d_o.data_ptr()); ctx.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        sc.trace_any_device(d_r.data_ptr(), n, d_o.data_ptr()); ctx.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    got = d_o.cpu().numpy()
    if ref is None:
        ref = got
    out["any_quad%d_ms" % quad] = round(float(np.median(ts)), 3)
    out["any_quad%d_same" % quad] = bool((got == ref).all())
out["occluded_fraction"] = float(ref.mean())
ctx.set_option("extend.quad", -1)
if kind == "clutter":
    W, H = 3840, 2160
    cam = mipt.look_at_camera((-17.0, 10.5, -16.0), (3.0, 1.0, 2.0), fovy_deg=50.0, aspect=W / H)
    film = mipt.Film(ctx, W, H)
    ctx.set_kernel_timing(True)
    for quad in (0, -1):
        ctx.set_option("extend.quad", quad)
        p = mipt.render_params(W, H, sample_count=2, max_depth=4, seed=5, background=(0.5, 0.6, 0.8), flags=mipt.PT_RENDER_NEE)
        mipt.render(ctx, sc, cam, p, film, stats=True)
        st = mipt.render(ctx, sc, cam, p, film, stats=True)
        out["render_quad%s" % ("auto" if quad < 0 else "0")] = {"extend_ms": round(st.extend_ms, 2), "occlude_ms": round(st.occlude_ms, 2),
                                                                "shade_ms": round(st.shade_ms, 2), "shadow_rays": int(st.rays_shadow)}
print(json.dumps(out))
