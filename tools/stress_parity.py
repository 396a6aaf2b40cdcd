"""Bit-exact comparison with the CPU restatement at sizes beyond the test suite's: millions of
rays on (a) a 262 K-triangle scene with a tenth of its triangles duplicated (every hit there is
a tie re-walked in batches) and (b) a GPU-built, GPU-optimised tree of a million triangles
(deep enough for stacks to spill), and (c) whole path-traced frames of the benchmark scene.  Prints one JSON line per case."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "gpu-pathtracer_amd"), REPO):
    sys.path.insert(0, p)
import numpy as np
import mipt
from mipt import scenes
from oracle import oracle

oracle.build()
ctx = mipt.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
OFF = int(os.environ.get("STRESS_SEED", "0"))      # shifts the seeds of every ray set
# STRESS_OPTIONS="extend.quad=1 extend.stack_entries=12": context options for the whole run, e.g.
# to put every case through the four-child walk that only large scenes get by default
for _kv in os.environ.get("STRESS_OPTIONS", "").split():
    _k, _, _v = _kv.partition("=")
    ctx.set_option(_k, int(_v))


def compare(label, sc, nodes, vtx, idx, rays, int_bits):
    got, st = sc.trace_closest(rays, stats=True)
    t0 = time.time()
    want = oracle.closest(nodes, vtx, idx, rays, nthreads=16, int_bits=int_bits)
    same = (got["primid"] == want["primid"]) & (got["shapeid"] == want["shapeid"]) & \
           (got["uvwt"].view(np.uint32) == want["uvwt"].view(np.uint32)).all(axis=1)
    for i in np.nonzero(~same)[:1][0][:4]:
        exact = sc.trace_closest(rays[i:i + 1], mipt.PT_TRACE_EXACT_ORDER)
        print("differs: ray", int(i), "o", rays["o"][i].tolist(), "d", rays["d"][i].tolist(),
              "\n  ordered kernel", got[i], "\n  restatement   ", want[i], "\n  fixed-order kernel", exact[0],
              "\n  t bits", hex(int(got["uvwt"][i, 3].view(np.uint32))), hex(int(want["uvwt"][i, 3].view(np.uint32))),
              flush=True)
    print(json.dumps({"case": label, "rays": len(rays), "differing": int((~same).sum()),
                      "re_resolved": int(st.tie_rays), "hit_fraction": round(float((got["primid"] >= 0).mean()), 3),
                      "oracle_s": round(time.time() - t0, 1)}), flush=True)
    return int((~same).sum())


bad = 0
vtx, idx, _, _ = scenes.atrium(262144, seed=42)
rng = np.random.default_rng(3)
idx2 = np.concatenate([idx, idx[rng.choice(idx.shape[0], idx.shape[0] // 10, replace=False)]])
nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx2), mipt.bvh_options(**{"bvh.builder": "sah"}))
rays = scenes.surface_rays(vtx, idx, n, seed=5 + OFF).astype(mipt.ray_dtype)
sc = mipt.Scene(ctx, nodes, vtx, idx2)
bad += compare("atrium, 10 % duplicated triangles, host SAH tree", sc, nodes, vtx, idx2, rays, False)
sc.close()

vtx, idx, _, _ = scenes.clutter(1_000_000, seed=11)
rays = scenes.surface_rays(vtx, idx, n, seed=9 + OFF).astype(mipt.ray_dtype)
for radius, passes in ((0, 16), (8, 4)):
    ctx.set_option("build.ploc_radius", radius)
    ctx.set_option("build.optimize_passes", passes)
    sc = mipt.Scene(ctx, None, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS)
    nodes = sc.read_nodes()
    bad += compare(f"clutter 1 M, GPU tree (radius {radius}, {passes} passes)", sc, nodes, vtx, idx, rays, True)
    sc.close()
# duplicated triangles in a deep GPU-optimised tree, awkward rays: ties, spills and re-walks together
idxd = np.concatenate([idx, idx[np.random.default_rng(4).choice(idx.shape[0], idx.shape[0] // 20, replace=False)]])
raysd = scenes.awkward_rays(vtx, idxd, min(n, 1 << 20), seed=55 + OFF).astype(mipt.ray_dtype)
for radius, passes in ((0, 16), (8, 0)):
    ctx.set_option("build.ploc_radius", radius)
    ctx.set_option("build.optimize_passes", passes)
    sc = mipt.Scene(ctx, None, vtx, idxd, link_encoding=mipt.PT_LINKS_INT_BITS)
    nodes = sc.read_nodes()
    bad += compare(f"clutter 1 M + 5 % duplicates, GPU tree (radius {radius}, {passes} passes), awkward rays", sc,
                   nodes, vtx, idxd, raysd, True)
    sc.close()
ctx.set_option("build.ploc_radius", 0)
ctx.set_option("build.optimize_passes", 0)

# (a4) triangles in several leaves and ties together: a latitude / longitude surface (two pole fans
# of slivers, cut into up to 64 leaves each by the GPU builder) with a twentieth of its triangles
# and every tenth sliver of the fans twice; rays at the surface and rays aimed at the fans.  The
# same mesh through the host builder with spatial splits.
vtx, idx, _, _ = scenes.displaced_sphere(1_000_000, seed=7)
fan = np.nonzero(np.abs(vtx[idx][:, :, 1]).max(axis=1) > 0.999 * np.abs(vtx[:, 1]).max())[0]
idxf = np.concatenate([idx, idx[np.random.default_rng(6).choice(idx.shape[0], idx.shape[0] // 20, replace=False)],
                       idx[fan[::10]]])
m = min(n, 1 << 21)
raysf = scenes.surface_rays(vtx, idx, m, seed=17 + OFF).astype(mipt.ray_dtype)
aim = scenes.surface_rays(vtx, idx[fan], m // 4, seed=18 + OFF).astype(mipt.ray_dtype)
aim["o"][:, :3] -= 0.5 * aim["d"][:, :3]            # start off the surface, towards and through the fans
aim["d"][:, :3] *= np.where(np.random.default_rng(19 + OFF).random((len(aim), 1)) < 0.5, 1, -1).astype(np.float32)
raysf = np.concatenate([raysf, aim])
for passes in (0, 8):
    ctx.set_option("build.optimize_passes", passes)
    sc = mipt.Scene(ctx, None, vtx, idxf)
    nodes = sc.read_nodes()
    leaves = (len(nodes) + 1) // 2
    bad += compare(f"sphere 1 M with pole fans + duplicates, GPU tree with split leaves ({leaves} leaves, {passes} passes)",
                   sc, nodes, vtx, idxf, raysf, False)
    ra = raysf.copy()
    ra["o"][:, 3] = np.random.default_rng(8).uniform(0.01, 1.0, len(ra)).astype(np.float32)
    d_any = int((sc.trace_any(ra) != oracle.any_hit(nodes, vtx, idxf, ra, nthreads=16)).sum())
    print(json.dumps({"case": "  the same tree, occlusion", "rays": len(ra), "differing": d_any}), flush=True)
    bad += d_any
    sc.close()
ctx.set_option("build.optimize_passes", 0)
nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idxf), mipt.bvh_options(**{"bvh.builder": "sah", "bvh.sah.use_splits": "1",
                                                                          "bvh.sah.min_overlap": "0.0001"}))
sc = mipt.Scene(ctx, nodes, vtx, idxf)
bad += compare(f"the same mesh, host SAH tree with spatial splits ({(len(nodes) + 1) // 2} leaves)", sc, nodes, vtx, idxf,
               raysf, False)
sc.close()

# (b1) optional: a 9 M-triangle tree (18 M nodes, integer links) built and optimised on the GPU:
# structure checked on the host, a sample of rays against the restatement walking that tree
if len(sys.argv) > 2 and sys.argv[2] == "big":
    vtx, idx, _, _ = scenes.clutter(9_000_000, seed=11)
    rays = scenes.surface_rays(vtx, idx, 1 << 20, seed=13 + OFF).astype(mipt.ray_dtype)
    for radius, passes in ((0, 8), (8, 2)):
        ctx.set_option("build.ploc_radius", radius)
        ctx.set_option("build.optimize_passes", passes)
        sc = mipt.Scene(ctx, None, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS)
        nodes = sc.read_nodes()
        mipt.validate_bvh(nodes, idx.shape[0], mipt.PT_LINKS_INT_BITS)
        bad += compare(f"clutter 9 M, GPU tree (radius {radius}, {passes} passes), structure valid", sc, nodes,
                       vtx, idx, rays, True)
        sc.close()
    ctx.set_option("build.ploc_radius", 0)
    ctx.set_option("build.optimize_passes", 0)

# (a2) rays made to be awkward on the benchmark scene (axis-aligned walls): along the axes through
# vertices and edge midpoints, from vertex to vertex (in the planes of the walls, through shared
# edges), starting on a surface, starting very far away, with a reach of almost nothing
vtx, idx, _, _ = scenes.atrium(262144, seed=42)
nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx), mipt.bvh_options(**{"bvh.builder": "sah"}))
m = min(n, 1 << 21)
rays = scenes.awkward_rays(vtx, idx, m, seed=21 + OFF).astype(mipt.ray_dtype)
sc = mipt.Scene(ctx, nodes, vtx, idx)
bad += compare("atrium, awkward rays (axes through vertices, vertex to vertex, on surfaces, far away)", sc, nodes,
               vtx, idx, rays, False)
got = sc.trace_any(rays)
d_any = int((got != oracle.any_hit(nodes, vtx, idx, rays, nthreads=16)).sum())
print(json.dumps({"case": "the same rays, occlusion", "rays": m, "differing": d_any}), flush=True)
bad += d_any
# the Moller-Trumbore closest-hit variant of the shader (bvh.glslh:181-202), fixed order only
got = sc.trace_closest(rays, mipt.PT_TRACE_MOLLER_TRUMBORE)
want = oracle.closest(nodes, vtx, idx, rays, nthreads=16, moller_trumbore=True)
nan_ok = np.isnan(got["uvwt"]) & np.isnan(want["uvwt"])
d_mt = int(((got["primid"] != want["primid"]) |
            ((got["uvwt"].view(np.uint32) != want["uvwt"].view(np.uint32)) & ~nan_ok).any(axis=1)).sum())
print(json.dumps({"case": "the same rays, Moller-Trumbore closest hit", "rays": m, "differing": d_mt}), flush=True)
bad += d_mt
sc.close()

# (a3) random triangle soups, several seeds and sizes, host builders in turn
for k, (tris, builder) in enumerate(((3000, "sah"), (50000, "median"), (200000, "sah"), (20000, "sbvh"))):
    vtx, idx, _, _ = scenes.random_soup(tris, seed=100 + k)
    try:
        nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx), mipt.bvh_options(**{"bvh.builder": builder}))
    except mipt.MiptError as err:
        print(json.dumps({"case": f"soup {tris} {builder}", "skipped": str(err)}), flush=True)
        continue
    rng = np.random.default_rng(200 + k)
    rays = np.zeros(min(n, 1 << 20), dtype=mipt.ray_dtype)
    rays["o"][:, :3] = rng.uniform(-1.5, 1.5, (len(rays), 3))
    dd = rng.normal(size=(len(rays), 3))
    rays["d"][:, :3] = dd / np.linalg.norm(dd, axis=1, keepdims=True)
    rays["o"][:, 3] = np.where(rng.random(len(rays)) < 0.3, rng.uniform(0.01, 2.0, len(rays)), 1e30)
    sc = mipt.Scene(ctx, nodes, vtx, idx)
    bad += compare(f"random soup, {tris} triangles, {builder} tree", sc, nodes, vtx, idx, rays, False)
    d_any = int((sc.trace_any(rays) != oracle.any_hit(nodes, vtx, idx, rays, nthreads=16)).sum())
    if d_any:
        print(json.dumps({"case": "  occlusion differs", "differing": d_any}), flush=True)
    bad += d_any
    sc.close()

# (b2) occlusion queries on the tie-heavy scene (finite reach, so that some are blocked)
vtx, idx, _, _ = scenes.atrium(262144, seed=42)
nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx), mipt.bvh_options(**{"bvh.builder": "sah"}))
rays = scenes.surface_rays(vtx, idx, n, seed=6 + OFF).astype(mipt.ray_dtype)
rays["o"][:, 3] = np.random.default_rng(8).uniform(0.05, 6.0, n).astype(np.float32)
sc = mipt.Scene(ctx, nodes, vtx, idx)
got = sc.trace_any(rays)
t0 = time.time()
want = oracle.any_hit(nodes, vtx, idx, rays, nthreads=16)
d = int((got != want).sum())
print(json.dumps({"case": "atrium, occlusion queries", "rays": n, "differing": d,
                  "occluded_fraction": round(float(got.mean()), 3), "oracle_s": round(time.time() - t0, 1)}), flush=True)
bad += d
sc.close()

# (b3) a two-level scene: 400 placements (rotated, scaled, some mirrored, some coincident) of
# two meshes of 20 000 triangles
sys.path.insert(0, os.path.join(REPO, "tests"))
import two_level_util as tl
rng = np.random.default_rng(12)
meshes = [tl.blob_mesh(20000, 1), tl.blob_mesh(20000, 2)]
tr, mesh_of = [], []
for k in range(400):
    lin = tl._rot(rng.normal(size=3), rng.uniform(0, 6.28)) @ np.diag(rng.uniform(0.5, 1.5, size=3))
    if k % 5 == 4:
        lin = lin @ np.diag([-1.0, 1.0, 1.0])
    tr.append(tl._affine(lin, rng.uniform(-30, 30, size=3) * (1, 0.3, 1)))
    mesh_of.append(k % 2)
tr += [tr[3].copy(), tr[10].copy()]
mesh_of += [mesh_of[3], mesh_of[10]]
tr = np.stack(tr)
opt = mipt.bvh_options(**{"bvh.builder": "sah"})
built = [(mipt.build_bvh(mipt.mesh_leaf_bounds(v, i), opt), v, i, m) for v, i, m in meshes]
top = mipt.build_bvh(mipt.instance_bounds(built, tr, mesh_of), opt)
t = oracle.TwoLevel([b[:3] for b in built], tr, mesh_of, top)
sc = mipt.Scene.instanced(ctx, built, tr, mesh_of, top, materials=tl.materials(mipt))
m = min(n, 1 << 22)
rays = np.zeros(m, dtype=mipt.ray_dtype)
rays["o"][:, :3] = rng.uniform(-40, 40, size=(m, 3)).astype(np.float32) * (1, 0.4, 1)
rays["o"][:, 3] = 1e30
dd = rng.uniform(-30, 30, size=(m, 3)) * (1, 0.3, 1) - rays["o"][:, :3]
rays["d"][:, :3] = (dd / np.linalg.norm(dd, axis=1, keepdims=True)).astype(np.float32)
got, st = sc.trace_closest(rays, stats=True)
t0 = time.time()
want = t.closest(rays, nthreads=16)
same = (got["primid"] == want["primid"]) & (got["shapeid"] == want["shapeid"]) & \
       (got["uvwt"].view(np.uint32) == want["uvwt"].view(np.uint32)).all(axis=1)
d = int((~same).sum())
d2 = int((sc.trace_any(rays) != t.any_hit(rays, nthreads=16)).sum())
print(json.dumps({"case": "two-level, 402 instances of 20 000-triangle meshes", "rays": m, "differing": d,
                  "differing_occlusion": d2, "re_resolved": int(st.tie_rays),
                  "hit_fraction": round(float((got["primid"] >= 0).mean()), 3),
                  "oracle_s": round(time.time() - t0, 1)}), flush=True)
bad += d + d2
# the same scene with awkward rays: aimed at the (approximate) world positions of vertices
wv, wi, off = [], [], 0
for k in range(0, tr.shape[0], 7):
    v, i, _ = meshes[mesh_of[k]]
    wv.append((v.astype(np.float64) @ tr[k][:, :3].astype(np.float64).T + tr[k][:, 3]).astype(np.float32))
    wi.append(i + off)
    off += v.shape[0]
rays = scenes.awkward_rays(np.concatenate(wv), np.concatenate(wi), min(n, 1 << 20), seed=33 + OFF).astype(mipt.ray_dtype)
got, st = sc.trace_closest(rays, stats=True)
want = t.closest(rays, nthreads=16)
same = (got["primid"] == want["primid"]) & (got["shapeid"] == want["shapeid"]) & \
       ((got["uvwt"].view(np.uint32) == want["uvwt"].view(np.uint32)) | (np.isnan(got["uvwt"]) & np.isnan(want["uvwt"]))).all(axis=1)
d = int((~same).sum())
for i in np.nonzero(~same)[0][:3]:
    print("differs: ray", int(i), rays[i], "\n  gpu", got[i], "\n  cpu", want[i], flush=True)
d2 = int((sc.trace_any(rays) != t.any_hit(rays, nthreads=16)).sum())
print(json.dumps({"case": "two-level, awkward rays", "rays": len(rays), "differing": d, "differing_occlusion": d2,
                  "re_resolved": int(st.tie_rays), "hit_fraction": round(float((got["primid"] >= 0).mean()), 3)}), flush=True)
bad += d + d2
sc.close()
# a two-level frame with lamps, glossy material and next-event estimation (the scene of the tests, larger)
tsc = tl.scene(n_blob_tris=4000, n_blobs=40, seed=9)
tbuilt = [(mipt.build_bvh(mipt.mesh_leaf_bounds(v, i), opt), v, i, mm) for v, i, mm in tsc["meshes"]]
ttop = mipt.build_bvh(mipt.instance_bounds(tbuilt, tsc["transforms"], tsc["mesh_of"]), opt)
tt = oracle.TwoLevel([b[:3] for b in tbuilt], tsc["transforms"], tsc["mesh_of"], ttop)
sc = mipt.Scene.instanced(ctx, tbuilt, tsc["transforms"], tsc["mesh_of"], ttop, materials=tl.materials(mipt),
                          material_offsets=tsc["material_offsets"])
Wt, Ht = 640, 360
camt = mipt.look_at_camera((0.0, 3.0, -14.0), (0.0, 0.5, 0.0), fovy_deg=45.0, aspect=Wt / Ht)
filmt = mipt.Film(ctx, Wt, Ht)
for flags in (0, mipt.PT_RENDER_NEE):
    filmt.clear()
    pt_ = mipt.render_params(Wt, Ht, sample_count=4, max_depth=6, seed=5, flags=flags, background=(0.05, 0.06, 0.08))
    stt = mipt.render(ctx, sc, camt, pt_, filmt, stats=True)
    wantt, _ = tt.render(tl.materials(oracle), tl.flat_material_ids(tsc), camt, pt_, nthreads=16)
    dpx = int((filmt.read().view(np.uint32) != wantt.view(np.uint32)).any(axis=-1).sum())
    print(json.dumps({"case": "two-level frame 640x360, 4 spp, depth 6" + (", next-event estimation" if flags else ""),
                      "rays": int(stt.rays_primary + stt.rays_bounce + stt.rays_shadow), "differing_pixels": dpx}), flush=True)
    bad += dpx
filmt.close()
sc.close()

# (b5) the reference's camera (depth.frag:47-83) at 3840x2160
camr = mipt.look_at_camera((0.0, 0.0, -10.0), (0, 0, 0), fovy_deg=60.0, aspect=16 / 9, mode=mipt.PT_CAMERA_REFERENCE)
gr = ctx.raygen(camr, 3840, 2160)
ocam = oracle.camera_from(tuple(camr.eye), tuple(camr.c0), tuple(camr.c1), tuple(camr.c2), camr.sx, camr.sy)
dr = int((gr.view(np.uint32) != oracle.raygen_reference(ocam, 3840, 2160).view(np.uint32)).any(axis=-1).sum()) \
    if hasattr(oracle, "raygen_reference") else -1
print(json.dumps({"case": "reference camera rays, 3840x2160", "rays": 3840 * 2160, "differing": dr}), flush=True)
bad += max(dr, 0)

# (b6) degenerate meshes through the GPU builders: all triangles in one place, zero-area
# triangles, a planar grid (flat boxes), coordinates spanning twelve orders of magnitude
rng = np.random.default_rng(77)
def _mesh(v):
    v = np.asarray(v, dtype=np.float32).reshape(-1, 3)
    return v, np.arange(v.shape[0], dtype=np.int32).reshape(-1, 3)
one = np.tile(np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0]]), (5000, 1))
pts = rng.uniform(-1, 1, (30000, 1, 3)).astype(np.float32)
zero_area = np.concatenate([np.repeat(pts, 3, axis=1).reshape(-1, 3),                       # points
                            rng.uniform(-1, 1, (30000, 3, 3)).astype(np.float32).reshape(-1, 3)])
gx, gy = np.meshgrid(np.arange(200, dtype=np.float32), np.arange(200, dtype=np.float32))
q = np.stack([gx, gy, 0 * gx], -1).reshape(-1, 1, 3)
plane = np.concatenate([q, q + np.float32([1, 0, 0]), q + np.float32([0, 1, 0])], 1).reshape(-1, 3)
scale = (10.0 ** rng.uniform(-6, 6, (40000, 1, 1))).astype(np.float32)
wide = (rng.uniform(-1, 1, (40000, 3, 3)).astype(np.float32) * scale).reshape(-1, 3)
for label, verts in (("5 000 copies of one triangle", one), ("points and triangles", zero_area),
                     ("planar grid", plane), ("twelve orders of magnitude", wide)):
    v, i = _mesh(verts)
    r = scenes.awkward_rays(v, i, min(n, 1 << 18), seed=5 + OFF).astype(mipt.ray_dtype)
    for radius, passes in ((0, 0), (0, 6), (8, 2)):
        ctx.set_option("build.ploc_radius", radius)
        ctx.set_option("build.optimize_passes", passes)
        sc = mipt.Scene(ctx, None, v, i)           # float-valued links: the occlusion restatement reads no others
        nd = sc.read_nodes()
        mipt.validate_bvh(nd, i.shape[0], mipt.PT_LINKS_FLOAT_VALUE)
        got = sc.trace_closest(r)
        want = oracle.closest(nd, v, i, r, nthreads=16)
        nan_ok = np.isnan(got["uvwt"]) & np.isnan(want["uvwt"])
        dd = int(((got["primid"] != want["primid"]) |
                  ((got["uvwt"].view(np.uint32) != want["uvwt"].view(np.uint32)) & ~nan_ok).any(axis=1)).sum())
        da = int((sc.trace_any(r) != oracle.any_hit(nd, v, i, r, nthreads=16)).sum())
        print(json.dumps({"case": f"GPU build of {label} (radius {radius}, {passes} passes)", "rays": len(r),
                          "differing": dd, "differing_occlusion": da,
                          "hit_fraction": round(float((got["primid"] >= 0).mean()), 3)}), flush=True)
        bad += dd + da
        sc.close()
ctx.set_option("build.ploc_radius", 0)
ctx.set_option("build.optimize_passes", 0)

# (b4) the Cornell box (32 triangles, every surface axis-aligned): awkward rays, and a whole frame
vtx, idx, mats, mids = scenes.cornell_box()
nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx), mipt.bvh_options(**{"bvh.builder": "sah"}))
sc = mipt.Scene(ctx, nodes, vtx, idx, mats, mids)
rays = scenes.awkward_rays(vtx, idx, min(n, 1 << 20), seed=44 + OFF).astype(mipt.ray_dtype)
bad += compare("Cornell box, awkward rays", sc, nodes, vtx, idx, rays, False)
d_any = int((sc.trace_any(rays) != oracle.any_hit(nodes, vtx, idx, rays, nthreads=16)).sum())
Wc, Hc = 960, 540
camc = mipt.look_at_camera((278.0, 273.0, -800.0), (278.0, 273.0, 0.0), fovy_deg=39.3, aspect=Wc / Hc)
filmc = mipt.Film(ctx, Wc, Hc)
pc = mipt.render_params(Wc, Hc, sample_count=8, max_depth=6, seed=3, flags=mipt.PT_RENDER_NEE)
mipt.render(ctx, sc, camc, pc, filmc)
wantc, _ = oracle.render(nodes, vtx, idx, mats, mids, camc, pc, nthreads=16)
d_film = int((filmc.read().view(np.uint32) != wantc.view(np.uint32)).any(axis=-1).sum())
print(json.dumps({"case": "Cornell box: occlusion of the same rays; 960x540 frame, 8 spp, depth 6, next-event estimation",
                  "differing": d_any, "differing_pixels": d_film}), flush=True)
bad += d_any + d_film
filmc.close()
sc.close()

# (c) whole frames: the wavefront path tracer against the CPU restatement of the same step
vtx, idx, mats, mids = scenes.atrium(262144, seed=42)
nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(vtx, idx), mipt.bvh_options(**{"bvh.builder": "sah"}))
sc = mipt.Scene(ctx, nodes, vtx, idx, mats, mids)
W, H = (1920, 1080) if n > (1 << 22) else (960, 540)
cam = scenes.atrium_camera(W / H)
film = mipt.Film(ctx, W, H)
for flags, label in ((0, "atrium %dx%d, 4 spp, depth 8" % (W, H)), (mipt.PT_RENDER_NEE, "the same with next-event estimation")):
    film.clear()
    p = mipt.render_params(W, H, sample_count=4, max_depth=8, seed=42, rr_start_depth=3,
                           background=(0.55, 0.65, 0.8), flags=flags)
    st = mipt.render(ctx, sc, cam, p, film, stats=True)
    t0 = time.time()
    want, _ = oracle.render(nodes, vtx, idx, mats, mids, cam, p, nthreads=16)
    got = film.read()
    diff = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=-1).sum())
    print(json.dumps({"case": label, "rays": int(st.rays_primary + st.rays_bounce + st.rays_shadow),
                      "differing_pixels": diff, "oracle_s": round(time.time() - t0, 1)}), flush=True)
    bad += diff
film.close()
sc.close()
sys.exit(1 if bad else 0)
