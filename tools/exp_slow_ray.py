"""Finds the rays that hold a launch up: times sub-ranges of the HBM-regime ray set and counts the
slab / triangle tests of the slowest single rays."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "gpu-pathtracer_amd"))
import numpy as np
import torch
import mipt
from mipt import scenes

ctx = mipt.Context(0)
ctx.set_kernel_timing(True)
for a in sys.argv[1:]:
    if "=" in a:
        k_, _, v_ = a.partition("=")
        ctx.set_option(k_, int(v_))
gen = scenes.displaced_cube_sphere if os.environ.get("SCENE", "uv") == "cube" else scenes.displaced_sphere
vtx, idx, _, _ = gen(int(os.environ.get("TRIS", "28000000")), seed=7)
ctx.set_option("build.optimize_passes", 8)
sc = mipt.Scene(ctx, None, vtx, idx, link_encoding=mipt.PT_LINKS_INT_BITS)
N = 1 << int(os.environ.get('LGN', '16'))
rays = scenes.surface_rays(vtx, idx, 1 << int(os.environ.get('RAYSET', '24')), seed=7)[:N]
d_r = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
d_h = torch.empty_like(d_r)
ctx.set_option("trace.sort_rays", 0)

def ms(lo, n):
    assert 0 <= lo and lo + n <= N
    p = d_r.data_ptr() + 32 * lo
    sc.trace_closest_device(p, n, d_h.data_ptr()); ctx.synchronize()
    sc.trace_closest_device(p, n, d_h.data_ptr())
    return ctx.last_trace_ms()

times = [(ms(lo, 1024), lo) for lo in range(0, N, 1024)]
med = float(np.median([t for t, _ in times]))
THRESH = float(os.environ.get("THRESH", "0")) or 2.0 * med      # ms: what counts as slow
print(json.dumps({"ranges_of_1024": len(times), "median_ms": round(med, 4),
                  "max_ms": round(max(times)[0], 4), "threshold_ms": round(THRESH, 4)}), flush=True)
slow = sorted([(lo, t) for t, lo in times if t > THRESH], key=lambda x: -x[1])
print("slow 1024-ray ranges:", slow[:20], flush=True)
found = []
for lo, _ in slow[:6]:
    for l2 in range(lo, lo + 1024, 64):
        if ms(l2, 64) > THRESH:
            for r in range(l2, l2 + 64):
                t = ms(r, 1)
                if t > THRESH:
                    st = sc.trace_closest_device_counted(d_r.data_ptr() + 32 * r, 1, d_h.data_ptr())
                    hit = d_h[0].cpu().numpy()
                    found.append(r)
                    print(json.dumps({"ray": r, "ms": round(t, 3), "box_tests": int(st.box_tests), "tri_tests": int(st.tri_tests),
                                      "ties": int(st.tie_rays), "o": rays["o"][r].tolist(), "d": rays["d"][r].tolist(),
                                      "hit_primid": int(hit.view(np.int32)[1]), "t": float(hit[7])}), flush=True)
# a typical ray for comparison
for r in (0, 1, 2):
    st = sc.trace_closest_device_counted(d_r.data_ptr() + 32 * r, 1, d_h.data_ptr())
    print(json.dumps({"ray": r, "ms": round(ms(r, 1), 3), "box_tests": int(st.box_tests), "tri_tests": int(st.tri_tests)}), flush=True)
print("--- launches starting at ray 8192", flush=True)
for n in (1024, 2048, 4096, 8192, 16384, 32768):
    if 8192 + n <= N:                               # never read past the rays uploaded
        print(json.dumps({"start": 8192, "rays": n, "ms": round(ms(8192, n), 3)}), flush=True)
print("--- the same launch sizes starting at ray 16384 (no slow ray before 60416)", flush=True)
for n in (1024, 4096, 16384, 32768):
    if 16384 + n <= N:
        print(json.dumps({"start": 16384, "rays": n, "ms": round(ms(16384, n), 3)}), flush=True)
if found:
    one = rays[found[0]:found[0] + 1]
    for copies in (64, 1024, 16384):
        rep = np.repeat(one, copies)
        d_c = torch.from_numpy(rep.view(np.float32).reshape(-1, 8)).cuda()
        d_hc = torch.empty_like(d_c)
        sc.trace_closest_device(d_c.data_ptr(), copies, d_hc.data_ptr()); ctx.synchronize()
        sc.trace_closest_device(d_c.data_ptr(), copies, d_hc.data_ptr())
        print(json.dumps({"copies_of_slow_ray": copies, "ms": round(ctx.last_trace_ms(), 3)}), flush=True)
