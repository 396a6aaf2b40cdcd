import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "gpu-pathtracer_amd")); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import torch
import mipt
from oracle import oracle
from conftest import same_hits
import test_gpu_build as T
oracle.build(); oracle.lib()
ctx = mipt.Context(0)
vtx, idx = T._cone_fan()
rays = T._fan_rays(mipt, vtx, idx[:2048], 60000, seed=3)
idx = np.concatenate([idx, idx[0:2048:3], idx[0:2048:30]])
ctx.set_option("build.split_budget", 400)
for ratio in (0, 8):
    ctx.set_option("build.split_ratio", ratio)
    sc = mipt.Scene(ctx, None, vtx, idx)
    nodes = sc.read_nodes()
    got, st = sc.trace_closest(rays, stats=True)
    want = oracle.closest(nodes, vtx, idx, rays, nthreads=8)
    ex = sc.trace_closest(rays, mipt.PT_TRACE_EXACT_ORDER)
    nofix = sc.trace_closest(rays, mipt.PT_TRACE_NO_TIE_FIXUP)
    bad = same_hits(got, want)
    print("ratio", ratio, "ties", st.tie_rays, "bad", bad, "exact-vs-oracle", len(same_hits(ex, want)))
    raw = nodes.view(np.float32).reshape(-1, 8)
    payload = raw[:, 3].astype(np.int64)
    for b in bad:
        print(" ray", b, "got", got[b], "want", want[b], "nofix", nofix[b])
        for tri in (got["primid"][b], want["primid"][b]):
            same = [k for k in range(len(idx)) if (idx[k] == idx[tri]).all()]
            print("   tri", tri, "copies", same, "leaves", [np.nonzero(payload == k)[0].tolist() for k in same])
    sc.close()

# entry / exit distances of the candidate leaves of the rays that differed (last scene: ratio 8)
ctx.set_option("build.split_ratio", 8)
sc = mipt.Scene(ctx, None, vtx, idx)
nodes = sc.read_nodes()
raw = nodes.view(np.float32).reshape(-1, 8)
payload = raw[:, 3].astype(np.int64)
f = np.float32
for b in (28986, 48714, 59525):
    o = rays["o"][b, :3].astype(f); d = rays["d"][b, :3].astype(f)
    inv = (f(1.0) / d).astype(f)
    tri = int(want["primid"][b]) if False else None
    g = sc.trace_closest(rays[b:b + 1])[0]
    w = oracle.closest(nodes, vtx, idx, rays[b:b + 1])[0]
    t = g["uvwt"][3]
    print("ray", b, "o", o, "d", d, "t", repr(float(t)), "got", g["primid"], "want", w["primid"], "bound", repr(float(f(t) + f(t) * f(2.0 ** -18))))
    copies = [k for k in range(len(idx)) if (idx[k] == idx[g["primid"]]).all()]
    for k in copies:
        for n in np.nonzero(payload == k)[0]:
            lo = raw[n, 0:3]; hi = raw[n, 4:7]
            fa = ((hi - o).astype(f) * inv).astype(f); ne = ((lo - o).astype(f) * inv).astype(f)
            t1 = np.minimum(fa, ne).max(); t1 = np.maximum(fa, ne).min(); t0 = max(np.minimum(fa, ne).max(), f(0))
            if t0 <= t1 * f(1.001) and t0 < 1e3:
                print("   tri", k, "leaf", n, "t0", repr(float(t0)), "t1", repr(float(t1)), "t0-t", float(t0) - float(t))
