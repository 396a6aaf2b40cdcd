"""Deterministic scene inputs for tests and benchmarks.

The reference ships no assets (its one scene, a Lucy .glb, is outside the
checkout: src/main_granite.cpp:265), and none can be downloaded here, so the
named workloads are seeded procedural stand-ins with matching triangle counts:

* :func:`cornell_box`   -- the measured Cornell box, loaded from an OBJ file;
* :func:`atrium`        -- Sponza stand-in: colonnaded hall, 262,144 triangles;
* :func:`clutter`       -- San-Miguel-like stand-in: shell plus dense clutter;
* :func:`displaced_sphere` -- Lucy-like stand-in: closed, finely displaced surface;
* :func:`fallback_quad` -- the two-triangle mesh at src/main.cpp:918-921;
* :func:`random_soup`   -- uniformly random triangles for parity tests.

Each returns ``(vtx float32 [V,3], idx int32 [T,3], materials, material_ids)``.
"""
from __future__ import annotations

import os

import numpy as np

from . import material_dtype

_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CORNELL_OBJ = os.path.join(_REPO, "tests", "golden", "cornell_box.obj")


def make_materials(rows):
    """rows: iterable of (kd, ks, ke, roughness)."""
    m = np.zeros(len(rows), dtype=material_dtype)
    for i, (kd, ks, ke, rough) in enumerate(rows):
        m["kd"][i] = kd
        m["ks"][i] = ks
        m["ke"][i] = ke
        m["roughness"][i] = rough
    return m


def fallback_quad():
    """The mesh the legacy app builds when no model is loaded (src/main.cpp:918-921)."""
    vtx = np.array([[0.0123, -0.4312, -1.0321], [0.3123, -0.5231, 1.0231],
                    [0.1213, 0.5132, 1.0312], [0.0321, 0.5231, -1.0123]], dtype=np.float32)
    idx = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int32)
    mats = make_materials([((0.5, 0.5, 0.5), (0, 0, 0), (0, 0, 0), 1.0)])
    return vtx, idx, mats, np.zeros(2, dtype=np.int32)


def cornell_box():
    from . import Mesh
    m = Mesh(CORNELL_OBJ)
    return m.vtx, m.idx, m.materials, m.material_ids


def random_soup(n_tri: int, seed: int = 0, extent: float = 10.0, size: float = 1.0):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, size=(n_tri, 1, 3))
    v = c + rng.uniform(-size, size, size=(n_tri, 3, 3))
    vtx = v.reshape(-1, 3).astype(np.float32)
    idx = np.arange(3 * n_tri, dtype=np.int32).reshape(-1, 3)
    mats = make_materials([((0.6, 0.6, 0.6), (0, 0, 0), (0, 0, 0), 1.0)])
    return vtx, idx, mats, np.zeros(n_tri, dtype=np.int32)


# --------------------------------------------------------------------------
class _MeshBuilder:
    def __init__(self):
        self.v, self.i, self.m = [], [], []
        self.nv = 0

    @property
    def n_tri(self):
        return sum(a.shape[0] for a in self.i)

    def add(self, vtx, idx, mat):
        vtx = np.asarray(vtx, dtype=np.float64).reshape(-1, 3)
        idx = np.asarray(idx, dtype=np.int64).reshape(-1, 3)
        self.v.append(vtx)
        self.i.append(idx + self.nv)
        self.m.append(np.full(idx.shape[0], mat, dtype=np.int32))
        self.nv += vtx.shape[0]

    def grid(self, origin, du, dv, nu, nv, mat, height=None):
        """(nu x nv) quads spanning origin + s*du + t*dv, optional height field h(s,t) along du x dv."""
        origin, du, dv = (np.asarray(a, dtype=np.float64) for a in (origin, du, dv))
        s, t = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="ij")
        p = origin + s[..., None] * du + t[..., None] * dv
        if height is not None:
            n = np.cross(du, dv)
            n /= np.linalg.norm(n)
            p = p + height(s, t)[..., None] * n
        a = (np.arange(nu)[:, None] * (nv + 1) + np.arange(nv)[None, :]).reshape(-1)
        b, c, d = a + (nv + 1), a + (nv + 1) + 1, a + 1
        self.add(p.reshape(-1, 3), np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)]), mat)

    def strip(self, origin, du, dv, cols, n_quads, mat, height=None):
        """Exactly n_quads quads laid out row-major on a lattice `cols` wide (last row partial)."""
        if n_quads <= 0:
            return
        origin, du, dv = (np.asarray(a, dtype=np.float64) for a in (origin, du, dv))
        rows = -(-n_quads // cols)
        s, t = np.meshgrid(np.linspace(0, 1, cols + 1), np.linspace(0, 1, rows + 1), indexing="ij")
        p = origin + s[..., None] * du + t[..., None] * dv
        if height is not None:
            n = np.cross(du, dv)
            n /= np.linalg.norm(n)
            p = p + height(s, t)[..., None] * n
        q = np.arange(n_quads)
        r, c = q // cols, q % cols
        a = c * (rows + 1) + r
        b, cc, d = a + (rows + 1), a + (rows + 1) + 1, a + 1
        self.add(p.reshape(-1, 3), np.concatenate([np.stack([a, b, cc], 1), np.stack([a, cc, d], 1)]), mat)

    def revolve(self, centre, axis_u, axis_v, axis_w, radius, z0, z1, n_around, n_along, mat,
                angle0=0.0, angle1=2 * np.pi):
        """Surface of revolution about axis_w with radius(z) callable; closed when the angle is 2 pi."""
        centre = np.asarray(centre, dtype=np.float64)
        u, v, w = (np.asarray(a, dtype=np.float64) for a in (axis_u, axis_v, axis_w))
        ang = np.linspace(angle0, angle1, n_around + 1)
        z = np.linspace(z0, z1, n_along + 1)
        A, Z = np.meshgrid(ang, z, indexing="ij")
        R = radius(Z) if callable(radius) else np.full_like(Z, radius)
        p = centre + (R * np.cos(A))[..., None] * u + (R * np.sin(A))[..., None] * v + Z[..., None] * w
        a = (np.arange(n_around)[:, None] * (n_along + 1) + np.arange(n_along)[None, :]).reshape(-1)
        b, c, d = a + (n_along + 1), a + (n_along + 1) + 1, a + 1
        self.add(p.reshape(-1, 3), np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)]), mat)

    def sphere(self, centre, radius, n_lon, n_lat, mat, displace=None):
        lon = np.linspace(0, 2 * np.pi, n_lon + 1)
        lat = np.linspace(0, np.pi, n_lat + 1)
        LO, LA = np.meshgrid(lon, lat, indexing="ij")
        d = np.stack([np.sin(LA) * np.cos(LO), np.cos(LA), np.sin(LA) * np.sin(LO)], -1)
        r = radius * (1.0 + (displace(d) if displace is not None else 0.0))
        p = np.asarray(centre, dtype=np.float64) + d * np.asarray(r)[..., None]
        a = (np.arange(n_lon)[:, None] * (n_lat + 1) + np.arange(n_lat)[None, :]).reshape(-1)
        b, c, e = a + (n_lat + 1), a + (n_lat + 1) + 1, a + 1
        self.add(p.reshape(-1, 3), np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, e], 1)]), mat)

    def finish(self, materials):
        vtx = np.concatenate(self.v).astype(np.float32)
        idx = np.concatenate(self.i).astype(np.int32)
        mid = np.concatenate(self.m).astype(np.int32)
        return vtx, idx, materials, mid


def _fbm(rng, octaves=4):
    """Smooth pseudo-noise on [0,1]^2 as a sum of random sinusoids (seeded, resolution independent)."""
    k = rng.uniform(1.0, 3.0, size=(octaves, 2)) * (2.0 ** np.arange(octaves))[:, None]
    ph = rng.uniform(0, 2 * np.pi, size=(octaves, 2))
    amp = 0.5 ** np.arange(octaves)

    def f(s, t):
        out = np.zeros_like(s)
        for o in range(octaves):
            out += amp[o] * np.sin(2 * np.pi * k[o, 0] * s + ph[o, 0]) * np.cos(2 * np.pi * k[o, 1] * t + ph[o, 1])
        return out
    return f


def atrium(n_tri: int = 262144, seed: int = 42):
    """Sponza stand-in: a 36 x 14 x 16 hall with two storeys of columns and arches, draped
    cloth, rough floor relief and emissive ceiling panels; exactly ``n_tri`` triangles
    (an even number >= 65536).  Materials mix Lambert and GGX lobes."""
    if n_tri % 2 or n_tri < 65536:
        raise ValueError("atrium: n_tri must be even and >= 65536")
    rng = np.random.default_rng(seed)
    rough_table = rng.uniform(0.15, 0.6, size=4)
    mats = make_materials([
        ((0.70, 0.66, 0.58), (0.00, 0.00, 0.00), (0, 0, 0), 1.0),             # 0 stone walls
        ((0.55, 0.52, 0.48), (0.04, 0.04, 0.04), (0, 0, 0), rough_table[0]),  # 1 polished floor
        ((0.62, 0.58, 0.50), (0.00, 0.00, 0.00), (0, 0, 0), 1.0),             # 2 columns
        ((0.60, 0.10, 0.08), (0.00, 0.00, 0.00), (0, 0, 0), 1.0),             # 3 red cloth
        ((0.10, 0.25, 0.55), (0.00, 0.00, 0.00), (0, 0, 0), 1.0),             # 4 blue cloth
        ((0.05, 0.05, 0.05), (0.85, 0.65, 0.35), (0, 0, 0), rough_table[1]),  # 5 brass
        ((0.20, 0.20, 0.20), (0.50, 0.50, 0.50), (0, 0, 0), rough_table[2]),  # 6 steel
        ((0.80, 0.80, 0.80), (0.00, 0.00, 0.00), (12, 11, 9), 1.0),           # 7 light panels
        ((0.30, 0.45, 0.20), (0.02, 0.02, 0.02), (0, 0, 0), rough_table[3]),  # 8 plants
    ])
    scale = max(1, int(round((n_tri / 262144.0) ** 0.5 * 8)))  # tessellation multiplier (8 at 262k)
    mb = _MeshBuilder()
    L, W, H = 36.0, 14.0, 16.0
    x0, z0 = -L / 2, -W / 2
    # shell: walls and ceiling (floor is the filler strip at the end)
    mb.grid((x0, 0, z0), (L, 0, 0), (0, H, 0), 9 * scale, 4 * scale, 0)            # wall z-
    mb.grid((x0, 0, -z0), (0, H, 0), (L, 0, 0), 4 * scale, 9 * scale, 0)           # wall z+
    mb.grid((x0, 0, z0), (0, H, 0), (0, 0, W), 4 * scale, 4 * scale, 0)            # wall x-
    mb.grid((-x0, 0, z0), (0, 0, W), (0, H, 0), 4 * scale, 4 * scale, 0)           # wall x+
    mb.grid((x0, H, z0), (0, 0, W), (L, 0, 0), 4 * scale, 9 * scale, 0)            # ceiling
    # light panels just under the ceiling
    for k in range(6):
        cx = x0 + (k + 0.5) * L / 6
        mb.grid((cx - 1.5, H - 0.05, -1.0), (3.0, 0, 0), (0, 0, 2.0), 2, 2, 7)
    # gallery slab on each side (upper storey floor)
    for side in (-1, 1):
        zc = side * (W / 2 - 1.5)
        mb.grid((x0, 6.0, zc - 1.5), (L, 0, 0), (0, 0, 3.0), 9 * scale, scale, 0)
        mb.grid((x0, 5.6, zc - 1.5), (0, 0, 3.0), (L, 0, 0), scale, 9 * scale, 0)
    # columns: two storeys, both sides
    n_col = 10
    for storey, (y0c, y1c, rad) in enumerate(((0.0, 5.6, 0.45), (6.0, 11.5, 0.32))):
        for side in (-1, 1):
            zc = side * (W / 2 - 3.0)
            for k in range(n_col):
                cx = x0 + (k + 0.5) * L / n_col
                prof = (lambda z, r=rad, a=y0c, b=y1c:
                        r * (1.0 + 0.18 * np.exp(-((z - a) / 0.35) ** 2) + 0.22 * np.exp(-((z - b) / 0.3) ** 2)
                             - 0.06 * np.sin(np.pi * (z - a) / (b - a))))
                mb.revolve((cx, 0, zc), (1, 0, 0), (0, 0, 1), (0, 1, 0), prof, y0c, y1c,
                           3 * scale, 5 * scale, 2 if (k + storey) % 4 else 5)
    # arches between neighbouring columns (half tori), lower storey
    for side in (-1, 1):
        zc = side * (W / 2 - 3.0)
        for k in range(n_col - 1):
            cx = x0 + (k + 1.0) * L / n_col
            span = L / n_col / 2 - 0.45
            tube = 0.22
            ang = np.linspace(0, np.pi, 2 * scale + 1)
            sec = np.linspace(0, 2 * np.pi, scale + 1)
            A, S = np.meshgrid(ang, sec, indexing="ij")
            px = cx + (span + tube * np.cos(S)) * np.cos(A)
            py = 4.2 + (span * 0.8 + tube * np.cos(S)) * np.sin(A)
            pz = zc + tube * np.sin(S)
            p = np.stack([px, py, pz], -1)
            nu, nv = 2 * scale, scale
            a = (np.arange(nu)[:, None] * (nv + 1) + np.arange(nv)[None, :]).reshape(-1)
            b, c, d = a + (nv + 1), a + (nv + 1) + 1, a + 1
            mb.add(p.reshape(-1, 3), np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)]), 2)
    # draped cloth hanging across the nave
    for k in range(5):
        cx = x0 + (k + 1.0) * L / 6
        noise = _fbm(rng)
        sag = (lambda s, t, f=noise: -1.6 * np.sin(np.pi * t) + 0.25 * f(s, t))
        mb.grid((cx - 1.2, 12.5, -W / 2 + 3.2), (2.4, 0, 0), (0, 0, W - 6.4), 3 * scale, 6 * scale,
                3 + (k % 2), height=(lambda s, t, g=sag: -g(s, t)))
    # clutter: vases / spheres on the floor
    for k in range(24):
        cx = rng.uniform(x0 + 2, -x0 - 2)
        cz = rng.uniform(-2.5, 2.5)
        r = rng.uniform(0.25, 0.7)
        noise = _fbm(rng, 3)
        mb.sphere((cx, r * 0.98, cz), r, 2 * scale, scale, (5, 6, 8, 2)[k % 4],
                  displace=(lambda d, f=noise: 0.08 * f(d[..., 0] * 0.5 + 0.5, d[..., 1] * 0.5 + 0.5)))
    # floor: relief strip with exactly the remaining triangle count
    remaining = n_tri - mb.n_tri
    if remaining <= 0 or remaining % 2:
        raise ValueError(f"atrium: tessellation leaves {remaining} triangles for the floor")
    n_quads = remaining // 2
    cols = max(8, int(round((n_quads * L / W) ** 0.5)))
    relief = _fbm(rng, 5)
    mb.strip((x0, 0, z0), (L, 0, 0), (0, 0, W), cols, n_quads, 1,
             height=(lambda s, t: -0.03 * relief(s, t)))
    out = mb.finish(mats)
    assert out[1].shape[0] == n_tri
    return out


def atrium_camera(aspect: float = 16.0 / 9.0):
    from . import look_at_camera
    return look_at_camera((-15.0, 3.2, 0.6), (10.0, 5.0, -0.4), fovy_deg=55.0, aspect=aspect)


def displaced_sphere(n_tri: int, seed: int = 7):
    """Lucy-like stand-in: one closed surface with fine fBm displacement, ~n_tri triangles
    (exactly 2*n_lon*n_lat with n_lon = 2*n_lat)."""
    rng = np.random.default_rng(seed)
    n_lat = max(4, int(round((n_tri / 4.0) ** 0.5)))
    n_lon = 2 * n_lat
    noise = [_fbm(rng, 6) for _ in range(2)]
    mats = make_materials([((0.15, 0.15, 0.15), (0.75, 0.75, 0.75), (0, 0, 0), 0.2)])
    mb = _MeshBuilder()

    def disp(d):
        u = np.arctan2(d[..., 2], d[..., 0]) / (2 * np.pi) + 0.5
        v = np.arccos(np.clip(d[..., 1], -1, 1)) / np.pi
        return 0.12 * noise[0](u, v) + 0.05 * noise[1](4 * u, 4 * v)
    mb.sphere((0, 0, 0), 1.0, n_lon, n_lat, 0, displace=disp)
    return mb.finish(mats)


def _fbm3(rng, octaves=6):
    """Smooth pseudo-noise on R^3 as a sum of random plane waves (seeded, resolution independent)."""
    dirs = rng.normal(size=(octaves, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    k = dirs * (rng.uniform(1.0, 3.0, size=(octaves, 1)) * (2.0 ** np.arange(octaves))[:, None])
    ph = rng.uniform(0, 2 * np.pi, size=octaves)
    amp = 0.5 ** np.arange(octaves)

    def f(d):
        out = np.zeros(d.shape[:-1])
        for o in range(octaves):
            out += amp[o] * np.sin(2 * np.pi * (d @ k[o]) + ph[o])
        return out
    return f


def displaced_cube_sphere(n_tri: int, seed: int = 7):
    """Lucy-like stand-in without the two pole fans of `displaced_sphere`: a cube's six faces,
    each an n x n grid, pushed out to the unit sphere and displaced along the radius by fBm of the
    direction; ~n_tri triangles (exactly 12 n^2) of nearly uniform size, as a scanned surface has
    them.  Faces share their edge positions, so the surface is closed."""
    rng = np.random.default_rng(seed)
    n = max(2, int(round((n_tri / 12.0) ** 0.5)))
    noise = [_fbm3(rng, 6) for _ in range(2)]
    mats = make_materials([((0.15, 0.15, 0.15), (0.75, 0.75, 0.75), (0, 0, 0), 0.2)])
    mb = _MeshBuilder()
    # tan-warped grid coordinates: equal angles instead of equal cube-face steps
    g = np.tan(np.linspace(-np.pi / 4, np.pi / 4, n + 1))
    A, B = np.meshgrid(g, g, indexing="ij")
    one = np.ones_like(A)
    a = (np.arange(n)[:, None] * (n + 1) + np.arange(n)[None, :]).reshape(-1)
    b, c, e = a + (n + 1), a + (n + 1) + 1, a + 1
    quads_out = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, e], 1)])
    quads_in = quads_out[:, ::-1]
    for axis in range(3):
        for sign in (1.0, -1.0):
            comp = [None, None, None]
            comp[axis] = sign * one
            comp[(axis + 1) % 3] = A
            comp[(axis + 2) % 3] = B
            d = np.stack(comp, -1)
            d /= np.linalg.norm(d, axis=-1, keepdims=True)
            # wavelengths as on `displaced_sphere`, whose noise runs once round the equator per unit
            r = 1.0 + 0.12 * noise[0](d / 4.7) + 0.05 * noise[1](4.0 * d / 4.7)
            mb.add((d * r[..., None]).reshape(-1, 3), quads_out if sign > 0 else quads_in, 0)
    return mb.finish(mats)


def clutter(n_tri: int, seed: int = 11):
    """San-Miguel-like stand-in: a courtyard shell with many small displaced blobs ("foliage")
    flattened into one mesh; ~n_tri triangles."""
    rng = np.random.default_rng(seed)
    mats = make_materials([
        ((0.65, 0.6, 0.5), (0, 0, 0), (0, 0, 0), 1.0),
        ((0.2, 0.5, 0.15), (0.02, 0.02, 0.02), (0, 0, 0), 0.5),
        ((0.5, 0.3, 0.2), (0, 0, 0), (0, 0, 0), 1.0),
        ((0.8, 0.8, 0.8), (0, 0, 0), (10, 10, 9), 1.0),
    ])
    mb = _MeshBuilder()
    S = 40.0
    shell_res = max(8, int((n_tri * 0.05 / 10) ** 0.5))
    mb.grid((-S / 2, 0, -S / 2), (0, 0, S), (S, 0, 0), shell_res, shell_res, 0,
            height=(lambda s, t, f=_fbm(rng): 0.1 * f(s, t)))
    for a, b in (((-S / 2, 0, -S / 2), (S, 0, 0)), ((-S / 2, 0, S / 2), (0, 0, -S)),
                 ((S / 2, 0, S / 2), (-S, 0, 0)), ((S / 2, 0, -S / 2), (0, 0, S))):
        mb.grid(a, b, (0, 12, 0), shell_res, shell_res // 2, 0)
    mb.grid((-6, 11.9, -6), (12, 0, 0), (0, 0, 12), 4, 4, 3)
    n_blobs = min(max(8, n_tri // 4096), 240)      # more triangles = finer blobs, not a denser cloud
    per = max(64, (n_tri - mb.n_tri) // n_blobs)
    n_lat = max(4, int(round((per / 4.0) ** 0.5)))
    for k in range(n_blobs):
        c = (rng.uniform(-S / 2 + 2, S / 2 - 2), rng.uniform(0.3, 7.0), rng.uniform(-S / 2 + 2, S / 2 - 2))
        r = rng.uniform(0.3, 1.4)
        noise = _fbm(rng, 4)
        mb.sphere(c, r, 2 * n_lat, n_lat, 1 + (k % 2),
                  displace=(lambda d, f=noise: 0.3 * f(d[..., 0] * 0.5 + 0.5, d[..., 2] * 0.5 + 0.5)))
    return mb.finish(mats)


def surface_rays(vtx, idx, n_rays: int, seed: int = 7, tmax: float = 1e30):
    """Incoherent secondary rays: origins on random triangles (offset along the normal),
    cosine-distributed directions about it.  Returns a ray_dtype array."""
    from . import ray_dtype
    rng = np.random.default_rng(seed)
    va, vb, vc = (vtx[idx[:, k]].astype(np.float64) for k in range(3))
    valid = np.nonzero(np.linalg.norm(np.cross(vb - va, vc - va), axis=1) > 1e-12)[0]
    t = valid[rng.integers(0, valid.shape[0], size=n_rays)]   # skip zero-area triangles
    a, b, c = va[t], vb[t], vc[t]
    u, v = rng.random(n_rays), rng.random(n_rays)
    flip = u + v > 1
    u[flip], v[flip] = 1 - u[flip], 1 - v[flip]
    p = a + (b - a) * u[:, None] + (c - a) * v[:, None]
    n = np.cross(b - a, c - a)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    n = n / np.maximum(ln, 1e-30)
    n *= np.where(rng.random(n_rays) < 0.5, 1.0, -1.0)[:, None]
    r1, r2 = rng.random(n_rays), rng.random(n_rays)
    phi = 2 * np.pi * r1
    st, ct = np.sqrt(r2), np.sqrt(1 - r2)
    helper = np.where(np.abs(n[:, :1]) > 0.9, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
    tx = np.cross(helper, n)
    tx /= np.linalg.norm(tx, axis=1, keepdims=True)
    ty = np.cross(n, tx)
    d = tx * (st * np.cos(phi))[:, None] + ty * (st * np.sin(phi))[:, None] + n * ct[:, None]
    rays = np.zeros(n_rays, dtype=ray_dtype)
    rays["o"][:, :3] = (p + n * 1e-3).astype(np.float32)
    rays["o"][:, 3] = tmax
    rays["d"][:, :3] = d.astype(np.float32)
    return rays


def awkward_rays(vtx, idx, m, seed=0):
    """Rays made to be awkward for a scene of axis-aligned walls: along the axes through vertices
    and edge midpoints, from vertex to vertex (in the planes of the walls, through shared edges),
    starting on a surface, starting very far away; a fifth with a short reach, often exactly the
    distance to the surface aimed at."""
    ray_dtype = np.dtype([("o", np.float32, 4), ("d", np.float32, 4)])
    rng = np.random.default_rng(seed)
    tri = idx[rng.integers(0, idx.shape[0], m)]
    corner = vtx[tri[:, 0]]
    mid = (vtx[tri[:, 0]] + vtx[tri[:, 1]]) * np.float32(0.5)
    rays = np.zeros(m, dtype=ray_dtype)
    kind = rng.integers(0, 6, m)
    axis = rng.integers(0, 3, m)
    sign = rng.choice(np.float32([-1, 1]), m)
    e = np.zeros((m, 3), dtype=np.float32)
    e[np.arange(m), axis] = sign
    target = np.where((kind % 2 == 0)[:, None], corner, mid)
    dist = rng.choice(np.float32([0.5, 3.0, 40.0, 1e6]), m)
    o = target - e * dist[:, None]                          # kinds 0,1: along an axis through a vertex / an edge midpoint
    d = e.copy()
    other = vtx[idx[rng.integers(0, idx.shape[0], m), 1]]
    v2v = (kind == 2) | (kind == 3)                         # kinds 2,3: from one vertex (or just off it) to another
    o[v2v] = corner[v2v] + np.where((kind[v2v] == 3)[:, None], np.float32(1e-3) * e[v2v], 0)
    dd = other[v2v] - o[v2v]
    dd[np.abs(dd).sum(axis=1) == 0] = (1, 0, 0)
    d[v2v] = dd / np.linalg.norm(dd, axis=1, keepdims=True)
    onsurf = kind >= 4                                      # kinds 4,5: from a point of a surface, random direction
    w = rng.dirichlet((1, 1, 1), m).astype(np.float32)
    p_on = (vtx[tri[:, 0]] * w[:, :1] + vtx[tri[:, 1]] * w[:, 1:2] + vtx[tri[:, 2]] * w[:, 2:])
    o[onsurf] = p_on[onsurf]
    rd = rng.normal(size=(m, 3)).astype(np.float32)
    d[onsurf] = (rd / np.linalg.norm(rd, axis=1, keepdims=True))[onsurf]
    rays["o"][:, :3] = o
    rays["d"][:, :3] = d
    rays["o"][:, 3] = np.where(rng.random(m) < 0.2, rng.choice(np.float32([1e-6, 1e-3, 0.5, 3.0]), m), np.float32(1e30))
    return rays
