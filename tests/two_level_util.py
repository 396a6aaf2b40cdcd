"""A small two-level scene for the instancing tests: shared by the CPU and the GPU tests."""
import numpy as np


def _rot(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    axis /= np.linalg.norm(axis)
    x, y, z = axis
    c, s = np.cos(angle), np.sin(angle)
    return np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
                     [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s],
                     [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)]])


def _affine(linear, t):
    m = np.zeros((3, 4), dtype=np.float32)
    m[:, :3] = np.asarray(linear, dtype=np.float32)
    m[:, 3] = np.asarray(t, dtype=np.float32)
    return m


def blob_mesh(n_tri, seed):
    """Random small triangles on a unit sphere-ish shell: a closed-looking clump."""
    rng = np.random.default_rng(seed)
    c = rng.normal(size=(n_tri, 3))
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    c *= rng.uniform(0.7, 1.0, size=(n_tri, 1))
    v = (c[:, None, :] + rng.normal(scale=0.18, size=(n_tri, 3, 3))).reshape(-1, 3).astype(np.float32)
    idx = np.arange(3 * n_tri, dtype=np.int32).reshape(-1, 3)
    mids = (np.arange(n_tri) % 2).astype(np.int32)          # grey / glossy
    return v, idx, mids


def quad_mesh():
    v = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], dtype=np.float32)
    idx = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int32)
    return v, idx, np.zeros(2, dtype=np.int32)


def single_triangle_mesh():
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32)
    return v, np.array([[0, 1, 2]], dtype=np.int32), np.zeros(1, dtype=np.int32)


def materials(mipt_or_oracle):
    m = np.zeros(3, dtype=mipt_or_oracle.material_dtype)
    m["kd"][0] = (0.7, 0.7, 0.7); m["roughness"][0] = 1.0
    m["kd"][1] = (0.2, 0.3, 0.5); m["ks"][1] = (0.6, 0.6, 0.6); m["roughness"][1] = 0.25
    m["ke"][2] = (12.0, 11.0, 9.0); m["roughness"][2] = 1.0
    return m


def scene(n_blob_tris=300, n_blobs=9, seed=5):
    """Returns dict(meshes=[(vtx, idx, mids)], transforms [n,3,4], mesh_of, material_offsets)."""
    rng = np.random.default_rng(seed)
    meshes = [quad_mesh(), blob_mesh(n_blob_tris, seed), single_triangle_mesh(),
              blob_mesh(max(8, n_blob_tris // 7), seed + 1)]
    tr, mesh_of, moff = [], [], []

    def add(m, mesh, off=0):
        tr.append(m); mesh_of.append(mesh); moff.append(off)

    add(_affine(np.eye(3) * 9.0, (0, -2.0, 0)), 0)                       # floor
    add(_affine(_rot((1, 0, 0), np.pi) * 2.0, (0, 6.0, 0)), 0, 2)         # lamp, facing down
    for k in range(n_blobs):
        lin = _rot(rng.normal(size=3), rng.uniform(0, 2 * np.pi)) @ np.diag(rng.uniform(0.5, 1.6, size=3))
        if k % 4 == 3:
            lin = lin @ np.diag([-1.0, 1.0, 1.0])                        # mirrored
        add(_affine(lin, rng.uniform(-5, 5, size=3) * (1, 0.4, 1)), 1 if k % 3 else 3)
    # two copies of one mesh in exactly the same place: every hit on them is a tie between
    # instances, which the fixed top-level order decides
    dup = _affine(_rot((0, 1, 0), 0.3) * 1.2, (2.5, 1.0, -3.0))
    add(dup, 3)
    add(dup.copy(), 3)
    add(_affine(np.eye(3) * 2.0, (-3.0, 0.5, 2.0)), 2)                    # single-triangle meshes
    add(_affine(_rot((0, 0, 1), 1.0) * 3.0, (3.0, -1.0, 3.0)), 2, 1)
    return dict(meshes=meshes, transforms=np.stack(tr), mesh_of=mesh_of, material_offsets=moff)


def rays_for(n, seed, ray_dtype):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, dtype=ray_dtype)
    rays["o"][:, :3] = rng.uniform(-8, 8, size=(n, 3)).astype(np.float32) * (1, 0.6, 1)
    rays["o"][:, 3] = np.where(rng.uniform(size=n) < 0.2, rng.uniform(0.5, 8.0, size=n), 1e30)
    target = rng.uniform(-5, 5, size=(n, 3)) * (1, 0.4, 1)
    d = target - rays["o"][:, :3]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays["d"][:, :3] = d.astype(np.float32)
    # a few degenerate ones
    rays["d"][0, :3] = 0
    rays["o"][1, 0] = np.nan
    rays["d"][2, :3] = (0, -1, 0)
    rays["o"][3, 3] = 0.0
    return rays


def flat_material_ids(sc):
    out = []
    for k, m in enumerate(sc["mesh_of"]):
        out.append(sc["meshes"][m][2] + sc["material_offsets"][k])
    return np.concatenate(out).astype(np.int32)
