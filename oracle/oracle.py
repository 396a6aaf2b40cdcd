"""ctypes loader for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
the cpu_baseline leg of bench.py as the checker / host-core baseline.  The
product library (libmipt.so) never links or calls it.  Parity status and the
floating-point contract are stated in pt_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

bbox_dtype = np.dtype([("pmin", np.float32, 4), ("pmax", np.float32, 4)])
ray_dtype = np.dtype([("o", np.float32, 4), ("d", np.float32, 4)])
isect_dtype = np.dtype([("shapeid", np.int32), ("primid", np.int32), ("padding", np.int32, 2),
                        ("uvwt", np.float32, 4)])


class Camera(C.Structure):
    _fields_ = [("ip00", C.c_float), ("ip11", C.c_float),
                ("inv_view_c0", C.c_float * 3), ("inv_view_c1", C.c_float * 3),
                ("inv_view_c2", C.c_float * 3), ("camera_position", C.c_float * 3)]


class Counters(C.Structure):
    _fields_ = [("box_tests", C.c_uint64), ("tri_tests", C.c_uint64)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc if the shared object is missing or stale."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(LIB_PATH) or
             any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs))
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        sig = {
            "o_closest_batch": (None, [vp, vp, vp, vp, C.c_uint64, vp, C.POINTER(Counters), C.c_int]),
            "o_closest_mt_batch": (None, [vp, vp, vp, vp, C.c_uint64, vp, C.c_int]),
            "o_any_batch": (None, [vp, vp, vp, vp, C.c_uint64, vp, C.c_int]),
            "o_brute_closest_batch": (None, [vp, vp, C.c_uint32, vp, C.c_uint64, vp, C.c_int]),
            "o_raygen_reference_image": (None, [C.POINTER(Camera), C.c_uint32, C.c_uint32, vp]),
            "o_depth_shade_batch": (None, [vp, C.c_uint64, vp]),
            "o_mesh_leaf_bounds": (None, [vp, vp, C.c_uint32, vp]),
            "o_inflate_leaf_bounds": (None, [vp, C.c_uint32]),
            "o_build_bvh": (C.c_int, [vp, C.c_uint32, C.c_int, C.c_int, C.c_float, vp]),
            "o_linearize_arrays": (C.c_int, [vp, vp, vp, vp, C.c_int32, vp]),
            "o_intersect_triangle_watertight": (C.c_int, [vp, vp, vp, vp, vp]),
            "o_intersect_triangle_mt": (C.c_int, [vp, vp, vp, vp, vp]),
            "o_precompute_ray": (None, [vp, vp]),
            "o_invert_affine": (C.c_int, [vp, vp]),
            "o_transform_points": (None, [vp, vp, C.c_uint64, vp]),
            "o_instance_bounds": (None, [vp, vp, vp]),
            "o_two_level_closest_batch": (None, [vp, vp, vp, vp, C.c_uint64, vp,
                                                 C.POINTER(Counters), C.c_int]),
            "o_two_level_any_batch": (None, [vp, vp, vp, vp, C.c_uint64, vp, C.c_int]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def mesh_leaf_bounds(vtx, idx) -> np.ndarray:
    vtx, idx = _c(vtx, np.float32), _c(idx, np.int32)
    out = np.zeros(idx.size // 3, dtype=bbox_dtype)
    lib().o_mesh_leaf_bounds(vtx.ctypes.data, idx.ctypes.data, idx.size // 3, out.ctypes.data)
    return out


def inflate(boxes) -> np.ndarray:
    b = _c(boxes, bbox_dtype).copy()
    lib().o_inflate_leaf_bounds(b.ctypes.data, b.shape[0])
    return b


def build_bvh(leaf_bounds, use_sah=True, num_bins=64, traversal_cost=10.0) -> np.ndarray:
    lb = _c(leaf_bounds, bbox_dtype)
    out = np.zeros(2 * lb.shape[0] - 1, dtype=bbox_dtype)
    rc = lib().o_build_bvh(lb.ctypes.data, lb.shape[0], int(use_sah), num_bins, traversal_cost,
                           out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"o_build_bvh failed: {rc}")
    return out


def linearize(bounds, left, right, prim) -> np.ndarray:
    bounds = _c(bounds, bbox_dtype)
    left, right, prim = _c(left, np.int32), _c(right, np.int32), _c(prim, np.int32)
    out = np.zeros(bounds.shape[0], dtype=bbox_dtype)
    rc = lib().o_linearize_arrays(bounds.ctypes.data, left.ctypes.data, right.ctypes.data,
                                  prim.ctypes.data, bounds.shape[0], out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"o_linearize_arrays failed: {rc}")
    return out


def closest(nodes, vtx, idx, rays, nthreads=1, counters=False, moller_trumbore=False,
            int_bits=False):
    nodes, vtx, idx, rays = (_c(nodes, bbox_dtype), _c(vtx, np.float32), _c(idx, np.int32),
                             _c(rays, ray_dtype))
    out = np.zeros(rays.shape[0], dtype=isect_dtype)
    if int_bits:
        fn = lib().o_closest_intbits_batch
        fn.restype = None
        fn.argtypes = [C.c_void_p] * 4 + [C.c_uint64, C.c_void_p, C.c_int]
        fn(nodes.ctypes.data, vtx.ctypes.data, idx.ctypes.data, rays.ctypes.data, rays.shape[0],
           out.ctypes.data, nthreads)
        return out
    if moller_trumbore:
        lib().o_closest_mt_batch(nodes.ctypes.data, vtx.ctypes.data, idx.ctypes.data,
                                 rays.ctypes.data, rays.shape[0], out.ctypes.data, nthreads)
        return out
    cnt = Counters()
    lib().o_closest_batch(nodes.ctypes.data, vtx.ctypes.data, idx.ctypes.data, rays.ctypes.data,
                          rays.shape[0], out.ctypes.data, C.byref(cnt), nthreads)
    return (out, cnt) if counters else out


def any_hit(nodes, vtx, idx, rays, nthreads=1) -> np.ndarray:
    """IntersectSceneAny over a node array with float-valued links (the reference's encoding; the
    integer-bit encoding is not read here and would be misinterpreted)."""
    nodes, vtx, idx, rays = (_c(nodes, bbox_dtype), _c(vtx, np.float32), _c(idx, np.int32),
                             _c(rays, ray_dtype))
    out = np.zeros(rays.shape[0], dtype=np.int32)
    lib().o_any_batch(nodes.ctypes.data, vtx.ctypes.data, idx.ctypes.data, rays.ctypes.data,
                      rays.shape[0], out.ctypes.data, nthreads)
    return out


def brute_closest(vtx, idx, rays, nthreads=1) -> np.ndarray:
    vtx, idx, rays = _c(vtx, np.float32), _c(idx, np.int32), _c(rays, ray_dtype)
    out = np.zeros(rays.shape[0], dtype=isect_dtype)
    lib().o_brute_closest_batch(vtx.ctypes.data, idx.ctypes.data, idx.size // 3, rays.ctypes.data,
                                rays.shape[0], out.ctypes.data, nthreads)
    return out


def raygen_reference(cam: Camera, width: int, height: int) -> np.ndarray:
    out = np.zeros(width * height, dtype=ray_dtype)
    lib().o_raygen_reference_image(C.byref(cam), width, height, out.ctypes.data)
    return out


def depth_shade(isects) -> np.ndarray:
    isects = _c(isects, isect_dtype)
    out = np.zeros((isects.shape[0], 4), dtype=np.float32)
    lib().o_depth_shade_batch(isects.ctypes.data, isects.shape[0], out.ctypes.data)
    return out


class ShadeCamera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("sx", C.c_float), ("c0", C.c_float * 3), ("sy", C.c_float),
                ("c1", C.c_float * 3), ("mode", C.c_uint32), ("c2", C.c_float * 3), ("pad", C.c_float)]


class ShadeParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("sample_begin", C.c_uint32),
                ("sample_count", C.c_uint32), ("max_depth", C.c_uint32), ("seed", C.c_uint32),
                ("rr_start_depth", C.c_uint32), ("flags", C.c_uint32), ("tile_size", C.c_uint32),
                ("shard_index", C.c_uint32), ("shard_count", C.c_uint32),
                ("samples_per_pass", C.c_uint32), ("background", C.c_float * 3),
                ("ray_offset", C.c_float)]


material_dtype = np.dtype([("kd", np.float32, 3), ("roughness", np.float32),
                           ("ks", np.float32, 3), ("_pad0", np.float32),
                           ("ke", np.float32, 3), ("_pad1", np.float32)])


def render(nodes, vtx, idx, materials, material_ids, cam, params, film=None, rect=None,
           nthreads=1):
    """CPU restatement of the path tracer.  cam / params: any ctypes structs with the byte
    layout of ShadeCamera / ShadeParams (the product's pt_camera / pt_render_params have it).
    Adds into `film` ([H, W, 4] float32) and returns (film, rays_traced)."""
    L = lib()
    L.os_render.restype = C.c_uint64
    L.os_render.argtypes = [C.c_void_p] * 7 + [C.c_uint32] * 4 + [C.c_void_p, C.c_int, C.c_uint32]
    nodes, vtx, idx = _c(nodes, bbox_dtype), _c(vtx, np.float32), _c(idx, np.int32)
    mats, mids = _c(materials, material_dtype), _c(material_ids, np.int32)
    assert C.sizeof(cam) == C.sizeof(ShadeCamera) and C.sizeof(params) == C.sizeof(ShadeParams)
    W, H = params.width, params.height
    if film is None:
        film = np.zeros((H, W, 4), dtype=np.float32)
    x0, y0, x1, y1 = rect if rect is not None else (0, 0, W, H)
    rays = L.os_render(nodes.ctypes.data, vtx.ctypes.data, idx.ctypes.data, mats.ctypes.data,
                       mids.ctypes.data, C.addressof(cam), C.addressof(params), x0, y0, x1, y1,
                       film.ctypes.data, nthreads, idx.shape[0])
    return film, int(rays)


# ---- two-level scenes --------------------------------------------------------------------
class _OMesh(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("vertices", C.c_void_p), ("indices", C.c_void_p)]


class _OInstance(C.Structure):
    _fields_ = [("inv", C.c_float * 12), ("mesh", C.c_int32), ("pad", C.c_int32 * 3)]


def invert_affine(m) -> np.ndarray:
    m = _c(m, np.float32).reshape(12)
    out = np.empty(12, dtype=np.float32)
    if not lib().o_invert_affine(m.ctypes.data, out.ctypes.data):
        raise ValueError("singular transform")
    return out.reshape(3, 4)


def transform_points(m, p) -> np.ndarray:
    m = _c(m, np.float32).reshape(12)
    p = _c(p, np.float32).reshape(-1, 3)
    out = np.empty_like(p)
    lib().o_transform_points(m.ctypes.data, p.ctypes.data, p.shape[0], out.ctypes.data)
    return out


class TwoLevel:
    """meshes: [(nodes, vtx, idx)], each with its own skip-link tree; transforms: [n, 3, 4]
    object-to-world; mesh_of: mesh index per instance; top: skip-link tree over the instances
    (None until bounds() has been used to build it; assign .top)."""

    def __init__(self, meshes, transforms, mesh_of, top=None):
        self.meshes = [(_c(n, bbox_dtype), _c(v, np.float32).reshape(-1, 3),
                        _c(i, np.int32).reshape(-1, 3)) for n, v, i in meshes]
        self.transforms = _c(transforms, np.float32).reshape(-1, 3, 4)
        self.mesh_of = [int(m) for m in mesh_of]
        assert len(self.mesh_of) == self.transforms.shape[0]
        self.top = None if top is None else _c(top, bbox_dtype)
        self._m = (_OMesh * len(self.meshes))()
        for k, (n, v, i) in enumerate(self.meshes):
            self._m[k].nodes, self._m[k].vertices, self._m[k].indices = \
                n.ctypes.data, v.ctypes.data, i.ctypes.data
        self._i = (_OInstance * len(self.mesh_of))()
        for k, m in enumerate(self.mesh_of):
            inv = invert_affine(self.transforms[k]).reshape(12)
            for j in range(12):
                self._i[k].inv[j] = float(inv[j])
            self._i[k].mesh = m

    def bounds(self) -> np.ndarray:
        """World-space leaf boxes of the instances, for the top-level build."""
        out = np.zeros(len(self.mesh_of), dtype=bbox_dtype)
        L = lib()
        for k, m in enumerate(self.mesh_of):
            t = _c(self.transforms[k], np.float32).reshape(12)
            L.o_instance_bounds(t.ctypes.data, self.meshes[m][0][:1].ctypes.data,
                                out[k:k + 1].ctypes.data)
        return out

    def world_triangles(self):
        """(vtx [3T,3], idx [T,3], first_tri [n_instances]): every instanced triangle in
        world space, instance after instance -- what shading reads."""
        vs, first, n = [], [], 0
        for k, m in enumerate(self.mesh_of):
            _, v, i = self.meshes[m]
            vs.append(transform_points(self.transforms[k], v[i.reshape(-1)]))
            first.append(n)
            n += i.shape[0]
        vtx = np.concatenate(vs, axis=0)
        idx = np.arange(3 * n, dtype=np.int32).reshape(-1, 3)
        return vtx, idx, np.asarray(first, dtype=np.uint32)

    def closest(self, rays, nthreads=1, counters=False):
        rays = _c(rays, ray_dtype)
        out = np.empty(rays.shape[0], dtype=isect_dtype)
        cnt = Counters(0, 0)
        lib().o_two_level_closest_batch(self.top.ctypes.data, C.addressof(self._m),
                                        C.addressof(self._i), rays.ctypes.data, rays.shape[0],
                                        out.ctypes.data, C.byref(cnt), nthreads)
        return (out, (cnt.box_tests, cnt.tri_tests)) if counters else out

    def any_hit(self, rays, nthreads=1) -> np.ndarray:
        rays = _c(rays, ray_dtype)
        out = np.empty(rays.shape[0], dtype=np.int32)
        lib().o_two_level_any_batch(self.top.ctypes.data, C.addressof(self._m),
                                    C.addressof(self._i), rays.ctypes.data, rays.shape[0],
                                    out.ctypes.data, nthreads)
        return out

    def render(self, materials, material_ids, cam, params, film=None, rect=None, nthreads=1):
        """As render(); material_ids has one entry per instanced triangle (world_triangles
        order)."""
        L = lib()
        L.os_render_two_level.restype = C.c_uint64
        L.os_render_two_level.argtypes = [C.c_void_p] * 10 + [C.c_uint32] * 4 + \
            [C.c_void_p, C.c_int, C.c_uint32]
        vtx, idx, first = self.world_triangles()
        mats, mids = _c(materials, material_dtype), _c(material_ids, np.int32)
        assert mids.shape[0] == idx.shape[0]
        W, H = params.width, params.height
        if film is None:
            film = np.zeros((H, W, 4), dtype=np.float32)
        x0, y0, x1, y1 = rect if rect is not None else (0, 0, W, H)
        rays = L.os_render_two_level(self.top.ctypes.data, C.addressof(self._m),
                                     C.addressof(self._i), first.ctypes.data, vtx.ctypes.data,
                                     idx.ctypes.data, mats.ctypes.data, mids.ctypes.data,
                                     C.addressof(cam), C.addressof(params), x0, y0, x1, y1,
                                     film.ctypes.data, nthreads, idx.shape[0])
        return film, int(rays)


def sincos_2pi(u):
    L = lib()
    L.os_sincos_2pi.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    s, c = C.c_float(), C.c_float()
    L.os_sincos_2pi(u, C.byref(s), C.byref(c))
    return s.value, c.value


def rnd(seed, pixel, sample, dim):
    L = lib()
    L.os_rnd.restype = C.c_float
    L.os_rnd.argtypes = [C.c_uint32] * 4
    return L.os_rnd(seed, pixel, sample, dim)


def camera_from(eye, c0, c1, c2, sx, sy) -> Camera:
    cam = Camera()
    cam.ip00, cam.ip11 = sx, sy
    for k in range(3):
        cam.inv_view_c0[k] = c0[k]
        cam.inv_view_c1[k] = c1[k]
        cam.inv_view_c2[k] = c2[k]
        cam.camera_position[k] = eye[k]
    return cam
