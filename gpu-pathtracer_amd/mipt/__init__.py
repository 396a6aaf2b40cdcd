"""ctypes binding of libmipt.so -- the MI355X path-tracing core.

Thin mirror of ``include/mipt.h``: numpy arrays in, numpy arrays out, every
non-zero status raised as :class:`MiptError` with ``pt_last_error()``.  The
shared library is built in-tree by ``gpu-pathtracer_amd/Makefile`` (see
``__graft_entry__.build``); importing this module fails loudly if it is
missing -- there is no Python or CPU fallback for the device entry points.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIPT_LIB") or os.path.join(os.path.dirname(_HERE), "libmipt.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `make -C {os.path.dirname(_HERE)}` "
        "(or __graft_entry__.build())")

lib = C.CDLL(LIB_PATH)

# ---------------------------------------------------------------- constants
PT_OK, PT_ERR_INVALID, PT_ERR_DEVICE, PT_ERR_UNSUPPORTED, PT_ERR_IO, PT_ERR_NOMEM, PT_ERR_STALE = range(7)
PT_IMAGE_PPM, PT_IMAGE_PFM, PT_IMAGE_PNG = 0, 1, 2
PT_IMAGE_TONEMAP_REINHARD, PT_IMAGE_TONEMAP_FILMIC = 0x100, 0x200
PT_LINKS_FLOAT_VALUE, PT_LINKS_INT_BITS = 0, 1
PT_CAMERA_REFERENCE, PT_CAMERA_PINHOLE = 0, 1
PT_TRACE_DEFAULT, PT_TRACE_EXACT_ORDER, PT_TRACE_MOLLER_TRUMBORE, PT_TRACE_NO_TIE_FIXUP = 0, 1, 2, 4
PT_RENDER_DEFAULT, PT_RENDER_DEPTH_GREY, PT_RENDER_NEE, PT_RENDER_PIXEL_CENTRE = 0, 1, 2, 4
PT_RENDER_COUNT_WORK = 8

# ---------------------------------------------------------------- dtypes
bbox_dtype = np.dtype([("pmin", np.float32, 4), ("pmax", np.float32, 4)])
ray_dtype = np.dtype([("o", np.float32, 4), ("d", np.float32, 4)])
hit_dtype = np.dtype([("shapeid", np.int32), ("primid", np.int32), ("padding", np.int32, 2),
                      ("uvwt", np.float32, 4)])
material_dtype = np.dtype([("kd", np.float32, 3), ("roughness", np.float32),
                           ("ks", np.float32, 3), ("_pad0", np.float32),
                           ("ke", np.float32, 3), ("_pad1", np.float32)])
assert bbox_dtype.itemsize == 32 and ray_dtype.itemsize == 32 and hit_dtype.itemsize == 32
assert material_dtype.itemsize == 48


class BvhOptions(C.Structure):
    _fields_ = [("use_sah", C.c_int32), ("use_splits", C.c_int32), ("max_split_depth", C.c_int32),
                ("num_bins", C.c_int32), ("min_overlap", C.c_float), ("traversal_cost", C.c_float),
                ("extra_node_budget", C.c_float), ("link_encoding", C.c_uint32),
                ("num_threads", C.c_int32), ("optimize_passes", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("sx", C.c_float),
                ("c0", C.c_float * 3), ("sy", C.c_float),
                ("c1", C.c_float * 3), ("mode", C.c_uint32),
                ("c2", C.c_float * 3), ("_pad", C.c_float)]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("sample_begin", C.c_uint32),
                ("sample_count", C.c_uint32), ("max_depth", C.c_uint32), ("seed", C.c_uint32),
                ("rr_start_depth", C.c_uint32), ("flags", C.c_uint32), ("tile_size", C.c_uint32),
                ("shard_index", C.c_uint32), ("shard_count", C.c_uint32),
                ("samples_per_pass", C.c_uint32), ("background", C.c_float * 3),
                ("ray_offset", C.c_float)]


class TraceStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64),
                ("tie_rays", C.c_uint64), ("lds_box_tests", C.c_uint64), ("long_rays", C.c_uint64)]


class RenderStats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays_primary", C.c_uint64), ("rays_bounce", C.c_uint64),
                ("rays_shadow", C.c_uint64), ("tie_rays", C.c_uint64),
                ("extend_launches", C.c_uint64), ("extend_ms", C.c_double),
                ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64),
                ("raygen_ms", C.c_double), ("shade_ms", C.c_double), ("occlude_ms", C.c_double),
                ("accumulate_ms", C.c_double), ("lds_box_tests", C.c_uint64)]


class MeshDesc(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("vertices", C.c_void_p), ("indices", C.c_void_p),
                ("material_ids", C.c_void_p), ("n_nodes", C.c_uint32), ("n_vertices", C.c_uint32),
                ("n_triangles", C.c_uint32), ("_pad", C.c_uint32)]


class Instance(C.Structure):
    _fields_ = [("object_to_world", C.c_float * 12), ("mesh", C.c_uint32),
                ("material_offset", C.c_int32)]


class MiptError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"[{code}] {message}")
        self.code = code
        self.message = message


# ---------------------------------------------------------------- prototypes
_vp = C.c_void_p
_PROTOS = {
    "pt_last_error": (C.c_char_p, []),
    "pt_version": (C.c_char_p, []),
    "pt_device_count": (C.c_int, []),
    "pt_bvh_options_default": (None, [C.POINTER(BvhOptions)]),
    "pt_bvh_options_set": (C.c_int, [C.POINTER(BvhOptions), C.c_char_p, C.c_char_p]),
    "pt_mesh_leaf_bounds": (C.c_int, [_vp, C.c_uint32, _vp, C.c_uint32, _vp]),
    "pt_bvh_build": (C.c_int, [_vp, C.c_uint32, C.POINTER(BvhOptions), C.POINTER(_vp)]),
    "pt_bvh_optimize": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_int32, C.POINTER(_vp)]),
    "pt_bvh_node_count": (C.c_uint32, [_vp]),
    "pt_bvh_nodes": (_vp, [_vp]),
    "pt_bvh_link_encoding": (C.c_uint32, [_vp]),
    "pt_bvh_destroy": (C.c_int, [_vp]),
    "pt_bvh_linearize": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, C.c_uint32, _vp]),
    "pt_bvh_validate": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32]),
    "pt_obj_load": (C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    "pt_gltf_load": (C.c_int, [C.c_char_p, C.c_uint32, C.POINTER(_vp)]),
    "pt_mesh_vertex_count": (C.c_uint32, [_vp]),
    "pt_mesh_triangle_count": (C.c_uint32, [_vp]),
    "pt_mesh_material_count": (C.c_uint32, [_vp]),
    "pt_mesh_vertices": (_vp, [_vp]),
    "pt_mesh_indices": (_vp, [_vp]),
    "pt_mesh_material_ids": (_vp, [_vp]),
    "pt_mesh_materials": (_vp, [_vp]),
    "pt_mesh_shape_count": (C.c_uint32, [_vp]),
    "pt_mesh_shape_ids": (_vp, [_vp]),
    "pt_mesh_shape_prim_ids": (_vp, [_vp]),
    "pt_mesh_normalize_unit_cube": (C.c_int, [_vp]),
    "pt_mesh_destroy": (C.c_int, [_vp]),
    "pt_context_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "pt_context_set_stream": (C.c_int, [_vp, _vp]),
    "pt_context_stream": (_vp, [_vp]),
    "pt_context_synchronize": (C.c_int, [_vp]),
    "pt_context_set_kernel_timing": (C.c_int, [_vp, C.c_int]),
    "pt_context_set_option": (C.c_int, [_vp, C.c_char_p, C.c_char_p]),
    "pt_context_last_trace_ms": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "pt_context_destroy": (C.c_int, [_vp]),
    "pt_measure_gather": (C.c_int, [_vp, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]),
    "pt_scene_create": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, _vp, C.c_uint32, _vp,
                                  C.c_uint32, _vp, C.c_uint32, _vp, C.POINTER(_vp)]),
    "pt_scene_destroy": (C.c_int, [_vp]),
    "pt_scene_build": (C.c_int, [_vp, _vp, C.c_uint32, _vp, C.c_uint32, _vp, C.c_uint32, _vp,
                                 C.c_uint32, C.POINTER(_vp)]),
    "pt_instance_bounds": (C.c_int, [C.POINTER(MeshDesc), C.c_uint32, C.POINTER(Instance),
                                     C.c_uint32, _vp]),
    "pt_scene_create_instanced": (C.c_int, [_vp, C.POINTER(MeshDesc), C.c_uint32,
                                            C.POINTER(Instance), C.c_uint32, _vp, C.c_uint32,
                                            C.c_uint32, _vp, C.c_uint32, C.POINTER(_vp)]),
    "pt_scene_instance_count": (C.c_uint32, [_vp]),
    "pt_gltf_load_model": (C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    "pt_model_mesh_count": (C.c_uint32, [_vp]),
    "pt_model_mesh": (_vp, [_vp, C.c_uint32]),
    "pt_model_instance_count": (C.c_uint32, [_vp]),
    "pt_model_instances": (C.POINTER(Instance), [_vp]),
    "pt_model_material_count": (C.c_uint32, [_vp]),
    "pt_model_materials": (_vp, [_vp]),
    "pt_model_destroy": (C.c_int, [_vp]),
    "pt_scene_top_node_count": (C.c_uint32, [_vp]),
    "pt_scene_read_top_nodes": (C.c_int, [_vp, _vp]),
    "pt_scene_mesh_node_count": (C.c_uint32, [_vp, C.c_uint32]),
    "pt_scene_read_mesh_nodes": (C.c_int, [_vp, C.c_uint32, _vp]),
    "pt_scene_node_count": (C.c_uint32, [_vp]),
    "pt_scene_link_encoding": (C.c_uint32, [_vp]),
    "pt_scene_tree_depth": (C.c_uint32, [_vp]),
    "pt_scene_set_face_ids": (C.c_int, [_vp, _vp, _vp, C.c_uint32]),
    "pt_scene_triangle_count": (C.c_uint32, [_vp]),
    "pt_scene_read_nodes": (C.c_int, [_vp, _vp]),
    "pt_trace_closest": (C.c_int, [_vp, _vp, _vp, C.c_uint64, _vp, C.c_uint32,
                                   C.POINTER(TraceStats)]),
    "pt_trace_closest_device": (C.c_int, [_vp, _vp, _vp, C.c_uint64, _vp, C.c_uint32]),
    "pt_trace_closest_device_static": (C.c_int, [_vp, _vp, _vp, C.c_uint64, _vp, C.c_uint32]),
    "pt_trace_closest_device_counted": (C.c_int, [_vp, _vp, _vp, C.c_uint64, _vp, C.c_uint32,
                                                  C.POINTER(TraceStats)]),
    "pt_trace_any": (C.c_int, [_vp, _vp, _vp, C.c_uint64, _vp]),
    "pt_trace_any_device": (C.c_int, [_vp, _vp, _vp, C.c_uint64, _vp]),
    "pt_camera_from_matrices": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                          C.c_uint32, C.POINTER(Camera)]),
    "pt_camera_from_orientation": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float,
                                             C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(Camera)]),
    "pt_camera_look_at": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                    C.c_float, C.c_float, C.c_uint32, C.POINTER(Camera)]),
    "pt_raygen": (C.c_int, [_vp, C.POINTER(Camera), C.c_uint32, C.c_uint32, _vp]),
    "pt_raygen_device": (C.c_int, [_vp, C.POINTER(Camera), C.c_uint32, C.c_uint32, _vp]),
    "pt_depth_shade": (C.c_int, [_vp, _vp, C.c_uint64, _vp]),
    "pt_film_create": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.POINTER(_vp)]),
    "pt_film_wrap": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _vp, C.POINTER(_vp)]),
    "pt_film_clear": (C.c_int, [_vp]),
    "pt_film_device_ptr": (_vp, [_vp]),
    "pt_film_read": (C.c_int, [_vp, _vp]),
    "pt_film_destroy": (C.c_int, [_vp]),
    "pt_bvh_cache_key": (C.c_uint64, [_vp, C.c_uint32, _vp]),
    "pt_bvh_save": (C.c_int, [_vp, C.c_char_p, C.c_uint64]),
    "pt_bvh_load": (C.c_int, [C.c_char_p, C.c_uint64, C.POINTER(_vp)]),
    "pt_film_save": (C.c_int, [_vp, C.c_char_p]),
    "pt_film_load": (C.c_int, [_vp, C.c_char_p]),
    "pt_film_write_image": (C.c_int, [_vp, C.c_char_p, C.c_uint32, C.c_float]),
    "pt_shard_layout": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                  C.POINTER(C.c_uint64), _vp]),
    "pt_film_packed_size": (C.c_uint64, [_vp, C.c_uint32, C.c_uint32, C.c_uint32]),
    "pt_film_pack_tiles": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp]),
    "pt_film_unpack_tiles": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp]),
    "pt_comm_available": (C.c_int, []),
    "pt_comm_unique_id": (C.c_int, [_vp]),
    "pt_comm_create": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "pt_comm_create_all": (C.c_int, [C.POINTER(_vp), C.c_int, C.POINTER(_vp)]),
    "pt_comm_rank": (C.c_int, [_vp]),
    "pt_comm_size": (C.c_int, [_vp]),
    "pt_comm_destroy": (C.c_int, [_vp]),
    "pt_film_reduce": (C.c_int, [_vp, _vp, C.c_int]),
    "pt_film_gather_tiles": (C.c_int, [_vp, _vp, C.c_uint32, C.c_int]),
    "pt_render": (C.c_int, [_vp, _vp, C.POINTER(Camera), C.POINTER(RenderParams), _vp,
                            C.POINTER(RenderStats)]),
    "pt_render_params_default": (None, [C.POINTER(RenderParams), C.c_uint32, C.c_uint32]),
}
for _name, (_res, _args) in _PROTOS.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args


def last_error() -> str:
    return lib.pt_last_error().decode(errors="replace")     # messages may quote bytes of a damaged file


def _check(rc: int) -> None:
    if rc != PT_OK:
        raise MiptError(rc, last_error())


def _ptr(a: np.ndarray) -> int:
    return a.ctypes.data


def _as(a, dtype, name) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=dtype)
    if not a.flags.c_contiguous:
        raise ValueError(f"{name} must be contiguous")
    return a


def device_count() -> int:
    return int(lib.pt_device_count())


# ---------------------------------------------------------------- host BVH path
def bvh_options(**kw) -> BvhOptions:
    """Options with the reference's string keys, e.g. ``bvh_options(**{"bvh.builder": "sah"})``."""
    opt = BvhOptions()
    lib.pt_bvh_options_default(C.byref(opt))
    for k, v in kw.items():
        _check(lib.pt_bvh_options_set(C.byref(opt), k.encode(), str(v).encode()))
    return opt


def mesh_leaf_bounds(vtx, idx) -> np.ndarray:
    vtx = _as(vtx, np.float32, "vtx").reshape(-1, 3)
    idx = _as(idx, np.int32, "idx").reshape(-1, 3)
    out = np.empty(idx.shape[0], dtype=bbox_dtype)
    _check(lib.pt_mesh_leaf_bounds(_ptr(vtx), vtx.shape[0], _ptr(idx), idx.shape[0], _ptr(out)))
    return out


def build_bvh(leaf_bounds, options: Optional[BvhOptions] = None) -> np.ndarray:
    """``bvh::build_bvh``: leaf boxes -> skip-link node array (a copy owned by numpy)."""
    leaf_bounds = _as(leaf_bounds, bbox_dtype, "leaf_bounds")
    h = _vp()
    _check(lib.pt_bvh_build(_ptr(leaf_bounds), leaf_bounds.shape[0],
                            C.byref(options) if options is not None else None, C.byref(h)))
    try:
        n = lib.pt_bvh_node_count(h)
        src = (C.c_char * (n * 32)).from_address(lib.pt_bvh_nodes(h))
        return np.frombuffer(src, dtype=bbox_dtype).copy()
    finally:
        lib.pt_bvh_destroy(h)


def optimize_bvh(nodes, passes: int = 2, link_encoding: int = PT_LINKS_FLOAT_VALUE) -> np.ndarray:
    """``pt_bvh_optimize``: insertion-based optimisation of any skip-link tree."""
    nodes = _as(nodes, bbox_dtype, "nodes")
    h = _vp()
    _check(lib.pt_bvh_optimize(_ptr(nodes), nodes.shape[0], link_encoding, passes, C.byref(h)))
    return _take_nodes(h)


def _take_nodes(h) -> np.ndarray:
    try:
        n = lib.pt_bvh_node_count(h)
        src = (C.c_char * (n * 32)).from_address(lib.pt_bvh_nodes(h))
        return np.frombuffer(src, dtype=bbox_dtype).copy()
    finally:
        lib.pt_bvh_destroy(h)


def build_bvh_cached(leaf_bounds, path: str, options: Optional[BvhOptions] = None):
    """Load the BVH of these leaf boxes + options from `path` if it holds one, else build
    and save it.  Returns (nodes, loaded_from_cache)."""
    leaf_bounds = _as(leaf_bounds, bbox_dtype, "leaf_bounds")
    opt = C.byref(options) if options is not None else None
    key = lib.pt_bvh_cache_key(_ptr(leaf_bounds), leaf_bounds.shape[0], opt)
    h = _vp()
    rc = lib.pt_bvh_load(path.encode(), key, C.byref(h))
    if rc == PT_OK:
        return _take_nodes(h), True
    if rc not in (PT_ERR_IO, PT_ERR_STALE):
        _check(rc)
    _check(lib.pt_bvh_build(_ptr(leaf_bounds), leaf_bounds.shape[0], opt, C.byref(h)))
    try:
        _check(lib.pt_bvh_save(h, path.encode(), key))
    except MiptError:
        lib.pt_bvh_destroy(h)
        raise
    return _take_nodes(h), False


def linearize(bounds, left, right, prim, link_encoding=PT_LINKS_FLOAT_VALUE) -> np.ndarray:
    bounds = _as(bounds, bbox_dtype, "bounds")
    left = _as(left, np.int32, "left")
    right = _as(right, np.int32, "right")
    prim = _as(prim, np.int32, "prim")
    out = np.empty(bounds.shape[0], dtype=bbox_dtype)
    _check(lib.pt_bvh_linearize(_ptr(bounds), _ptr(left), _ptr(right), _ptr(prim), bounds.shape[0],
                                link_encoding, _ptr(out)))
    return out


def validate_bvh(nodes, n_tri=0, link_encoding=PT_LINKS_FLOAT_VALUE) -> None:
    nodes = _as(nodes, bbox_dtype, "nodes")
    _check(lib.pt_bvh_validate(_ptr(nodes), nodes.shape[0], link_encoding, n_tri))


PT_GLTF_FIRST_MESH_ONLY = 1


class Mesh:
    """Result of ``pt_obj_load`` (or ``pt_gltf_load`` for .gltf / .glb) copied into numpy arrays."""

    def __init__(self, path: str, normalize: bool = False, gltf_flags: int = 0):
        h = _vp()
        if str(path).lower().endswith((".gltf", ".glb")):
            _check(lib.pt_gltf_load(os.fsencode(path), gltf_flags, C.byref(h)))
        else:
            _check(lib.pt_obj_load(os.fsencode(path), C.byref(h)))
        try:
            if normalize:
                _check(lib.pt_mesh_normalize_unit_cube(h))
            nv, nt, nm = (lib.pt_mesh_vertex_count(h), lib.pt_mesh_triangle_count(h),
                          lib.pt_mesh_material_count(h))

            def grab(ptr, nbytes, dtype):
                return np.frombuffer((C.c_char * nbytes).from_address(ptr), dtype=dtype).copy()

            self.vtx = grab(lib.pt_mesh_vertices(h), nv * 12, np.float32).reshape(-1, 3)
            self.idx = grab(lib.pt_mesh_indices(h), nt * 12, np.int32).reshape(-1, 3)
            self.material_ids = grab(lib.pt_mesh_material_ids(h), nt * 4, np.int32)
            self.materials = grab(lib.pt_mesh_materials(h), nm * 48, material_dtype)
            # OBJ files: each triangle's shape (o / g group) and its index within the shape
            self.n_shapes = int(lib.pt_mesh_shape_count(h))
            sp, pp = lib.pt_mesh_shape_ids(h), lib.pt_mesh_shape_prim_ids(h)
            self.shape_ids = grab(sp, nt * 4, np.int32) if sp else None
            self.shape_prim_ids = grab(pp, nt * 4, np.int32) if pp else None
        finally:
            lib.pt_mesh_destroy(h)


# ---------------------------------------------------------------- device side
class Model:
    """``pt_gltf_load_model``: .meshes = [(vtx, idx, material_ids)], .transforms [n,3,4],
    .mesh_of, .materials -- the arguments of ``Scene.instanced`` (trees left to the library)."""

    def __init__(self, path: str):
        h = _vp()
        _check(lib.pt_gltf_load_model(path.encode(), C.byref(h)))
        try:
            self.meshes = []
            for k in range(lib.pt_model_mesh_count(h)):
                m = lib.pt_model_mesh(h, k)
                nv, nt = lib.pt_mesh_vertex_count(m), lib.pt_mesh_triangle_count(m)
                vtx = np.ctypeslib.as_array(C.cast(lib.pt_mesh_vertices(m), C.POINTER(C.c_float)), (nv, 3)).copy()
                idx = np.ctypeslib.as_array(C.cast(lib.pt_mesh_indices(m), C.POINTER(C.c_int32)), (nt, 3)).copy()
                mids = np.ctypeslib.as_array(C.cast(lib.pt_mesh_material_ids(m), C.POINTER(C.c_int32)), (nt,)).copy()
                self.meshes.append((vtx, idx, mids))
            n = lib.pt_model_instance_count(h)
            inst = lib.pt_model_instances(h)
            self.transforms = np.array([[inst[i].object_to_world[j] for j in range(12)] for i in range(n)],
                                       dtype=np.float32).reshape(n, 3, 4)
            self.mesh_of = [int(inst[i].mesh) for i in range(n)]
            nm = lib.pt_model_material_count(h)
            self.materials = np.frombuffer(
                C.string_at(lib.pt_model_materials(h), nm * material_dtype.itemsize), dtype=material_dtype).copy()
        finally:
            lib.pt_model_destroy(h)

    def scene(self, ctx: "Context", link_encoding: int = PT_LINKS_FLOAT_VALUE) -> "Scene":
        return Scene.instanced(ctx, [(None, v, i, m) for v, i, m in self.meshes], self.transforms,
                               self.mesh_of, None, materials=self.materials, link_encoding=link_encoding)


class Context:
    def __init__(self, device: int = 0):
        self._h = _vp()
        _check(lib.pt_context_create(device, C.byref(self._h)))
        self.device = device

    def set_stream(self, hip_stream: Optional[int]) -> None:
        _check(lib.pt_context_set_stream(self._h, hip_stream))

    @property
    def stream(self) -> int:
        return lib.pt_context_stream(self._h) or 0

    def synchronize(self) -> None:
        _check(lib.pt_context_synchronize(self._h))

    def set_option(self, key: str, value) -> None:
        _check(lib.pt_context_set_option(self._h, key.encode(), str(int(value)).encode()))

    def set_kernel_timing(self, on: bool) -> None:
        _check(lib.pt_context_set_kernel_timing(self._h, int(on)))

    def last_trace_ms(self) -> float:
        """Device time of the last trace call's traversal kernel (kernel timing must be on)."""
        out = C.c_double()
        _check(lib.pt_context_last_trace_ms(self._h, C.byref(out)))
        return out.value

    def measure_gather(self, table_bytes: int, record_bytes: int = 64, iterations: int = 400) -> float:
        """``pt_measure_gather``: dependent random gathers per second from a table of that size."""
        out = C.c_double()
        _check(lib.pt_measure_gather(self._h, table_bytes, record_bytes, iterations, C.byref(out)))
        return out.value

    def close(self) -> None:
        """pt_context_destroy; raises while a Comm of this context is still open."""
        if self._h:
            _check(lib.pt_context_destroy(self._h))
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- host-buffer entry points
    def raygen(self, cam: Camera, width: int, height: int) -> np.ndarray:
        rays = np.empty(width * height, dtype=ray_dtype)
        _check(lib.pt_raygen(self._h, C.byref(cam), width, height, _ptr(rays)))
        return rays

    def depth_shade(self, hits) -> np.ndarray:
        hits = _as(hits, hit_dtype, "hits")
        out = np.empty((hits.shape[0], 4), dtype=np.float32)
        _check(lib.pt_depth_shade(self._h, _ptr(hits), hits.shape[0], _ptr(out)))
        return out


class Scene:
    """nodes given: ``pt_scene_create`` (any skip-link array).  nodes=None: ``pt_scene_build``,
    the BVH is built on the GPU from the mesh."""

    def __init__(self, ctx: Context, nodes, vtx, idx, materials=None, material_ids=None,
                 link_encoding: int = PT_LINKS_FLOAT_VALUE):
        self.ctx = ctx
        if nodes is not None:
            nodes = _as(nodes, bbox_dtype, "nodes")
        vtx = _as(vtx, np.float32, "vtx").reshape(-1, 3)
        idx = _as(idx, np.int32, "idx").reshape(-1, 3)
        mats = mids = None
        if materials is not None:
            mats = _as(materials, material_dtype, "materials")
            mids = _as(material_ids, np.int32, "material_ids")
            if mids.shape[0] != idx.shape[0]:
                raise ValueError("material_ids must have one entry per triangle")
        self.n_tri = idx.shape[0]
        self._h = _vp()
        if nodes is None:
            _check(lib.pt_scene_build(ctx._h, _ptr(vtx), vtx.shape[0], _ptr(idx), idx.shape[0],
                                      _ptr(mats) if mats is not None else None,
                                      mats.shape[0] if mats is not None else 0,
                                      _ptr(mids) if mids is not None else None, link_encoding,
                                      C.byref(self._h)))
            return
        _check(lib.pt_scene_create(ctx._h, _ptr(nodes), nodes.shape[0], link_encoding, _ptr(vtx),
                                   vtx.shape[0], _ptr(idx), idx.shape[0],
                                   _ptr(mats) if mats is not None else None,
                                   mats.shape[0] if mats is not None else 0,
                                   _ptr(mids) if mids is not None else None, C.byref(self._h)))

    @classmethod
    def instanced(cls, ctx: Context, meshes, transforms, mesh_of, top_nodes, materials=None,
                  material_offsets=None, link_encoding: int = PT_LINKS_FLOAT_VALUE) -> "Scene":
        """``pt_scene_create_instanced``.  meshes: [(nodes, vtx, idx, material_ids or None)];
        transforms: [n, 3, 4] object-to-world; mesh_of: mesh index per instance; top_nodes:
        skip-link tree over ``instance_bounds(...)``.  A mesh whose nodes are None is built on
        the GPU; top_nodes None: the top level is built by the library."""
        descs, insts, keep = _instance_args(meshes, transforms, mesh_of, material_offsets)
        top = _as(top_nodes, bbox_dtype, "top_nodes") if top_nodes is not None else None
        mats = _as(materials, material_dtype, "materials") if materials is not None else None
        self = cls.__new__(cls)
        self.ctx = ctx
        self._h = _vp()
        _check(lib.pt_scene_create_instanced(
            ctx._h, descs, len(descs), insts, len(insts), _ptr(top) if top is not None else None,
            top.shape[0] if top is not None else 0, link_encoding,
            _ptr(mats) if mats is not None else None, mats.shape[0] if mats is not None else 0,
            C.byref(self._h)))
        self.n_tri = sum(int(descs[m].n_triangles) for m in mesh_of)
        return self

    def instance_count(self) -> int:
        return int(lib.pt_scene_instance_count(self._h))

    def read_top_nodes(self) -> np.ndarray:
        out = np.empty(lib.pt_scene_top_node_count(self._h), dtype=bbox_dtype)
        _check(lib.pt_scene_read_top_nodes(self._h, _ptr(out)))
        return out

    def read_mesh_nodes(self, mesh: int) -> np.ndarray:
        out = np.empty(lib.pt_scene_mesh_node_count(self._h, mesh), dtype=bbox_dtype)
        _check(lib.pt_scene_read_mesh_nodes(self._h, mesh, _ptr(out)))
        return out

    def tree_depth(self) -> int:
        return int(lib.pt_scene_tree_depth(self._h))

    def set_face_ids(self, shape_ids, prim_ids) -> None:
        """pt_scene_set_face_ids: what the closest-hit queries report per triangle (None, None
        removes the table)."""
        if shape_ids is None and prim_ids is None:
            _check(lib.pt_scene_set_face_ids(self._h, None, None, 0))
            return
        a = np.ascontiguousarray(shape_ids, dtype=np.int32)
        b = np.ascontiguousarray(prim_ids, dtype=np.int32)
        if a.shape != b.shape or a.ndim != 1:
            raise ValueError("shape_ids and prim_ids must be one-dimensional and of equal length")
        _check(lib.pt_scene_set_face_ids(self._h, _ptr(a), _ptr(b), a.shape[0]))

    def triangle_count(self) -> int:
        return int(lib.pt_scene_triangle_count(self._h))

    def read_nodes(self) -> np.ndarray:
        """The scene's skip-link node array."""
        out = np.empty(lib.pt_scene_node_count(self._h), dtype=bbox_dtype)
        _check(lib.pt_scene_read_nodes(self._h, _ptr(out)))
        return out

    def close(self) -> None:
        if self._h:
            lib.pt_scene_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def trace_closest(self, rays, flags: int = PT_TRACE_DEFAULT, stats: bool = False):
        rays = _as(rays, ray_dtype, "rays")
        hits = np.empty(rays.shape[0], dtype=hit_dtype)
        st = TraceStats()
        _check(lib.pt_trace_closest(self.ctx._h, self._h, _ptr(rays), rays.shape[0], _ptr(hits),
                                    flags, C.byref(st) if stats else None))
        return (hits, st) if stats else hits

    def trace_any(self, rays) -> np.ndarray:
        rays = _as(rays, ray_dtype, "rays")
        out = np.empty(rays.shape[0], dtype=np.int32)
        _check(lib.pt_trace_any(self.ctx._h, self._h, _ptr(rays), rays.shape[0], _ptr(out)))
        return out

    # -- device-pointer entry points (pointers are plain ints, e.g. torch .data_ptr())
    def trace_closest_device(self, rays_dev: int, n: int, hits_dev: int,
                             flags: int = PT_TRACE_DEFAULT) -> None:
        _check(lib.pt_trace_closest_device(self.ctx._h, self._h, rays_dev, n, hits_dev, flags))

    def trace_closest_device_static(self, rays_dev: int, n: int, hits_dev: int,
                                    flags: int = PT_TRACE_DEFAULT) -> None:
        """``pt_trace_closest_device`` without any device allocation (static scratch)."""
        _check(lib.pt_trace_closest_device_static(self.ctx._h, self._h, rays_dev, n, hits_dev, flags))

    def trace_closest_device_counted(self, rays_dev: int, n: int, hits_dev: int,
                                     flags: int = PT_TRACE_DEFAULT) -> TraceStats:
        st = TraceStats()
        _check(lib.pt_trace_closest_device_counted(self.ctx._h, self._h, rays_dev, n, hits_dev,
                                                   flags, C.byref(st)))
        return st

    def trace_any_device(self, rays_dev: int, n: int, out_dev: int) -> None:
        _check(lib.pt_trace_any_device(self.ctx._h, self._h, rays_dev, n, out_dev))


def _instance_args(meshes, transforms, mesh_of, material_offsets=None):
    keep = []
    descs = (MeshDesc * len(meshes))()
    for k, mesh in enumerate(meshes):
        nodes, vtx, idx = mesh[0], mesh[1], mesh[2]
        mids = mesh[3] if len(mesh) > 3 else None
        if nodes is not None:
            nodes = _as(nodes, bbox_dtype, "nodes")
        vtx = _as(vtx, np.float32, "vtx").reshape(-1, 3)
        idx = _as(idx, np.int32, "idx").reshape(-1, 3)
        keep += [nodes, vtx, idx]
        descs[k].nodes = _ptr(nodes) if nodes is not None else None
        descs[k].vertices, descs[k].indices = _ptr(vtx), _ptr(idx)
        descs[k].n_nodes = nodes.shape[0] if nodes is not None else 0
        descs[k].n_vertices, descs[k].n_triangles = vtx.shape[0], idx.shape[0]
        if mids is not None:
            mids = _as(mids, np.int32, "material_ids")
            if mids.shape[0] != idx.shape[0]:
                raise ValueError("material_ids must have one entry per triangle")
            keep.append(mids)
            descs[k].material_ids = _ptr(mids)
    tr = _as(transforms, np.float32, "transforms").reshape(-1, 12)
    if tr.shape[0] != len(mesh_of):
        raise ValueError("one transform per instance")
    insts = (Instance * len(mesh_of))()
    for k, m in enumerate(mesh_of):
        for j in range(12):
            insts[k].object_to_world[j] = float(tr[k, j])
        insts[k].mesh = int(m)
        insts[k].material_offset = int(material_offsets[k]) if material_offsets is not None else 0
    descs._keep = keep
    return descs, insts, keep


def instance_bounds(meshes, transforms, mesh_of) -> np.ndarray:
    """``pt_instance_bounds``: world-space leaf boxes for the top-level ``build_bvh``."""
    descs, insts, _keep = _instance_args(meshes, transforms, mesh_of)
    out = np.zeros(len(insts), dtype=bbox_dtype)
    _check(lib.pt_instance_bounds(descs, len(descs), insts, len(insts), _ptr(out)))
    return out


def shard_layout(width: int, height: int, tile_size: int, shard_index: int, shard_count: int):
    """Pixel (x, y) of every slot of a shard's packed tile buffer, (-1, -1) outside the image."""
    n = C.c_uint64()
    _check(lib.pt_shard_layout(width, height, tile_size, shard_index, shard_count, C.byref(n), None))
    xy = np.empty((n.value, 2), dtype=np.int32)
    _check(lib.pt_shard_layout(width, height, tile_size, shard_index, shard_count, C.byref(n),
                               _ptr(xy)))
    return xy


def render_params(width: int, height: int, **kw) -> RenderParams:
    p = RenderParams()
    lib.pt_render_params_default(C.byref(p), width, height)
    for k, v in kw.items():
        if k == "background":
            for i in range(3):
                p.background[i] = float(v[i])
        else:
            if not hasattr(p, k):
                raise AttributeError(f"pt_render_params has no field {k!r}")
            setattr(p, k, v)
    return p


class Film:
    """width x height float4 accumulation buffer on the context's GPU.

    ``device_ptr``: wrap caller-owned device memory (e.g. ``tensor.data_ptr()`` of a
    float32 CUDA tensor of shape [H, W, 4]) instead of allocating."""

    def __init__(self, ctx: Context, width: int, height: int, device_ptr: Optional[int] = None):
        self.ctx, self.width, self.height = ctx, width, height
        self._h = _vp()
        if device_ptr is None:
            _check(lib.pt_film_create(ctx._h, width, height, C.byref(self._h)))
        else:
            _check(lib.pt_film_wrap(ctx._h, width, height, device_ptr, C.byref(self._h)))

    def clear(self) -> None:
        _check(lib.pt_film_clear(self._h))

    @property
    def device_ptr(self) -> int:
        return lib.pt_film_device_ptr(self._h) or 0

    def read(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        _check(lib.pt_film_read(self._h, _ptr(out)))
        return out

    def save(self, path: str) -> None:
        """Accumulation checkpoint (sums and sample counts, exact)."""
        _check(lib.pt_film_save(self._h, path.encode()))

    def load(self, path: str) -> None:
        _check(lib.pt_film_load(self._h, path.encode()))

    def write_image(self, path: str, fmt: int = PT_IMAGE_PPM, exposure: float = 1.0) -> None:
        _check(lib.pt_film_write_image(self._h, path.encode(), fmt, exposure))

    def packed_size(self, tile_size: int, shard_index: int, shard_count: int) -> int:
        return int(lib.pt_film_packed_size(self._h, tile_size, shard_index, shard_count))

    def pack_tiles(self, tile_size, shard_index, shard_count, packed_dev: int) -> None:
        _check(lib.pt_film_pack_tiles(self._h, tile_size, shard_index, shard_count, packed_dev))

    def unpack_tiles(self, tile_size, shard_index, shard_count, packed_dev: int) -> None:
        _check(lib.pt_film_unpack_tiles(self._h, tile_size, shard_index, shard_count, packed_dev))

    def close(self) -> None:
        if self._h:
            lib.pt_film_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def comm_available() -> bool:
    return bool(lib.pt_comm_available())


def comm_unique_id() -> bytes:
    """``pt_comm_unique_id``: 128 bytes to hand to every rank."""
    buf = C.create_string_buffer(128)
    _check(lib.pt_comm_unique_id(buf))
    return buf.raw


class Comm:
    """``pt_comm_create``: this rank's end of an RCCL communicator on the context's GPU."""

    def __init__(self, ctx: Context, unique_id: bytes, n_ranks: int, rank: int):
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of comm_unique_id()")
        self.ctx = ctx
        self._h = _vp()
        _check(lib.pt_comm_create(ctx._h, unique_id, n_ranks, rank, C.byref(self._h)))

    @property
    def rank(self) -> int:
        return lib.pt_comm_rank(self._h)

    @property
    def size(self) -> int:
        return lib.pt_comm_size(self._h)

    def reduce_film(self, film: "Film", root: int = 0) -> None:
        _check(lib.pt_film_reduce(film._h, self._h, root))

    def gather_tiles(self, film: "Film", tile_size: int = 64, root: int = 0) -> None:
        _check(lib.pt_film_gather_tiles(film._h, self._h, tile_size, root))

    def close(self) -> None:
        if self._h:
            lib.pt_comm_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def render(ctx: Context, scene: "Scene", cam: Camera, params: RenderParams, film: Film,
           stats: bool = False):
    st = RenderStats()
    _check(lib.pt_render(ctx._h, scene._h, C.byref(cam), C.byref(params), film._h,
                         C.byref(st) if stats else None))
    return st if stats else None


# ---------------------------------------------------------------- camera helpers
def _normalize(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v)


def look_at_camera(eye, target, up=(0.0, 1.0, 0.0), fovy_deg=40.0, aspect=1.0,
                   mode=PT_CAMERA_PINHOLE) -> Camera:
    """Camera looking from ``eye`` to ``target``.

    c0, c1, c2 are right, up and back (the columns of mat3(inv_view)).  In
    pinhole mode sx, sy are tan(fov/2); in reference mode they are the
    inverse-projection diagonal the shader multiplies by (same numbers).
    """
    eye = np.asarray(eye, dtype=np.float64)
    back = _normalize(eye - np.asarray(target, dtype=np.float64))
    right = _normalize(np.cross(np.asarray(up, dtype=np.float64), back))
    upv = np.cross(back, right)
    cam = Camera()
    t = float(np.tan(np.radians(fovy_deg) * 0.5))
    for k in range(3):
        cam.eye[k] = float(eye[k])
        cam.c0[k] = float(right[k])
        cam.c1[k] = float(upv[k])
        cam.c2[k] = float(back[k])
    cam.sx = t * aspect
    cam.sy = t
    cam.mode = mode
    return cam


def _floats(values, n):
    if values is None:
        return None
    a = np.asarray(values, dtype=np.float32).reshape(-1)
    if a.size != n:
        raise ValueError("expected %d floats" % n)
    return (C.c_float * n)(*a.tolist())


def camera_from_matrices(inv_view, inv_projection, camera_position=None, mode=PT_CAMERA_PINHOLE) -> Camera:
    """pt_camera_from_matrices: 4x4 matrices as 16 floats, column-major (a numpy array laid out
    as GLSL's mat4, i.e. m[column][row])."""
    cam = Camera()
    _check(lib.pt_camera_from_matrices(_floats(inv_view, 16), _floats(inv_projection, 16),
                                       _floats(camera_position, 3), mode, C.byref(cam)))
    return cam


def camera_from_orientation(orientation_xyzw, eye_pos, fovy_radians, viewport, mode=PT_CAMERA_PINHOLE) -> Camera:
    """pt_camera_from_orientation: the reference's Camera struct (src/camera.h:11-24)."""
    cam = Camera()
    _check(lib.pt_camera_from_orientation(_floats(orientation_xyzw, 4), _floats(eye_pos, 3), float(fovy_radians),
                                          int(viewport[0]), int(viewport[1]), mode, C.byref(cam)))
    return cam


def camera_look_at(eye, target, up=None, fovy_radians=0.7, aspect=1.0, mode=PT_CAMERA_PINHOLE) -> Camera:
    """pt_camera_look_at (the library's own; look_at_camera above is the binding's numpy form)."""
    cam = Camera()
    _check(lib.pt_camera_look_at(_floats(eye, 3), _floats(target, 3), _floats(up, 3), float(fovy_radians),
                                 float(aspect), mode, C.byref(cam)))
    return cam
