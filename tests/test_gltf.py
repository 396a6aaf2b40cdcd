"""The glTF 2.0 reader (pt_gltf_load): the live application's scene format
(src/main_granite.cpp:265-268).  Files are written here, byte by byte, from known arrays."""
import base64
import json
import struct

import numpy as np
import pytest


def _document(tmp_path, kind):
    """Two meshes: an indexed quad (u16) under a translated, scaled node, and a non-indexed
    triangle in a child node rotated 90 degrees about z; interleaved positions for the quad."""
    quad = np.float32([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]])
    quad_idx = np.uint16([0, 1, 2, 0, 2, 3])
    tri = np.float32([[1, 0, 0], [2, 0, 0], [1, 0, 1]])
    inter = np.zeros((4, 5), dtype=np.float32)          # stride 20: xyz + 2 floats of padding
    inter[:, :3] = quad
    blob = inter.tobytes() + quad_idx.tobytes() + tri.tobytes()
    o_idx, o_tri = inter.nbytes, inter.nbytes + quad_idx.nbytes
    doc = {
        "asset": {"version": "2.0"},
        "scene": 0,
        "scenes": [{"nodes": [0]}],
        "nodes": [
            {"mesh": 0, "translation": [10, 0, 0], "scale": [2, 2, 2], "children": [1]},
            {"mesh": 1, "rotation": [0, 0, 0.7071067811865476, 0.7071067811865476]},
        ],
        "meshes": [
            {"primitives": [{"attributes": {"POSITION": 0}, "indices": 1, "material": 0}]},
            {"primitives": [{"attributes": {"POSITION": 2}, "material": 1},
                            {"attributes": {"POSITION": 2}, "mode": 1}]},      # lines: skipped
        ],
        "materials": [
            {"pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.4, 0.2, 1], "metallicFactor": 0.0,
                                      "roughnessFactor": 0.5}},
            {"pbrMetallicRoughness": {"baseColorFactor": [1, 1, 1, 1], "metallicFactor": 1.0},
             "emissiveFactor": [1, 0.5, 0.25],
             "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 4.0}}},
        ],
        "buffers": [{"byteLength": len(blob)}],
        "bufferViews": [
            {"buffer": 0, "byteOffset": 0, "byteLength": inter.nbytes, "byteStride": 20},
            {"buffer": 0, "byteOffset": o_idx, "byteLength": quad_idx.nbytes},
            {"buffer": 0, "byteOffset": o_tri, "byteLength": tri.nbytes},
        ],
        "accessors": [
            {"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3"},
            {"bufferView": 1, "componentType": 5123, "count": 6, "type": "SCALAR"},
            {"bufferView": 2, "componentType": 5126, "count": 3, "type": "VEC3"},
        ],
    }
    if kind == "glb":
        js = json.dumps(doc).encode()
        js += b" " * (-len(js) % 4)
        bn = blob + b"\0" * (-len(blob) % 4)
        body = struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(bn), 0x004E4942) + bn
        path = tmp_path / "scene.glb"
        path.write_bytes(struct.pack("<III", 0x46546C67, 2, 12 + len(body)) + body)
    elif kind == "external":
        (tmp_path / "scene.bin").write_bytes(blob)
        doc["buffers"][0]["uri"] = "scene.bin"
        path = tmp_path / "scene.gltf"
        path.write_text(json.dumps(doc, indent=1))
    else:
        doc["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(blob).decode()
        path = tmp_path / "scene.gltf"
        path.write_text(json.dumps(doc))
    return str(path), quad, quad_idx, tri


@pytest.mark.parametrize("kind", ["glb", "external", "base64"])
def test_gltf_scene_is_flattened_to_world_space(mipt, tmp_path, kind):
    path, quad, quad_idx, tri = _document(tmp_path, kind)
    m = mipt.Mesh(path)
    assert m.vtx.shape == (7, 3) and m.idx.shape == (3, 3)
    np.testing.assert_allclose(m.vtx[:4], quad * 2 + [10, 0, 0], rtol=0, atol=1e-6)
    # child: parent (translate, scale 2) x rotation by 90 degrees about z: (x, y, z) -> (-y, x, z)
    rot = np.stack([-tri[:, 1], tri[:, 0], tri[:, 2]], axis=1)
    np.testing.assert_allclose(m.vtx[4:], rot * 2 + [10, 0, 0], rtol=0, atol=1e-5)
    np.testing.assert_array_equal(m.idx, [[0, 1, 2], [0, 2, 3], [4, 5, 6]])
    np.testing.assert_array_equal(m.material_ids, [0, 0, 1])
    a, b = m.materials[0], m.materials[1]
    np.testing.assert_allclose(a["kd"], [0.8, 0.4, 0.2])
    np.testing.assert_allclose(a["ks"], [0.04] * 3)
    assert a["roughness"] == pytest.approx(0.5) and not a["ke"].any()
    np.testing.assert_allclose(b["kd"], [0, 0, 0])
    np.testing.assert_allclose(b["ks"], [1, 1, 1])
    np.testing.assert_allclose(b["ke"], [4, 2, 1])


def test_gltf_first_mesh_only_is_what_the_reference_uses(mipt, tmp_path):
    """src/main_granite.cpp:286-298: the first mesh, positions as stored."""
    path, quad, quad_idx, _ = _document(tmp_path, "glb")
    m = mipt.Mesh(path, gltf_flags=mipt.PT_GLTF_FIRST_MESH_ONLY)
    np.testing.assert_array_equal(m.vtx, quad)
    np.testing.assert_array_equal(m.idx.reshape(-1), quad_idx.astype(np.int32))
    # and it goes through the BVH path like any mesh
    nodes = mipt.build_bvh(mipt.mesh_leaf_bounds(m.vtx, m.idx), mipt.bvh_options(**{"bvh.builder": "sah"}))
    mipt.validate_bvh(nodes, 2)


def test_gltf_errors(mipt, tmp_path):
    with pytest.raises(mipt.MiptError, match="cannot open"):
        mipt.Mesh(str(tmp_path / "nope.glb"))
    p = tmp_path / "bad.gltf"
    p.write_text("{\"asset\": {\"version\": \"2.0\"}, \"meshes\": [")
    with pytest.raises(mipt.MiptError, match="JSON"):
        mipt.Mesh(str(p))
    p.write_text(json.dumps({"asset": {"version": "2.0"}, "scenes": [{"nodes": []}]}))
    with pytest.raises(mipt.MiptError, match="no triangles"):
        mipt.Mesh(str(p))
    path, *_ = _document(tmp_path, "external")
    doc = json.loads(open(path).read())
    doc["accessors"][1]["count"] = 600                      # indices beyond the buffer
    open(path, "w").write(json.dumps(doc))
    with pytest.raises(mipt.MiptError, match="beyond its buffer"):
        mipt.Mesh(path)
    doc["accessors"][1]["count"] = 6
    doc["accessors"][0]["count"] = 2                        # index 2, 3 outside the vertices
    open(path, "w").write(json.dumps(doc))
    with pytest.raises(mipt.MiptError, match="outside the primitive"):
        mipt.Mesh(path)
    raw = open(str(tmp_path / "scene.bin"), "rb").read()
    open(str(tmp_path / "scene.bin"), "wb").write(raw[:10])
    doc["accessors"][0]["count"] = 4
    open(path, "w").write(json.dumps(doc))
    with pytest.raises(mipt.MiptError, match="shorter than"):
        mipt.Mesh(path)


@pytest.mark.parametrize("field,where,value,message", [
    # sizes chosen so that count * stride, or the sum of the offsets, wraps 64-bit arithmetic
    ("count", ("accessors", 1), 2 ** 62, "non-negative integer"),       # beyond what a double holds exactly
    ("count", ("accessors", 1), 2 ** 52, "beyond its buffer"),
    ("count", ("accessors", 1), 2 ** 63 + 3, "non-negative integer"),
    ("count", ("accessors", 0), 2 ** 52 + 1, "beyond its buffer"),
    ("count", ("accessors", 1), -6, "non-negative integer"),
    ("count", ("accessors", 1), 5.5, "non-negative integer"),
    ("byteOffset", ("accessors", 1), 2 ** 64 - 4, "non-negative integer"),
    ("byteOffset", ("accessors", 1), 2 ** 53, "beyond its buffer"),
    ("byteOffset", ("accessors", 1), -2, "non-negative integer"),
    ("byteOffset", ("bufferViews", 1), 2 ** 53, "beyond its buffer"),
    ("byteOffset", ("bufferViews", 1), -80, "non-negative integer"),
    ("byteLength", ("bufferViews", 1), 2 ** 53, "beyond its buffer"),
    ("byteStride", ("bufferViews", 0), 2 ** 52, "beyond its buffer"),
    ("byteStride", ("bufferViews", 0), -20, "non-negative integer"),
    # the accessor must stay inside its bufferView even where the buffer goes on
    ("count", ("accessors", 1), 7, "beyond its buffer"),
])
def test_gltf_accessor_sizes_cannot_overflow(mipt, tmp_path, field, where, value, message):
    path, *_ = _document(tmp_path, "external")
    doc = json.loads(open(path).read())
    doc[where[0]][where[1]][field] = value
    open(path, "w").write(json.dumps(doc))
    with pytest.raises(mipt.MiptError, match=message):
        mipt.Mesh(path)


def test_gltf_count_times_stride_cannot_wrap(mipt, tmp_path):
    """count and byteStride each acceptable, their product beyond 2^64."""
    path, *_ = _document(tmp_path, "external")
    doc = json.loads(open(path).read())
    doc["accessors"][0]["count"] = 2 ** 40 + 1
    doc["bufferViews"][0]["byteStride"] = 2 ** 24 * 20        # (count - 1) * stride = 20 * 2^64
    open(path, "w").write(json.dumps(doc))
    with pytest.raises(mipt.MiptError, match="beyond its buffer"):
        mipt.Mesh(path)


def _instanced_document(tmp_path):
    """One quad mesh shown by three nodes (one of them nested), one triangle mesh shown once,
    and a mesh no node shows."""
    path, quad, quad_idx, tri = _document(tmp_path, "external")
    doc = json.loads(open(path).read())
    doc["nodes"] = [
        {"mesh": 0, "translation": [10, 0, 0], "scale": [2, 2, 2], "children": [1, 2]},
        {"mesh": 1, "rotation": [0, 0, 0.7071067811865476, 0.7071067811865476]},
        {"mesh": 0, "translation": [0, 3, 0]},
        {"mesh": 0, "matrix": [1, 0, 0, 0, 0, 0, 1, 0, 0, -1, 0, 0, -4, 1, 2, 1]},
    ]
    doc["scenes"] = [{"nodes": [0, 3]}]
    doc["meshes"].append({"primitives": [{"attributes": {"POSITION": 2}}]})      # unused
    open(path, "w").write(json.dumps(doc))
    return path, quad, quad_idx, tri


def test_gltf_model_keeps_meshes_and_instances(mipt, oracle, tmp_path):
    """pt_gltf_load_model: every shown mesh once in its own space, one instance per showing
    node; placing the meshes by the instances gives the flattened load's vertices."""
    path, quad, quad_idx, tri = _instanced_document(tmp_path)
    model = mipt.Model(path)
    assert len(model.meshes) == 2 and model.mesh_of == [0, 1, 0, 0]
    np.testing.assert_array_equal(model.meshes[0][0], quad)
    np.testing.assert_array_equal(model.meshes[0][1].reshape(-1), quad_idx.astype(np.int32))
    np.testing.assert_array_equal(model.meshes[1][0], tri)
    np.testing.assert_array_equal(model.meshes[0][2], [0, 0])
    np.testing.assert_array_equal(model.meshes[1][2], [1])
    assert model.materials.shape[0] == 2
    np.testing.assert_allclose(model.transforms[0], [[2, 0, 0, 10], [0, 2, 0, 0], [0, 0, 2, 0]], atol=1e-6)
    np.testing.assert_allclose(model.transforms[2], [[2, 0, 0, 10], [0, 2, 0, 6], [0, 0, 2, 0]], atol=1e-6)
    np.testing.assert_allclose(model.transforms[3], [[1, 0, 0, -4], [0, 0, -1, 1], [0, 1, 0, 2]], atol=1e-6)
    flat = mipt.Mesh(path)
    placed = np.concatenate([oracle.transform_points(model.transforms[k], model.meshes[m][0])
                             for k, m in enumerate(model.mesh_of)])
    np.testing.assert_allclose(placed, flat.vtx, rtol=0, atol=2e-5)
    assert flat.idx.shape[0] == sum(model.meshes[m][1].shape[0] for m in model.mesh_of)


@pytest.mark.gpu
def test_gltf_model_as_a_two_level_scene(mipt, gpu_ctx, tmp_path):
    """The model's scene (trees built by the library) and the flattened scene see the same
    surfaces: same hit distances for rays aimed at them, instance and mesh triangle reported."""
    path, *_ = _instanced_document(tmp_path)
    model = mipt.Model(path)
    two = model.scene(gpu_ctx)
    flat_mesh = mipt.Mesh(path)
    flat = mipt.Scene(gpu_ctx, None, flat_mesh.vtx, flat_mesh.idx, flat_mesh.materials, flat_mesh.material_ids)
    try:
        assert two.instance_count() == 4
        rng = np.random.default_rng(4)
        n = 20000
        tri = rng.integers(0, flat_mesh.idx.shape[0], n)
        w = rng.dirichlet([1, 1, 1], n).astype(np.float32)
        target = (flat_mesh.vtx[flat_mesh.idx[tri]] * w[:, :, None]).sum(1)
        rays = np.zeros(n, dtype=mipt.ray_dtype)
        rays["o"][:, :3] = target + rng.normal(size=(n, 3)).astype(np.float32) * 3
        d = target - rays["o"][:, :3]
        rays["d"][:, :3] = d / np.linalg.norm(d, axis=1, keepdims=True)
        rays["o"][:, 3] = 1e30
        a, b = two.trace_closest(rays), flat.trace_closest(rays)
        both = (a["primid"] >= 0) & (b["primid"] >= 0)
        assert both.mean() > 0.95 and ((a["primid"] >= 0) != (b["primid"] >= 0)).mean() < 0.01
        np.testing.assert_allclose(a["uvwt"][both, 3], b["uvwt"][both, 3], rtol=1e-4, atol=1e-4)
        first = np.cumsum([0] + [model.meshes[m][1].shape[0] for m in model.mesh_of])[:-1]
        glob = first[a["shapeid"][both]] + a["primid"][both]
        assert (glob == b["primid"][both]).mean() > 0.97       # coplanar neighbours may swap
    finally:
        two.close()
        flat.close()


@pytest.mark.gpu
def test_cpp_host_program_renders_the_instanced_model(mipt, gpu_ctx, tmp_path):
    """examples/ptrender.cpp --instanced: glTF model -> pt_scene_create_instanced with every tree
    left to the library -> render; the film equals the Python binding's on the same calls."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    exe = os.path.join(here, "..", "gpu-pathtracer_amd", "ptrender")
    if not os.path.exists(exe):
        pytest.fail("gpu-pathtracer_amd/ptrender is not built (make -C gpu-pathtracer_amd ptrender)")
    path, *_ = _instanced_document(tmp_path)
    W, H = 80, 60
    raw, png = str(tmp_path / "f.raw"), str(tmp_path / "o.png")
    view = ["--eye", "6", "4", "14", "--target", "8", "2", "0", "--fov", "60"]
    out = subprocess.run([exe, path, png, "--instanced", "--size", str(W), str(H), "--spp", "4", "--depth", "3",
                          "--nee", "--seed", "3", "--film", raw] + view, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "2 meshes, 4 instances" in out.stdout
    got = np.fromfile(raw, dtype=np.float32).reshape(H, W, 4)
    two = mipt.Model(path).scene(gpu_ctx, link_encoding=mipt.PT_LINKS_INT_BITS)
    film = mipt.Film(gpu_ctx, W, H)
    try:
        # the program's camera: pt_camera_look_at on binary32 arguments
        cam = mipt.camera_look_at((6.0, 4.0, 14.0), (8.0, 2.0, 0.0), None,
                                  float(np.float32(60.0 * 3.14159265358979323846 / 180.0)),
                                  float(np.float32(W) / np.float32(H)))
        mipt.render(gpu_ctx, two, cam, mipt.render_params(W, H, sample_count=4, max_depth=3, seed=3,
                                                          flags=mipt.PT_RENDER_NEE), film)
        want = film.read()
        np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
        assert want[..., :3].sum() > 0          # the emissive triangle is in view or lights something
    finally:
        film.close()
        two.close()
    assert open(png, "rb").read()[:8] == b"\x89PNG\r\n\x1a\n"
