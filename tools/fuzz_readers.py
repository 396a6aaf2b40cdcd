"""Mutation fuzzing of the file readers (pt_obj_load, pt_gltf_load, pt_gltf_load_model, pt_bvh_load): host
code only, no GPU.  Valid files are damaged at random (bytes flipped, ranges cut out, duplicated
or replaced by numbers that tempt overflows) and loaded; a reader may refuse a file with an error
code, it may not crash, hang or hand out arrays that do not fit each other.  Runs in a child
process per batch so that a crash is reported, not suffered.
    python tools/fuzz_readers.py [iterations per kind] [seed] [gpu: also the film checkpoint]"""
import json
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "gpu-pathtracer_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))


def mutate(data: bytes, rng) -> bytes:
    b = bytearray(data)
    for _ in range(rng.integers(1, 4)):
        k = rng.integers(0, 6)
        pos = int(rng.integers(0, max(1, len(b))))
        if k == 0 and b:
            b[pos] = int(rng.integers(0, 256))
        elif k == 1:
            del b[pos:pos + int(rng.integers(1, 40))]
        elif k == 2:
            b[pos:pos] = b[pos:pos + int(rng.integers(1, 40))]
        elif k == 3:
            b[pos:pos] = rng.choice([b"4294967295", b"-1", b"1e39", b"nan", b"99999999999999999999", b"0", b"{", b"]", b"\"", b"/"])
        elif k == 4 and b:
            b[pos] ^= 1 << int(rng.integers(0, 8))
        else:
            del b[pos:]
    return bytes(b)


def child(kind: str, iterations: int, seed: int) -> None:
    import numpy as np
    import mipt
    rng = np.random.default_rng(seed)
    tmp = tempfile.mkdtemp()
    if kind == "bvh":
        # the BVH cache file: a damaged one must be refused or rebuilt from, never trusted blindly
        from mipt import scenes
        vtx, idx, _, _ = scenes.random_soup(300, seed=3)
        lb = mipt.mesh_leaf_bounds(vtx, idx)
        path = os.path.join(tmp, "tree.bvh")
        good, _ = mipt.build_bvh_cached(lb, path)
        src = open(path, "rb").read()
        ok = refused = 0
        for it in range(iterations):
            open(path, "wb").write(mutate(src, rng))
            try:
                nodes, cached = mipt.build_bvh_cached(lb, path)
                mipt.validate_bvh(nodes, idx.shape[0], mipt.PT_LINKS_FLOAT_VALUE)
                ok += 1 if cached else 0
                refused += 0 if cached else 1
            except mipt.MiptError:
                refused += 1
        print(json.dumps({"kind": kind, "iterations": iterations, "loaded": ok, "refused": refused}), flush=True)
        return
    if kind == "film":
        # the accumulation checkpoint (needs a GPU: the film lives on the device)
        ctx = mipt.Context(0)
        film = mipt.Film(ctx, 24, 16)
        path = os.path.join(tmp, "film.ckpt")
        film.save(path)
        src = open(path, "rb").read()
        ok = refused = 0
        for it in range(iterations):
            open(path, "wb").write(mutate(src, rng))
            try:
                film.load(path)
                assert film.read().shape == (16, 24, 4)
                ok += 1
            except mipt.MiptError:
                refused += 1
        print(json.dumps({"kind": kind, "iterations": iterations, "loaded": ok, "refused": refused}), flush=True)
        return
    if kind == "obj":
        src = open(os.path.join(REPO, "tests", "golden", "cornell_box.obj"), "rb").read()
        mtl = open(os.path.join(REPO, "tests", "golden", "cornell_box.mtl"), "rb").read()
        path = os.path.join(tmp, "f.obj")
    else:
        import pathlib
        import test_gltf
        p, *_ = test_gltf._document(pathlib.Path(tmp), "glb" if kind == "glb" else "base64")
        src = open(p, "rb").read()
        path = p
    ok = refused = 0
    for it in range(iterations):
        data = mutate(src, rng)
        open(path, "wb").write(data)
        if kind == "obj":
            open(os.path.join(tmp, "cornell_box.mtl"), "wb").write(mutate(mtl, rng) if it % 3 == 0 else mtl)
        try:
            if kind != "obj" and it % 2:
                m = mipt.Model(path)
                for v, i, mid in m.meshes:
                    assert i.size == 0 or (0 <= i.min() and i.max() < v.shape[0]), "index outside the vertices"
                    assert mid.shape[0] == i.shape[0]
                assert len(m.mesh_of) == m.transforms.shape[0]
                assert all(0 <= k < len(m.meshes) for k in m.mesh_of)
            else:
                m = mipt.Mesh(path)
                assert m.idx.size == 0 or (0 <= m.idx.min() and m.idx.max() < m.vtx.shape[0]), "index outside the vertices"
                assert m.material_ids.shape[0] == m.idx.shape[0]
                assert m.material_ids.size == 0 or (0 <= m.material_ids.min() and m.material_ids.max() < max(1, m.materials.shape[0]))
            ok += 1
        except mipt.MiptError:
            refused += 1
    print(json.dumps({"kind": kind, "iterations": iterations, "loaded": ok, "refused": refused}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
        sys.exit(0)
    iterations = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    failed = 0
    kinds = ("obj", "glb", "gltf", "bvh") + (("film",) if len(sys.argv) > 3 and sys.argv[3] == "gpu" else ())
    for kind in kinds:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", kind, str(iterations), str(seed)],
                           capture_output=True, text=True, timeout=1200)
        sys.stdout.write(r.stdout)
        if r.returncode != 0:
            failed += 1
            print(json.dumps({"kind": kind, "crashed_with": r.returncode, "stderr": r.stderr[-600:]}))
    sys.exit(1 if failed else 0)
