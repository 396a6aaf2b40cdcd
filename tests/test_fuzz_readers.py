"""The file readers against damaged input (tools/fuzz_readers.py, a few hundred mutations per
format): OBJ + MTL, glTF / GLB in both loading modes, the BVH cache.  A reader may refuse a file;
it may not crash, and what it returns must be consistent (indices inside the vertex array, one
material id per triangle, instances naming existing meshes)."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def test_damaged_files_are_refused_or_read_consistently():
    tool = os.path.join(HERE, "..", "tools", "fuzz_readers.py")
    out = subprocess.run([sys.executable, tool, "250", "11"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert [r["kind"] for r in rows] == ["obj", "glb", "gltf", "bvh"]
    for r in rows:
        assert r["loaded"] + r["refused"] == 250
    assert rows[0]["loaded"] > 20 and rows[0]["refused"] > 20       # both outcomes are exercised
    assert rows[3]["loaded"] <= 3          # the cache file is checksummed (a mutation can be a no-op)
