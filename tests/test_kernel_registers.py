"""The persistent traversal kernels over two-child nodes are launched five workgroups of four
waves per compute unit, i.e. five waves per SIMD, which the 512 vector registers of a SIMD lane
allow only up to 96 registers per lane (allocated in eights).  A change that takes the hot kernels
past that -- typically in a rarely taken path inlined into them -- costs a tenth of the benchmark
without failing anything, so the register counts of the built code objects are checked here (CPU
only: reads the device code embedded in the object files of the in-tree build).  The kernels over
four-child nodes, which run on scenes beyond the caches, are faster with four waves per SIMD and
may take up to 104."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def _kernel_metadata(obj, tmp_path):
    fat, code = str(tmp_path / "fatbin"), str(tmp_path / "code.co")
    subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={code}"], check=True)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", code], check=True, capture_output=True, text=True).stdout
    out = {}
    for block in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        if name:
            out[name.group(1)] = {k: int(v) for k, v in re.findall(r"\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|"
                                                                     r"private_segment_fixed_size):\s+(\d+)", block)}
    return out


@pytest.mark.parametrize("obj,kernel", [("render.hip.o", "k_pt_extend_persistentILb0"),
                                        ("render.hip.o", "k_pt_occlude_persistentILb0"),
                                        ("trace.hip.o", "k_trace_ordered_persistentILb0"),
                                        ("trace.hip.o", "k_trace_any_persistent")])
def test_hot_kernels_keep_five_waves_per_simd(tmp_path, obj, kernel):
    path = os.path.join(REPO, "gpu-pathtracer_amd", "build", "csrc", obj)
    if not os.path.exists(path) or not all(os.path.exists(f"{LLVM}/{t}") for t in
                                           ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")):
        pytest.skip("no in-tree build objects or no LLVM tools here")
    meta = {k: v for k, v in _kernel_metadata(path, tmp_path).items() if kernel in k and "two_level" not in k}
    assert meta, f"no kernel matching {kernel} in {obj}"
    for name, m in meta.items():
        # the last template argument of the closest-hit kernels says four-child nodes
        quad = name.split("EEEv")[0].endswith("ELb1") and ("extend" in name or "ordered" in name)
        assert m["vgpr_count"] <= (104 if quad else 96), (name, m)
        assert m["vgpr_spill_count"] == 0 and m["sgpr_spill_count"] == 0, (name, m)
