"""The C-ABI library loads on a machine without a GPU, exports every symbol the header
declares, and refuses device work loudly instead of falling back to the CPU."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for hdr in ("mipt.h", "mipt_render.h"):
        path = os.path.join(REPO, "include", hdr)
        if not os.path.exists(path):
            continue
        text = open(path).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_every_declared_symbol_is_exported(mipt):
    names = _declared_symbols()
    assert len(names) >= 35
    lib = ctypes.CDLL(mipt.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"


def test_python_binding_covers_the_header(mipt):
    missing = [n for n in _declared_symbols() if n not in mipt._PROTOS]
    assert not missing, f"no ctypes prototype for: {missing}"


def test_no_cpu_fallback_for_device_entry_points(mipt):
    if mipt.device_count() > 0:
        pytest.skip("a GPU is visible; this checks the no-GPU behaviour")
    with pytest.raises(mipt.MiptError) as e:
        mipt.Context(0)
    assert e.value.code == mipt.PT_ERR_DEVICE
    assert "no CPU path" in e.value.message


def test_product_does_not_link_the_oracle(mipt):
    """The shipped library must not depend on oracle/ in any form."""
    import subprocess
    out = subprocess.run(["ldd", mipt.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    syms = subprocess.run(["nm", "-D", "--defined-only", mipt.LIB_PATH], capture_output=True,
                          text=True).stdout
    assert " o_" not in syms
