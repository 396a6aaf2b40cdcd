"""Shared test plumbing: import paths, the `gpu` marker, common fixtures."""
import os
import sys

import numpy as np
import pytest

try:
    # torch ships its own HIP runtime; loading it before libmipt.so makes both share one,
    # which the tests that hand torch device pointers to the library rely on.
    import torch  # noqa: F401
except ImportError:
    pass

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "gpu-pathtracer_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def mipt():
    import mipt as m
    return m


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    o.lib()
    return o


@pytest.fixture(scope="session")
def scenes():
    from mipt import scenes as s
    return s


@pytest.fixture(scope="session")
def gpu_ctx(mipt):
    if mipt.device_count() == 0:
        pytest.fail("test marked gpu but no HIP device is visible")
    ctx = mipt.Context(0)
    yield ctx
    ctx.close()


def random_rays(n, seed, extent=14.0, tmax=1e30, dtype=None):
    """Origins uniform in a cube, directions uniform on the sphere."""
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, dtype=dtype)
    rays["o"][:, :3] = rng.uniform(-extent, extent, size=(n, 3)).astype(np.float32)
    rays["o"][:, 3] = tmax
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays["d"][:, :3] = d.astype(np.float32)
    return rays


def same_hits(a, b, nan_equal=False):
    """Bit-exact comparison of two hit arrays; returns indices that differ.  With nan_equal,
    any NaN matches any NaN (x86 and gfx950 generate different NaN sign/payload bits)."""
    bad = (a["shapeid"] != b["shapeid"]) | (a["primid"] != b["primid"])
    bad |= (a["padding"] != b["padding"]).any(axis=1)
    ua, ub = a["uvwt"].view(np.uint32).copy(), b["uvwt"].view(np.uint32).copy()
    if nan_equal:
        ua[np.isnan(a["uvwt"])] = 0x7FC00000
        ub[np.isnan(b["uvwt"])] = 0x7FC00000
    bad |= (ua != ub).any(axis=1)
    return np.nonzero(bad)[0]
