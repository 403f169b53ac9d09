"""Shared fixtures.  `-m "not gpu"` runs on any machine (oracle + host logic + C-ABI symbol checks);
`-m gpu` needs an MI355X and compares the HIP path with the CPU oracle (tests call through the C-ABI)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (MI355X); run with -m gpu on the GPU box")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Native pieces are built by __graft_entry__.build(); make sure they exist (cheap when up to date)."""
    import __graft_entry__ as g
    g.build()


@pytest.fixture(scope="session")
def luts():
    z = np.load(os.path.join(GOLDEN, "colormaps_rgba8.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def oracle(_built):
    import oracle as O
    return O


def heightmap(seed, n, m=None):
    """Synthetic R32F heightmap of SURVEY.md 8(d): rng.random(float32) * 0.5 - 0.25."""
    rng = np.random.default_rng(seed)
    return rng.random((m or n, n), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)


FILL_CAMERA = ((0.0, 2.2, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, -1.0), 60.0, 0.1, 100.0)
DEFAULT_CAMERA = ((3.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 45.0, 0.1, 100.0)
