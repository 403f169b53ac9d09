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


def _toolchain_present():
    """hipcc + pybind11 + a loadable HIP runtime: what the product's native pieces need to build and import."""
    import shutil
    if os.environ.get("VF_TEST_NO_TOOLCHAIN"):               # rehearse the plain-CI case on a ROCm machine
        return False
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        return False
    try:
        import pybind11  # noqa: F401
    except ImportError:
        return False
    return True


# Modules that import the compiled extension / the C-ABI library when they are collected.  On a machine without the ROCm
# toolchain (plain CI) they are left out with a note, and the oracle + fixture tests still run; where the toolchain exists a
# build failure stays a loud error (the HIP extension is the product, there is no fallback to test instead).
NATIVE_MODULES = ["test_host_api.py", "test_tools_cli.py", "test_cabi_symbols.py", "test_dist_gloo.py", "test_gpu_api.py",
                  "test_gpu_dem.py", "test_gpu_parity.py", "test_gpu_rccl_loopback.py", "test_gpu_dist_cabi.py", "test_gpu_c5.py", "test_gpu_soak.py"]
HAVE_TOOLCHAIN = _toolchain_present()
collect_ignore = [] if HAVE_TOOLCHAIN else list(NATIVE_MODULES)


def pytest_report_header(config):
    if not HAVE_TOOLCHAIN:
        return "vulkan_forge_amd: no hipcc/pybind11 here -- only the oracle and fixture tests are collected (" + ", ".join(NATIVE_MODULES) + " left out)"
    return None


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Native pieces are built by __graft_entry__.build(); make sure they exist (cheap when up to date).  Without the ROCm
    toolchain only the checker (oracle/) is built."""
    if HAVE_TOOLCHAIN:
        import __graft_entry__ as g
        g.build()
    else:
        import oracle as O
        O.build()


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
