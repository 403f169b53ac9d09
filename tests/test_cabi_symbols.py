"""The C-ABI shared library loads and exports exactly what include/vf_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_functions():
    src = open(os.path.join(ROOT, "include", "vf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vf_[a-z0-9_]+)\s*\(", src)))


def test_header_and_library_agree():
    from vulkan_forge_amd import cabi
    names = declared_functions()
    assert len(names) >= 20
    assert sorted(cabi.SYMBOLS) == names, set(cabi.SYMBOLS) ^ set(names)
    lib = ctypes.CDLL(cabi.DEFAULT_LIB)
    for n in names:
        assert hasattr(lib, n), f"libvf_hip.so does not export {n}"


def test_entry_points_are_plain_c():
    """extern "C": unmangled names only, and the product library does not link the oracle."""
    import subprocess
    from vulkan_forge_amd import cabi
    out = subprocess.run(["nm", "-D", "--defined-only", cabi.DEFAULT_LIB], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(declared_functions()) <= exported
    deps = subprocess.run(["readelf", "-d", cabi.DEFAULT_LIB], capture_output=True, text=True, check=True).stdout
    assert "oracle" not in deps
    ext = [f for f in os.listdir(os.path.join(ROOT, "vulkan_forge_amd")) if f.startswith("_vulkan_forge")][0]
    deps = subprocess.run(["readelf", "-d", os.path.join(ROOT, "vulkan_forge_amd", ext)], capture_output=True, text=True, check=True).stdout
    assert "libvf_hip.so" in deps and "oracle" not in deps


def test_no_device_is_reported_not_emulated():
    from vulkan_forge_amd import cabi
    lib = cabi.load()
    n = ctypes.c_int(-1)
    rc = lib.vf_device_count(ctypes.byref(n))
    if rc == cabi.VF_OK and n.value > 0:
        pytest.skip("a HIP device is present")
    ctx = ctypes.c_void_p()
    assert lib.vf_ctx_create(0, ctypes.byref(ctx)) == cabi.VF_ERR_NO_DEVICE
    assert lib.vf_last_error().decode() == "No suitable GPU adapter"
    assert not ctx.value


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "vulkan_forge_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), f
                assert "vf_oracle" not in text and "vfo_" not in text, f


def test_only_tests_bench_and_entry_touch_the_oracle():
    """Outside oracle/ itself, only tests/, bench.py (its cpu_baseline leg) and __graft_entry__.py (build + smoke) may import it:
    the measurement probes under tools/ time and inspect the product alone."""
    allowed = {os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")}
    for top in ("tools", "vulkan_forge", "vshade"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith(".py"):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), os.path.join(dirpath, f)
    for f in os.listdir(ROOT):
        path = os.path.join(ROOT, f)
        if f.endswith(".py") and path not in allowed:
            assert not re.search(r"^\s*(import|from)\s+oracle\b", open(path, errors="ignore").read(), flags=re.M), f

