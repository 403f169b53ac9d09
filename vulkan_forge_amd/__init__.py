"""vulkan_forge_amd -- MI355X-native drop-in for vulkan-forge's headless terrain rasteriser.

Python-visible surface of the reference's shim (python/vulkan_forge/__init__.py:46-177) on top of the
C++/HIP extension `_vulkan_forge` (host: vulkan_forge_amd/host, kernels: vulkan_forge_amd/csrc, C-ABI:
include/vf_hip.h).  `import vulkan_forge` (the alias package at the repository root) gives the same names.

There is no CPU fallback: the extension must be built (`python -c "import __graft_entry__ as g; g.build()"`)
and a HIP device must be present for anything that renders.
"""
from __future__ import annotations

import importlib
import os

import numpy as _np

from ._validate import grid as _grid
from ._validate import png_path, size_wh

try:
    _ext = importlib.import_module("vulkan_forge_amd._vulkan_forge")
except ImportError as exc:  # fail loudly: the HIP extension is the product
    raise ImportError(
        "vulkan_forge_amd: compiled module '_vulkan_forge' (and libvf_hip.so) not found or not loadable in "
        f"{os.path.dirname(os.path.abspath(__file__))}; run __graft_entry__.build() first. Cause: {exc}"
    ) from exc

Renderer = _ext.Renderer
TerrainSpike = _ext.TerrainSpike
Scene = _ext.Scene
colormap_supported = _ext.colormap_supported
colormap_rgba8 = _ext.colormap_rgba8     # extension: the (256, 4) uint8 texels behind a supported name (callers of the C-ABI)
camera_look_at = _ext.camera_look_at
camera_perspective = _ext.camera_perspective
camera_view_proj = _ext.camera_view_proj
enumerate_adapters = _ext.enumerate_adapters
device_probe = _ext.device_probe

__version__ = "0.1.0"


def render_triangle_rgba(width: int, height: int):
    """Deterministic triangle as (H, W, 4) uint8 (reference shim :54-58)."""
    w, h = size_wh(width, height)
    return Renderer(w, h).render_triangle_rgba()


def render_triangle_png(path, width: int, height: int) -> None:
    """Deterministic triangle written as PNG (reference shim :60-64)."""
    w, h = size_wh(width, height)
    out = png_path(path)          # argument errors first: they must not depend on a device being present
    Renderer(w, h).render_triangle_png(out)


def make_terrain(width: int, height: int, grid: int = 128):
    """Validated TerrainSpike constructor (reference shim :66-75)."""
    w, h = size_wh(width, height)
    return TerrainSpike(w, h, _grid(grid))


def dem_stats(heightmap):
    """(min, max, mean, std) of a 2-D float32/float64 C-contiguous heightmap (reference shim :120-127)."""
    a = _np.asarray(heightmap)
    if a.ndim != 2 or a.dtype not in (_np.float32, _np.float64) or not a.flags["C_CONTIGUOUS"]:
        raise RuntimeError("heightmap must be 2-D float32/float64 and C-contiguous")
    a = a.astype(_np.float32, copy=False)
    return float(a.min()), float(a.max()), float(a.mean()), float(a.std(dtype=_np.float32))


def dem_normalize(heightmap, *, mode="minmax", out_range=(0.0, 1.0), eps=1e-8, return_stats=False):
    """minmax / zscore normalisation (reference shim :129-142)."""
    mn, mx, mean, std = dem_stats(heightmap)
    a = _np.asarray(heightmap).astype(_np.float32, copy=False)
    if mode == "minmax":
        lo, hi = (float(v) for v in out_range)
        scale = 0.0 if mx == mn else (hi - lo) / max(mx - mn, float(eps))
        out = (a - mn) * scale + lo
    elif mode == "zscore":
        out = (a - mean) / max(std, float(eps))
    else:
        raise ValueError("mode must be 'minmax' or 'zscore'")
    return (out, (mn, mx, mean, std)) if return_stats else out


def grid_generate(nx: int, nz: int, spacing=(1.0, 1.0), origin="center"):
    """Regular grid mesh: (XY (nx*nz,2) f32, UV (nx*nz,2) f32, indices (M,) u32) (reference shim :153-169)."""
    return _ext.grid_generate(int(nx), int(nz), tuple(float(s) for s in spacing), str(origin))


generate_grid = grid_generate  # legacy alias (reference shim :172)

__all__ = [
    "Renderer", "TerrainSpike", "Scene", "render_triangle_rgba", "render_triangle_png", "make_terrain",
    "colormap_supported", "camera_look_at", "camera_perspective", "camera_view_proj",
    "enumerate_adapters", "device_probe", "dem_stats", "dem_normalize", "grid_generate", "generate_grid",
    "__version__",
]
