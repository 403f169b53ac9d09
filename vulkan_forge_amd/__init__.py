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


def _dem_f32(heightmap):
    """The heightmap as a float32 array, or the shim's RuntimeError: 2-D, float32 or float64, C-contiguous."""
    dem = _np.asarray(heightmap)
    acceptable = dem.ndim == 2 and dem.dtype.kind == "f" and dem.dtype.itemsize in (4, 8) and dem.flags.c_contiguous
    if not acceptable:
        raise RuntimeError("heightmap must be 2-D float32/float64 and C-contiguous")
    return dem if dem.dtype == _np.float32 else dem.astype(_np.float32)


def dem_stats(heightmap):
    """(min, max, mean, std) of a heightmap, all taken on its float32 values; std accumulates in float32 as well
    (python/vulkan_forge/__init__.py:120-127; pinned by tests/test_host_api.py against values captured from the reference shim)."""
    dem = _dem_f32(heightmap)
    return tuple(float(v) for v in (dem.min(), dem.max(), dem.mean(), dem.std(dtype=_np.float32)))


def _rescale_minmax(dem, stats, out_range, eps):
    lowest, highest = stats[0], stats[1]
    lo, hi = float(out_range[0]), float(out_range[1])
    gain = (hi - lo) / max(highest - lowest, eps) if highest != lowest else 0.0     # a flat map lands on `lo`
    return (dem - lowest) * gain + lo


def _rescale_zscore(dem, stats, _out_range, eps):
    return (dem - stats[2]) / max(stats[3], eps)


_DEM_MODES = {"minmax": _rescale_minmax, "zscore": _rescale_zscore}


def dem_normalize(heightmap, *, mode="minmax", out_range=(0.0, 1.0), eps=1e-8, return_stats=False):
    """Heights rescaled to `out_range` ("minmax") or to zero mean / unit deviation ("zscore"), float32; with return_stats also the
    statistics they were derived from (python/vulkan_forge/__init__.py:129-142)."""
    stats = dem_stats(heightmap)                             # (argument errors first, like the shim)
    rescale = _DEM_MODES.get(mode)
    if rescale is None:
        raise ValueError("mode must be 'minmax' or 'zscore'")
    out = rescale(_dem_f32(heightmap), stats, out_range, float(eps))
    return (out, stats) if return_stats else out


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
