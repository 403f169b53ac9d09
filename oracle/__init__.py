"""CPU oracle bindings (ctypes) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (vulkan_forge_amd) never does.  See oracle/vf_oracle.c for the restated algorithm
and its parity status ("parity unpinned" for rendered pixels; pinned for grid/camera/uniforms).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)


def _cpu_has_v3() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    flags = set(line.split(":", 1)[1].split())
                    return "fma" in flags and "avx2" in flags
    except OSError:
        pass
    return False


def build(force: bool = False) -> None:
    """Compile the oracle with gcc (oracle/Makefile)."""
    targets = [os.path.join(_HERE, n) for n in ("libvf_oracle.so", "libvf_oracle_generic.so")]
    src = os.path.join(_HERE, "vf_oracle.c")
    stale = force or any(not os.path.exists(t) or os.path.getmtime(t) < os.path.getmtime(src) for t in targets)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)


def _load() -> C.CDLL:
    name = "libvf_oracle.so" if _cpu_has_v3() else "libvf_oracle_generic.so"
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    lib.vfo_sincos.argtypes = [_f32p, _f32p, _f32p, C.c_int]
    lib.vfo_srgb_tables.argtypes = [_f32p, _f32p]
    lib.vfo_srgb_encode.argtypes = [_f32p, _u8p, C.c_int]
    lib.vfo_lut_to_linear_u8.argtypes = [_u8p, _u8p, C.c_int]
    lib.vfo_camera_look_at.argtypes = [_f32p, _f32p, _f32p, _f32p]
    lib.vfo_camera_look_at.restype = C.c_char_p
    lib.vfo_camera_perspective.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, _f32p]
    lib.vfo_camera_perspective.restype = C.c_char_p
    lib.vfo_camera_view_proj.argtypes = [_f32p, _f32p, _f32p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, _f32p]
    lib.vfo_camera_view_proj.restype = C.c_char_p
    lib.vfo_default_uniforms.argtypes = [C.c_int, C.c_uint32, C.c_uint32, _f32p]
    lib.vfo_look_at_uniforms.argtypes = [C.c_int, C.c_uint32, C.c_uint32, _f32p, _f32p, _f32p,
                                         C.c_float, C.c_float, C.c_float, _f32p]
    lib.vfo_look_at_uniforms.restype = C.c_char_p
    lib.vfo_grid_generate.argtypes = [C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_char_p, _f32p, _f32p, _u32p]
    lib.vfo_grid_generate.restype = C.c_char_p
    lib.vfo_grid_uses_u16.argtypes = [C.c_uint32, C.c_uint32]
    lib.vfo_build_grid_xyuv.argtypes = [C.c_uint32, _f32p, _u32p]
    lib.vfo_render_terrain.argtypes = [_f32p, C.c_uint32, C.c_uint32, C.c_uint32, _f32p, C.c_uint32, C.c_uint32,
                                       _u8p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, _u8p, _u32p, C.c_int]
    lib.vfo_render_terrain_mode.argtypes = lib.vfo_render_terrain.argtypes + [C.c_int]
    lib.vfo_render_triangle.argtypes = [C.c_uint32, C.c_uint32, _u8p]
    lib.vfo_raster_triangles.argtypes = [_f32p, C.c_uint32, C.c_uint32, C.c_uint32, _u32p]
    lib.vfo_dem_ingest_f32.argtypes = [_f32p, _f32p, C.c_size_t, C.c_float]
    lib.vfo_dem_ingest_f64.argtypes = [C.POINTER(C.c_double), _f32p, C.c_size_t, C.c_float]
    lib.vfo_dem_stats.argtypes = [_f32p, C.c_size_t, _f32p]
    lib.vfo_dem_normalize.argtypes = [_f32p, C.c_size_t, C.c_int, C.c_float, C.c_float, C.c_float, _f32p]
    lib.vfo_dem_percentile_range.argtypes = [_f32p, C.c_size_t, _f32p, _f32p]
    return lib


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def _vec3(v):
    return np.ascontiguousarray(np.asarray(v, dtype=np.float32).reshape(3))


_CLIP = {"wgpu": 0, "gl": 1}

# dummy height textures of the two classes (src/terrain/mod.rs:342-378, src/scene/mod.rs:142-189)
SPIKE_DUMMY_HEIGHT = np.zeros((1, 1), dtype=np.float32)
SCENE_DUMMY_HEIGHT = np.array([[0.0, 0.25], [0.5, 0.75]], dtype=np.float32)
KIND_SPIKE, KIND_SCENE = 0, 1


def sincos(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    s = np.empty_like(x)
    c = np.empty_like(x)
    lib().vfo_sincos(_p(x, _f32p), _p(s, _f32p), _p(c, _f32p), x.size)
    return s, c


def srgb_tables():
    d = np.empty(256, np.float32)
    t = np.empty(256, np.float32)
    lib().vfo_srgb_tables(_p(d, _f32p), _p(t, _f32p))
    return d, t


def srgb_encode(c):
    c = np.ascontiguousarray(c, dtype=np.float32)
    out = np.empty(c.shape, np.uint8)
    lib().vfo_srgb_encode(_p(c, _f32p), _p(out, _u8p), c.size)
    return out


def lut_to_linear_u8(lut):
    lut = np.ascontiguousarray(lut, dtype=np.uint8).reshape(-1, 4)
    out = np.empty_like(lut)
    lib().vfo_lut_to_linear_u8(_p(lut, _u8p), _p(out, _u8p), lut.shape[0])
    return out


def _raise(err, exc=RuntimeError):
    if err:
        raise exc(err.decode())


def camera_look_at(eye, target, up):
    out = np.empty((4, 4), np.float32)
    _raise(lib().vfo_camera_look_at(_p(_vec3(eye), _f32p), _p(_vec3(target), _f32p), _p(_vec3(up), _f32p), _p(out, _f32p)))
    return out


def camera_perspective(fovy_deg, aspect, znear, zfar, clip_space="wgpu"):
    out = np.empty((4, 4), np.float32)
    _raise(lib().vfo_camera_perspective(fovy_deg, aspect, znear, zfar, _CLIP.get(clip_space, -1), _p(out, _f32p)))
    return out


def camera_view_proj(eye, target, up, fovy_deg, aspect, znear, zfar, clip_space="wgpu"):
    out = np.empty((4, 4), np.float32)
    _raise(lib().vfo_camera_view_proj(_p(_vec3(eye), _f32p), _p(_vec3(target), _f32p), _p(_vec3(up), _f32p),
                                      fovy_deg, aspect, znear, zfar, _CLIP.get(clip_space, -1), _p(out, _f32p)))
    return out


def default_uniforms(kind, W, H):
    u = np.empty(44, np.float32)
    lib().vfo_default_uniforms(kind, W, H, _p(u, _f32p))
    return u


def look_at_uniforms(kind, W, H, eye, target, up, fovy_deg, znear, zfar):
    u = np.empty(44, np.float32)
    _raise(lib().vfo_look_at_uniforms(kind, W, H, _p(_vec3(eye), _f32p), _p(_vec3(target), _f32p), _p(_vec3(up), _f32p),
                                      fovy_deg, znear, zfar, _p(u, _f32p)))
    return u


def grid_generate(nx, nz, spacing=(1.0, 1.0), origin="center"):
    nx, nz = int(nx), int(nz)
    nv = max(nx, 0) * max(nz, 0)
    ni = 6 * max(nx - 1, 0) * max(nz - 1, 0)
    xy = np.empty((nv, 2), np.float32)
    uv = np.empty((nv, 2), np.float32)
    idx = np.empty((ni,), np.uint32)
    _raise(lib().vfo_grid_generate(nx, nz, float(spacing[0]), float(spacing[1]), str(origin).encode(),
                                   _p(xy, _f32p), _p(uv, _f32p), _p(idx, _u32p)), ValueError)
    return xy, uv, idx


def grid_uses_u16(nx, nz):
    return bool(lib().vfo_grid_uses_u16(nx, nz))


def build_grid_xyuv(n):
    n = max(int(n), 2)
    verts = np.empty((n * n, 4), np.float32)
    idx = np.empty((6 * (n - 1) * (n - 1),), np.uint32)
    lib().vfo_build_grid_xyuv(n, _p(verts, _f32p), _p(idx, _u32p))
    return verts, idx


SHADE_REFERENCE, SHADE_SPEC_T32 = 0, 1


def render_terrain(u, W, H, grid, height, lut_rgba8, lut_is_srgb=True, rank=0, nranks=1, band_h=64,
                   want_vis=True, nthreads=1, shade_mode=SHADE_REFERENCE):
    """Returns (rgba (H,W,4) u8, vis (H,W) u32 or None). vis = primitive id + 1, 0 = background.
    shade_mode SHADE_SPEC_T32 = the documented-but-unimplemented fragment stage (see frag_terrain in vf_oracle.c)."""
    u = np.ascontiguousarray(u, dtype=np.float32)
    assert u.shape == (44,)
    height = np.ascontiguousarray(height, dtype=np.float32)
    assert height.ndim == 2
    lut = np.ascontiguousarray(lut_rgba8, dtype=np.uint8).reshape(1024)
    rgba = np.empty((H, W, 4), np.uint8)
    vis = np.empty((H, W), np.uint32)
    rc = lib().vfo_render_terrain_mode(_p(u, _f32p), W, H, grid, _p(height, _f32p), height.shape[1], height.shape[0],
                                       _p(lut, _u8p), int(bool(lut_is_srgb)), rank, nranks, band_h,
                                       _p(rgba, _u8p), _p(vis, _u32p), int(nthreads), int(shade_mode))
    if rc != 0:
        raise MemoryError("oracle allocation failed")
    return rgba, (vis if want_vis else None)


def render_triangle(W, H):
    rgba = np.empty((H, W, 4), np.uint8)
    lib().vfo_render_triangle(W, H, _p(rgba, _u8p))
    return rgba


def raster_triangles(clip_xyzw, W, H):
    """clip_xyzw: (ntris, 3, 4) float32 clip-space vertices -> (H, W) u32 surviving primitive id + 1."""
    v = np.ascontiguousarray(clip_xyzw, dtype=np.float32).reshape(-1, 3, 4)
    vis = np.empty((H, W), np.uint32)
    if lib().vfo_raster_triangles(_p(v, _f32p), v.shape[0], W, H, _p(vis, _u32p)) != 0:
        raise MemoryError("oracle allocation failed")
    return vis


def dem_ingest(heightmap, exaggeration):
    a = np.ascontiguousarray(heightmap)
    out = np.empty(a.shape, np.float32)
    if a.dtype == np.float32:
        lib().vfo_dem_ingest_f32(_p(a, _f32p), _p(out, _f32p), a.size, exaggeration)
    elif a.dtype == np.float64:
        lib().vfo_dem_ingest_f64(a.ctypes.data_as(C.POINTER(C.c_double)), _p(out, _f32p), a.size, exaggeration)
    else:
        raise TypeError("float32 or float64")
    return out


def dem_stats(heights):
    h = np.ascontiguousarray(heights, dtype=np.float32)
    out = np.empty(4, np.float32)
    lib().vfo_dem_stats(_p(h, _f32p), h.size, _p(out, _f32p))
    return tuple(float(v) for v in out)


def dem_normalize(heights, mode, eps=1e-8, out_range=(0.0, 1.0)):
    h = np.array(heights, dtype=np.float32, order="C")
    st = np.array(dem_stats(h), np.float32)
    lib().vfo_dem_normalize(_p(h, _f32p), h.size, 1 if mode == "zscore" else 0, eps, out_range[0], out_range[1], _p(st, _f32p))
    return h


def dem_percentile_range(heights):
    h = np.ascontiguousarray(heights, dtype=np.float32)
    p1, p99 = C.c_float(), C.c_float()
    if lib().vfo_dem_percentile_range(_p(h, _f32p), h.size, C.byref(p1), C.byref(p99)) != 0:
        raise MemoryError
    return p1.value, p99.value


def max_threads():
    return int(lib().vfo_max_threads())
