"""ctypes view of the C-ABI (include/vf_hip.h) -- for callers that own device memory and streams
(bench.py, the torch.distributed shard driver, the C-ABI tests).  The drop-in Python classes live in
the compiled `_vulkan_forge` module; this file only declares the same entry points for ctypes.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.environ.get("VF_HIP_LIB") or os.path.join(_HERE, "libvf_hip.so")     # override: kernel experiments only

VF_OK, VF_ERR_NO_DEVICE, VF_ERR_HIP, VF_ERR_INVALID, VF_ERR_NOMEM = 0, -1, -2, -3, -4

# every symbol include/vf_hip.h declares (checked by tests/test_cabi_symbols.py)
SYMBOLS = [
    "vf_last_error", "vf_device_count", "vf_device_query", "vf_ctx_create", "vf_ctx_destroy", "vf_ctx_device_info", "vf_ctx_stream",
    "vf_terrain_create", "vf_terrain_destroy", "vf_terrain_set_uniforms", "vf_terrain_set_height",
    "vf_terrain_set_height_device", "vf_terrain_set_shade_mode", "vf_terrain_set_shade_precision", "vf_terrain_set_raster_groups", "vf_terrain_raster_groups", "vf_terrain_set_shard", "vf_terrain_local_rows", "vf_terrain_set_tile_shard",
    "vf_terrain_local_tiles", "vf_terrain_read_tiles", "vf_tile_layout", "vf_terrain_tile_times", "vf_balance_stripes", "vf_tile_layout_register_map", "vf_terrain_set_output_device",
    "vf_terrain_rgba_device", "vf_terrain_render", "vf_terrain_render_batch", "vf_terrain_render_batch_host", "vf_terrain_sync", "vf_terrain_read_rgba", "vf_terrain_read_png_scanlines", "vf_terrain_read_visibility",
    "vf_terrain_enable_timing", "vf_terrain_timings", "vf_terrain_frame_times", "vf_terrain_debug_item_stats", "vf_terrain_debug_phase_cycles", "vf_grid_generate", "vf_grid_generate_device", "vf_triangle_render",
    "vf_stitch_bands_device", "vf_stitch_tiles_device",
    "vf_dist_available", "vf_dist_version", "vf_dist_unique_id", "vf_dist_comm_init", "vf_dist_comm_destroy", "vf_dist_gather_tiles", "vf_dist_gather_bands", "vf_dist_exchange_bands",
    "vf_terrain_debug_fragment_stage", "vf_host_alloc", "vf_host_free",
    "vf_dem_create", "vf_dem_destroy", "vf_dem_set_heights_f32", "vf_dem_set_heights_f64", "vf_dem_stats",
    "vf_dem_percentile_range", "vf_dem_normalize", "vf_dem_upload_height", "vf_dem_texture_size", "vf_dem_read_patch",
]


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 256), ("arch", C.c_char * 64), ("device_ordinal", C.c_int32), ("compute_units", C.c_int32),
                ("wavefront_size", C.c_int32), ("clock_khz", C.c_int32), ("total_mem_bytes", C.c_uint64),
                ("lds_bytes_per_cu", C.c_uint64), ("pci_bus_id", C.c_int32), ("pci_device_id", C.c_int32)]


class Timings(C.Structure):
    _fields_ = [("ranges_ms", C.c_float), ("plan_ms", C.c_float), ("tile_ms", C.c_float), ("total_ms", C.c_float),
                ("blocks_rasterised", C.c_uint32), ("tiles", C.c_uint32), ("frames", C.c_uint32), ("blocks_distinct", C.c_uint32)]


class FragmentTiming(C.Structure):
    _fields_ = [("resolve_ms", C.c_float), ("covered_pixels", C.c_uint32), ("repeats", C.c_uint32), ("equal_to_frame", C.c_uint32)]


DIST_UNIQUE_ID_BYTES = 128
_vp, _u32, _f, _i = C.c_void_p, C.c_uint32, C.c_float, C.c_int
_PROTOS = {
    "vf_last_error": (C.c_char_p, []),
    "vf_device_count": (_i, [C.POINTER(_i)]),
    "vf_device_query": (_i, [_i, C.POINTER(DeviceInfo)]),
    "vf_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "vf_ctx_destroy": (None, [_vp]),
    "vf_ctx_device_info": (_i, [_vp, C.POINTER(DeviceInfo)]),
    "vf_ctx_stream": (_i, [_vp, C.POINTER(_vp)]),
    "vf_terrain_create": (_i, [_vp, _u32, _u32, _u32, _vp, _i, C.POINTER(_vp)]),
    "vf_terrain_destroy": (None, [_vp]),
    "vf_terrain_set_uniforms": (_i, [_vp, _vp]),
    "vf_terrain_set_height": (_i, [_vp, _vp, _u32, _u32]),
    "vf_terrain_set_height_device": (_i, [_vp, _vp, _u32, _u32]),
    "vf_terrain_set_shade_mode": (_i, [_vp, _i]),
    "vf_terrain_set_shade_precision": (_i, [_vp, _i]),
    "vf_terrain_set_raster_groups": (_i, [_vp, _i]),
    "vf_terrain_raster_groups": (_i, [_vp, C.POINTER(_i), _vp]),
    "vf_terrain_set_shard": (_i, [_vp, _u32, _u32, _u32]),
    "vf_terrain_local_rows": (_i, [_vp, C.POINTER(_u32)]),
    "vf_terrain_set_tile_shard": (_i, [_vp, _u32, _u32, _u32]),
    "vf_terrain_local_tiles": (_i, [_vp, C.POINTER(_u32)]),
    "vf_terrain_read_tiles": (_i, [_vp, _vp, _u32, _u32]),
    "vf_tile_layout": (_i, [_u32, _u32, _u32, _u32, _u32, _vp, _u32, C.POINTER(_u32)]),
    "vf_terrain_set_output_device": (_i, [_vp, _vp]),
    "vf_terrain_rgba_device": (_i, [_vp, C.POINTER(_vp)]),
    "vf_terrain_render": (_i, [_vp, _vp]),
    "vf_terrain_render_batch": (_i, [_vp, _vp, _u32, _vp, _vp]),
    "vf_terrain_render_batch_host": (_i, [_vp, _vp, _u32, _vp]),
    "vf_terrain_sync": (_i, [_vp]),
    "vf_terrain_read_rgba": (_i, [_vp, _vp, _u32, _u32]),
    "vf_terrain_read_png_scanlines": (_i, [_vp, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "vf_terrain_read_visibility": (_i, [_vp, _vp]),
    "vf_terrain_enable_timing": (_i, [_vp, _i]),
    "vf_terrain_timings": (_i, [_vp, C.POINTER(Timings)]),
    "vf_terrain_frame_times": (_i, [_vp, _vp, _vp, _u32, C.POINTER(_u32)]),
    "vf_terrain_debug_item_stats": (_i, [_vp, _vp, _u32, C.POINTER(_u32)]),
    "vf_terrain_debug_phase_cycles": (_i, [_vp, _vp, _u32]),
    "vf_terrain_tile_times": (_i, [_vp, _vp, _u32, C.POINTER(_u32)]),
    "vf_balance_stripes": (_i, [_vp, _u32, _u32, _vp]),
    "vf_tile_layout_register_map": (_i, [_vp, _u32, _u32, _u32, C.POINTER(_u32)]),
    "vf_host_alloc": (_i, [C.c_size_t, C.POINTER(_vp)]),
    "vf_host_free": (None, [_vp]),
    "vf_grid_generate": (_i, [_vp, _u32, _u32, _f, _f, _vp, _vp, _vp]),
    "vf_grid_generate_device": (_i, [_vp, _u32, _u32, _f, _f, _vp, _vp, _vp, _vp]),
    "vf_triangle_render": (_i, [_vp, _u32, _u32, _vp]),
    "vf_stitch_bands_device": (_i, [_vp, _vp, _vp, _u32, _u32, _u32, _u32, _vp]),
    "vf_stitch_tiles_device": (_i, [_vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _vp]),
    "vf_dist_available": (_i, []),
    "vf_dist_version": (_i, [C.POINTER(_i)]),
    "vf_dist_unique_id": (_i, [_vp]),
    "vf_dist_comm_init": (_i, [_vp, _vp, _i, _i, C.POINTER(_vp)]),
    "vf_dist_comm_destroy": (None, [_vp]),
    "vf_dist_gather_tiles": (_i, [_vp, _vp, _i, _vp, _u32, _vp]),
    "vf_dist_gather_bands": (_i, [_vp, _vp, _i, _vp, _vp]),
    "vf_dist_exchange_bands": (_i, [_vp, _vp, _i, _vp, _vp]),
    "vf_terrain_debug_fragment_stage": (_i, [_vp, _u32, C.POINTER(FragmentTiming)]),
    "vf_dem_create": (_i, [_vp, C.POINTER(_vp)]),
    "vf_dem_destroy": (None, [_vp]),
    "vf_dem_set_heights_f32": (_i, [_vp, _vp, _u32, _u32, _f]),
    "vf_dem_set_heights_f64": (_i, [_vp, _vp, _u32, _u32, _f]),
    "vf_dem_stats": (_i, [_vp, _vp]),
    "vf_dem_percentile_range": (_i, [_vp, C.POINTER(_f), C.POINTER(_f)]),
    "vf_dem_normalize": (_i, [_vp, _i, _f, _f, _f]),
    "vf_dem_upload_height": (_i, [_vp]),
    "vf_dem_texture_size": (_i, [_vp, C.POINTER(_u32), C.POINTER(_u32)]),
    "vf_dem_read_patch": (_i, [_vp, _u32, _u32, _u32, _u32, _vp]),
}


def load(path: str = DEFAULT_LIB) -> C.CDLL:
    """dlopen the library and attach prototypes.  Raises OSError when it is missing: there is no fallback."""
    lib = C.CDLL(path)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


class VfError(RuntimeError):
    pass


def tile_layout(width, height, rank, nranks, skew=3, lib=None):
    """Tiles of `rank` in its storage order as an (n, 2) int array of (tx, ty) -- vf_tile_layout, host arithmetic only."""
    lib = lib or load()
    n = _u32()
    if lib.vf_tile_layout(width, height, rank, nranks, skew, None, 0, C.byref(n)) != VF_OK:
        raise VfError(lib.vf_last_error().decode())
    packed = np.zeros(max(n.value, 1), np.uint32)
    if lib.vf_tile_layout(width, height, rank, nranks, skew, packed.ctypes.data, n.value, C.byref(n)) != VF_OK:
        raise VfError(lib.vf_last_error().decode())
    packed = packed[:n.value]
    return np.stack([packed & 0xFFFF, packed >> 16], axis=1).astype(np.int64)


def balance_stripes(stripe_ms, nranks, lib=None):
    """vf_balance_stripes: per-stripe times -> owner per stripe (uint8), every rank with the same number of stripes."""
    lib = lib or load()
    ms = np.ascontiguousarray(stripe_ms, np.float32)
    owner = np.zeros(len(ms), np.uint8)
    if lib.vf_balance_stripes(ms.ctypes.data, len(ms), nranks, owner.ctypes.data) != VF_OK:
        raise VfError(lib.vf_last_error().decode())
    return owner


def register_stripe_map(stripe_owner, stripe_log2, nranks, lib=None):
    """vf_tile_layout_register_map: owner table -> layout word accepted wherever a layout is (set_tile_shard, tile_layout, stitch_tiles)."""
    lib = lib or load()
    owner = np.ascontiguousarray(stripe_owner, np.uint8)
    word = _u32()
    if lib.vf_tile_layout_register_map(owner.ctypes.data, len(owner), stripe_log2, nranks, C.byref(word)) != VF_OK:
        raise VfError(lib.vf_last_error().decode())
    return word.value


class Terrain:
    """Thin RAII wrapper over vf_ctx + vf_terrain for callers that pass raw device pointers / streams."""

    def __init__(self, width, height, grid, lut_rgba8, lut_is_srgb=True, device=0, lib=None, share_ctx=None):
        """`share_ctx`: another Terrain whose context (vf_ctx: device, stream, the side streams handles borrow) this one uses too -- what the
        drop-in module does for every object of a process (one context per process and device); the other Terrain must outlive this one."""
        self.lib = lib or load()
        self.W, self.H, self.grid = int(width), int(height), int(grid)
        self.ctx, self.t = _vp(), _vp()
        self._own_ctx = share_ctx is None
        if share_ctx is None:
            self._check(self.lib.vf_ctx_create(int(device), C.byref(self.ctx)))
        else:
            self.ctx = share_ctx.ctx
        lut = np.ascontiguousarray(lut_rgba8, dtype=np.uint8).reshape(1024)
        self._check(self.lib.vf_terrain_create(self.ctx, self.W, self.H, self.grid, lut.ctypes.data, int(bool(lut_is_srgb)),
                                               C.byref(self.t)))

    def _check(self, rc):
        if rc != VF_OK:
            msg = self.lib.vf_last_error().decode()
            raise VfError("No suitable GPU adapter" if rc == VF_ERR_NO_DEVICE else msg)

    def close(self):
        if self.t:
            self.lib.vf_terrain_destroy(self.t)
            self.t = _vp()
        if self.ctx:
            if self._own_ctx:
                self.lib.vf_ctx_destroy(self.ctx)
            self.ctx = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def device_info(self):
        """dict of the vf_device_info of this handle's device (name, arch, ordinal, PCI bus / device id, ...)."""
        di = DeviceInfo()
        self._check(self.lib.vf_ctx_device_info(self.ctx, C.byref(di)))
        return {k: (getattr(di, k).decode() if isinstance(getattr(di, k), bytes) else getattr(di, k)) for k, _ in DeviceInfo._fields_}

    def stream_handle(self):
        """hipStream_t of the context's own stream (what stream=None means), e.g. for torch.cuda.ExternalStream."""
        h = _vp()
        self._check(self.lib.vf_ctx_stream(self.ctx, C.byref(h)))
        return h.value

    def set_uniforms(self, u):
        u = np.ascontiguousarray(u, dtype=np.float32).reshape(44)
        self._check(self.lib.vf_terrain_set_uniforms(self.t, u.ctypes.data))

    def set_height(self, h):
        h = np.ascontiguousarray(h, dtype=np.float32)
        self._check(self.lib.vf_terrain_set_height(self.t, h.ctypes.data, h.shape[1], h.shape[0]))

    def set_height_device(self, dptr, tw, th):
        self._check(self.lib.vf_terrain_set_height_device(self.t, _vp(dptr), tw, th))

    def set_shade_mode(self, mode):
        """0 = REFERENCE (terrain.wgsl as coded), 1 = SPEC_T32 (documented-only stage; oracle-validated)."""
        self._check(self.lib.vf_terrain_set_shade_mode(self.t, int(mode)))

    def set_shade_precision(self, precision):
        """0 = EXACT (IEEE binary32 in a fixed order: the oracle bit for bit), 1 = FAST (default; hardware rcp / rsq / sin / cos /
        log / exp, within 1 LSB of EXACT, visibility identical)."""
        self._check(self.lib.vf_terrain_set_shade_precision(self.t, int(precision)))

    def set_raster_groups(self, mode):
        """-1: the handle measures both line loops of the raster stage on its own frames and keeps the faster (default); 0 / 1: fixed."""
        self._check(self.lib.vf_terrain_set_raster_groups(self.t, int(mode)))

    def raster_groups(self):
        """(variant that drew the last frame, [tile-kernel ms without groups, with groups]; 0 = not measured in this view)"""
        use, ms = _i(), (C.c_float * 2)()
        self._check(self.lib.vf_terrain_raster_groups(self.t, C.byref(use), C.cast(ms, _vp)))
        return use.value, [float(ms[0]), float(ms[1])]

    def set_shard(self, rank, nranks, band_h=64):
        self._check(self.lib.vf_terrain_set_shard(self.t, rank, nranks, band_h))

    def set_tile_shard(self, rank, nranks, skew=3):
        """`skew`: the layout word skew | stripe_log2 << 16 (include/vf_hip.h, VF_TILE_LAYOUT)."""
        self._check(self.lib.vf_terrain_set_tile_shard(self.t, rank, nranks, skew))

    def tile_times(self):
        """ms the last frame spent on each local tile (storage order: vf_tile_layout's)."""
        n = _u32()
        self._check(self.lib.vf_terrain_tile_times(self.t, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.float32)
        self._check(self.lib.vf_terrain_tile_times(self.t, out.ctypes.data, n.value, C.byref(n)))
        return out

    def local_tiles(self):
        n = _u32()
        self._check(self.lib.vf_terrain_local_tiles(self.t, C.byref(n)))
        return n.value

    def read_tiles(self):
        """(local_tiles, 64, 64, 4) uint8, the tile-major buffer of a tile-sharded handle."""
        n = self.local_tiles()
        out = np.empty((n, 64, 64, 4), np.uint8)
        if n:
            self._check(self.lib.vf_terrain_read_tiles(self.t, out.ctypes.data, 0, n))
        return out

    def stitch_tiles(self, gathered_dptr, image_dptr, nranks, skew, stride_tiles, stream=None, height=None):
        """`height`: stitch a frame of that many rows instead of the handle's (a band of the frame: dist.BandStitchExchange)."""
        self._check(self.lib.vf_stitch_tiles_device(self.ctx, _vp(gathered_dptr), _vp(image_dptr), self.W, self.H if height is None else int(height),
                                                    nranks, skew, stride_tiles, _vp(stream or 0)))

    # ---- RCCL exchange through the C-ABI (include/vf_hip.h, "multi-GPU exchange over RCCL") ----
    def dist_unique_id(self):
        """128 bytes for vf_dist_comm_init: made on one rank, handed to all of them by the host."""
        buf = (C.c_uint8 * DIST_UNIQUE_ID_BYTES)()
        self._check(self.lib.vf_dist_unique_id(C.cast(buf, _vp)))
        return bytes(buf)

    def dist_comm_init(self, unique_id: bytes, rank: int, nranks: int):
        """ncclCommInitRank on this handle's device; returns the communicator (an integer handle = ncclComm_t)."""
        buf = (C.c_uint8 * DIST_UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        comm = _vp()
        self._check(self.lib.vf_dist_comm_init(self.ctx, C.cast(buf, _vp), int(rank), int(nranks), C.byref(comm)))
        return comm.value

    def dist_comm_destroy(self, comm):
        self.lib.vf_dist_comm_destroy(_vp(comm))

    def dist_gather_tiles(self, comm, root, gathered_dptr, stride_tiles, stream=None):
        self._check(self.lib.vf_dist_gather_tiles(self.t, _vp(comm), int(root), _vp(gathered_dptr or 0), int(stride_tiles), _vp(stream or 0)))

    def dist_gather_bands(self, comm, root, image_dptr, stream=None):
        self._check(self.lib.vf_dist_gather_bands(self.t, _vp(comm), int(root), _vp(image_dptr or 0), _vp(stream or 0)))

    def dist_exchange_bands(self, comm, root, image_dptr, stream=None):
        """Column-stripe tile shards -> the frame on `root`: all-to-all + one band stitched per rank + bands gathered in place."""
        self._check(self.lib.vf_dist_exchange_bands(self.t, _vp(comm), int(root), _vp(image_dptr or 0), _vp(stream or 0)))

    def dist_version(self):
        v = _i()
        self._check(self.lib.vf_dist_version(C.byref(v)))
        return v.value

    def fragment_stage(self, repeats=10):
        """The fragment stage as a launch of its own (diagnostics): dict(resolve_ms, covered_pixels, repeats, equal_to_frame)."""
        ft = FragmentTiming()
        self._check(self.lib.vf_terrain_debug_fragment_stage(self.t, int(repeats), C.byref(ft)))
        return {k: getattr(ft, k) for k, _ in FragmentTiming._fields_}

    def local_rows(self):
        r = _u32()
        self._check(self.lib.vf_terrain_local_rows(self.t, C.byref(r)))
        return r.value

    def set_output_device(self, dptr):
        self._check(self.lib.vf_terrain_set_output_device(self.t, _vp(dptr)))

    def render(self, stream=None):
        self._check(self.lib.vf_terrain_render(self.t, _vp(stream or 0)))

    def render_batch(self, uniforms, outputs=None, stream=None):
        """n poses back to back (BASELINE config 5): uniforms (n, 44) float32; outputs: n device pointers (frame k -> outputs[k]) or None."""
        u = np.ascontiguousarray(uniforms, np.float32).reshape(-1, 44)
        ptrs = None
        if outputs is not None:
            assert len(outputs) == len(u)
            ptrs = (_vp * len(u))(*[_vp(int(p)) for p in outputs])
        self._check(self.lib.vf_terrain_render_batch(self.t, u.ctypes.data, len(u), ptrs, _vp(stream or 0)))

    def render_batch_host(self, uniforms):
        """n poses, every frame read back: (n, H, W, 4) uint8 (pageable NumPy memory: the runtime stages the copies)."""
        u = np.ascontiguousarray(uniforms, np.float32).reshape(-1, 44)
        out = np.empty((len(u), self.H, self.W, 4), np.uint8)
        ptrs = (_vp * len(u))(*[_vp(out[k].ctypes.data) for k in range(len(u))])
        self._check(self.lib.vf_terrain_render_batch_host(self.t, u.ctypes.data, len(u), ptrs))
        return out

    def sync(self):
        self._check(self.lib.vf_terrain_sync(self.t))

    def read_rgba(self):
        rows = self.local_rows()
        out = np.empty((rows, self.W, 4), np.uint8)
        self._check(self.lib.vf_terrain_read_rgba(self.t, out.ctypes.data, 0, rows))
        return out

    def read_png_scanlines(self):
        """(H, 4W+1) uint8 copy of the handle's pinned PNG scanlines (filter byte + filtered row) of the last frame."""
        ptr, n = _vp(), C.c_size_t()
        self._check(self.lib.vf_terrain_read_png_scanlines(self.t, C.byref(ptr), C.byref(n)))
        buf = (C.c_uint8 * n.value).from_address(ptr.value)
        return np.frombuffer(buf, np.uint8).reshape(self.H, self.W * 4 + 1).copy()

    def read_visibility(self):
        out = np.empty((self.local_rows(), self.W), np.uint32)
        self._check(self.lib.vf_terrain_read_visibility(self.t, out.ctypes.data))
        return out

    def enable_timing(self, on=True, stats=True, sampled=False):
        """stats=False: HIP events only, the kernels run exactly as untimed (no per-item statistics; blocks_* read 0);
        sampled=True (with stats=False): events on every fourth frame only."""
        self._check(self.lib.vf_terrain_enable_timing(self.t, (1 if stats else (3 if sampled else 2)) if on else 0))

    def item_stats(self):
        """(items, 4) u32 per work item of the last frame: code (local tile | strip << 20 | log2 strips << 24), candidate
        blocks, raster ticks, raster+fragment ticks (10 ns)."""
        cap = self.timings()["tiles"] + 2048
        out = np.zeros((cap, 4), np.uint32)
        n = _u32()
        self._check(self.lib.vf_terrain_debug_item_stats(self.t, out.ctypes.data, cap, C.byref(n)))
        return out[:n.value]

    def tile_stats(self):
        """(ntiles, 3) u32 per local tile: candidate blocks (sum over strips), raster ticks, raster+fragment ticks (max)."""
        it = self.item_stats()
        out = np.zeros((self.timings()["tiles"], 3), np.uint32)
        tile = it[:, 0] & 0xFFFFF
        np.add.at(out[:, 0], tile, it[:, 1])
        np.maximum.at(out[:, 1], tile, it[:, 2])
        np.maximum.at(out[:, 2], tile, it[:, 3])
        return out

    def phase_cycles(self):
        """(40,) u64: [0:8] shader-clock cycles per phase summed over waves, [8:16] event counts, [16:24] parts of the set-up phase, [24:30] wave-level executions of the line loop's parts, [30:34] parts of the vertex phase
        (libraries built with -DVF_PHASE_PROF only)."""
        out = np.zeros(40, np.uint64)
        self._check(self.lib.vf_terrain_debug_phase_cycles(self.t, out.ctypes.data, 40))
        return out

    def frame_times(self):
        """(tile_ms, period_ms) per timed frame, oldest first (at most the last 64); period_ms[0] is 0."""
        tile, period, n = np.zeros(64, np.float32), np.zeros(64, np.float32), _u32(0)
        self._check(self.lib.vf_terrain_frame_times(self.t, tile.ctypes.data, period.ctypes.data, 64, C.byref(n)))
        return tile[:n.value].copy(), period[:n.value].copy()

    def timings(self):
        tm = Timings()
        self._check(self.lib.vf_terrain_timings(self.t, C.byref(tm)))
        return {k: getattr(tm, k) for k, _ in Timings._fields_}
