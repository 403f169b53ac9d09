"""C5 (64-pose orbit, 1920x1080, grid 2048) through vf_terrain_render_batch: ms per pose over three laps, the tile kernel's share.
VF_HIP_LIB selects the library (tools/build_variant.sh); with a -DVF_EXPERIMENTS library VF_NO_MOTION_MAP=1 gives round 4's waiting plan."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get("VF_IMPORT_TORCH"): import torch
import vulkan_forge_amd as vf
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
W, H, G = 1920, 1080, 2048
h = np.random.default_rng(20250817).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
t = cabi.Terrain(W, H, G, vf.colormap_rgba8("viridis")); t.set_height(h)
if len(sys.argv) > 2: t.set_raster_groups(int(sys.argv[2]))
us = np.stack([b.orbit_uniforms(k, W, H) for k in range(64)])
outs = None
if os.environ.get("VF_C5_BUFFERS"):                      # poses into N buffers of their own in turn: with three plan states a batch overlaps consecutive poses
    import ctypes as C
    hip = C.CDLL("libamdhip64.so.7"); hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    bufs = []
    for _ in range(int(os.environ["VF_C5_BUFFERS"])):
        p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), 2048 * 1088 * 4) == 0; bufs.append(p.value)
    outs = [bufs[k % len(bufs)] for k in range(64)]
t.render_batch(us[:24], outs[:24] if outs else None); t.sync()
laps = []
for lap in range(3):
    t.enable_timing(True, stats=False)
    t0 = time.perf_counter(); t.render_batch(us, outs); t.sync(); laps.append((time.perf_counter() - t0) / 64 * 1e3)
    tile, period = t.frame_times(); t.enable_timing(False)
if os.environ.get("VF_C5_STATIC"):
    for k in [int(v) for v in os.environ.get("VF_C5_STATIC_POSES", "0,4,8,12").split(",")]:
        t.set_uniforms(us[k])
        for _ in range(40): t.render()
        t.sync(); t0 = time.perf_counter()
        for _ in range(40): t.render()
        t.sync(); print(f"   pose {k} with the camera at rest: {(time.perf_counter() - t0) / 40 * 1e3:.4f} ms per frame", flush=True)
if os.environ.get("VF_C5_POSES"):
    print("   tile kernel per pose, last lap (ms):", " ".join(f"{x:.2f}" for x in tile), flush=True)
print(f"{(sys.argv[1] if len(sys.argv) > 1 else 'default'):28s} ms per pose, three laps: {laps[0]:.4f} {laps[1]:.4f} {laps[2]:.4f}   last lap: tile kernel mean {tile.mean():.4f} median {np.median(tile):.4f}  period median {np.median(period[1:]):.4f}  line loop in use: {t.raster_groups()[0]}", flush=True)
