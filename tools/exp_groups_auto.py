import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
import numpy as np
import vulkan_forge_amd as vf
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", "bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
W = H = G = 4096
h = np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
t = cabi.Terrain(W, H, G, vf.colormap_rgba8("viridis")); t.set_height(h)
def period(n=100):
    for _ in range(40): t.render()
    best = 1e9
    for _ in range(2):
        t.sync(); t0 = time.perf_counter()
        for _ in range(n): t.render()
        t.sync(); best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best
for cam, shard in (("default", None), ("fill", None), ("default", (2, 8)), ("fill", (2, 8)), ("default", (1, 2)), ("fill", (1, 4))):
    t.set_uniforms(b.camera_uniforms(cam, W, H))
    if shard: t.set_tile_shard(shard[0], shard[1], 0)
    else: t.set_shard(0, 1, 64)
    res = []
    for mode in (0, 1, -1):
        t.set_raster_groups(mode)
        if shard: t.set_tile_shard(shard[0], shard[1], 0)      # (new epoch)
        else: t.set_shard(0, 1, 64)
        p = period(); res.append((mode, p, t.raster_groups()))
    print(cam, shard, " | ".join(f"mode {m}: {p:.4f} ms (in use {g[0]}, probes {g[1][0]:.3f}/{g[1][1]:.3f})" for m, p, g in res), flush=True)
