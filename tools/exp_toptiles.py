import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
W = H = G = 4096
lut = np.load("tests/golden/colormaps_rgba8.npz")["viridis"]
h = np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
t = cabi.Terrain(W, H, G, lut); t.set_height(h)
t.set_uniforms(b.camera_uniforms("default", W, H))
for (r, n) in ((0, 1), (2, 8), (3, 8)):
    t.set_shard(r, n, 64)
    for _ in range(3): t.render()
    t.enable_timing(True); t.render(); tm = t.timings(); ts = t.tile_stats(); t.enable_timing(False)
    order = np.argsort(-ts[:, 1].astype(np.int64))[:8]
    ntx = W // 64
    print(f"rank {r}/{n}: total {tm['total_ms']:.3f} ms tile {tm['tile_ms']:.3f} plan {tm['plan_ms']:.3f} boxes {tm['ranges_ms']:.3f}; busy tiles {(ts[:,0]>0).sum()} blocks {ts[:,0].sum()}")
    for k in order:
        print(f"    tile {k} (tx={k % ntx}, local ty={k // ntx}) blocks={ts[k,0]} raster_ms={ts[k,1]*1e-5:.3f} total_ms={ts[k,2]*1e-5:.3f}")
