"""How many distinct grid blocks does a frame draw, and how many (tile, block) pairs?  (C4, both cameras; one GPU)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vulkan_forge_amd as vf
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
W = H = G = 4096
h = np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
t = cabi.Terrain(W, H, G, vf.colormap_rgba8("viridis")); t.set_height(h)
for cam in ("default", "fill"):
    t.set_uniforms(b.camera_uniforms(cam, W, H))
    for _ in range(16): t.render()
    t.enable_timing(True); t.render(); t.render(); tm = t.timings(); t.enable_timing(False)
    nb = ((G - 1 + 7) // 8) ** 2
    print(f"{cam:8s}: pairs {tm['blocks_rasterised']}  distinct blocks {tm['blocks_distinct']} of {nb} ({100.0 * tm['blocks_distinct'] / nb:.1f} %)  tile_ms {tm['tile_ms']:.3f} period {tm['total_ms']:.3f}", flush=True)
