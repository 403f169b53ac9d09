import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
W = H = G = 4096
lut = np.load("tests/golden/colormaps_rgba8.npz")["viridis"]
h = np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
t = cabi.Terrain(W, H, G, lut, lib=cabi.load(sys.argv[1])); t.set_height(h)
for cam in ("default", "fill"):
    t.set_uniforms(b.camera_uniforms(cam, W, H))
    for (r, n) in ((0, 1), (2, 8)):
        t.set_shard(r, n, 64)
        out = []
        for f in range(14):
            t.enable_timing(True); t.render(); tm = t.timings(); t.enable_timing(False)
            out.append(round(tm["total_ms"], 3))
        print(sys.argv[1], cam, f"rank {r}/{n}", out, flush=True)
