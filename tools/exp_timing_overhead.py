import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import vulkan_forge_amd as vf
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", "bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
W = H = G = 4096
h = np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
t = cabi.Terrain(W, H, G, vf.colormap_rgba8("viridis")); t.set_height(h); t.set_uniforms(b.camera_uniforms("default", W, H))
def period(n=100):
    for _ in range(40): t.render()
    best = 1e9
    for _ in range(3):
        t.sync(); t0 = time.perf_counter()
        for _ in range(n): t.render()
        t.sync(); best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best
print("untimed", round(period(), 4))
t.enable_timing(True, stats=False); print("timing mode 2 (five events per frame)", round(period(), 4)); t.enable_timing(False)
print("untimed", round(period(), 4))
