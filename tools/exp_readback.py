"""Read-back latency of a 4096 x 4096 frame: fresh NumPy array (first-touch page faults) vs an array that was written before."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
W = H = G = 4096
lut = np.load("tests/golden/colormaps_rgba8.npz")["viridis"]
h = np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
t = cabi.Terrain(W, H, G, lut); t.set_height(h); t.set_uniforms(b.camera_uniforms("default", W, H))
for _ in range(4): t.render()
t.sync()
warm = np.zeros((H, W, 4), np.uint8)
for name, fresh in (("array written before", False), ("fresh np.empty", True), ("array written before", False), ("fresh np.empty", True)):
    ts = []
    for _ in range(5):
        dst = np.empty((H, W, 4), np.uint8) if fresh else warm
        t0 = time.perf_counter(); t._check(t.lib.vf_terrain_read_rgba(t.t, dst.ctypes.data, 0, H)); ts.append((time.perf_counter() - t0) * 1e3)
        ref = dst if fresh else None
    print(f"read_rgba into {name:22s}: min {min(ts):.2f} ms  median {sorted(ts)[2]:.2f} ms  ({W*H*4/min(ts)/1e6:.1f} GB/s)")
