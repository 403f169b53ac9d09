"""Render back-to-back frames as one rank of an N-rank tile shard (default: rank 2 of 8) -- run under rocprofv3 --kernel-trace
to see the per-frame kernel timeline of a lightly loaded GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (first: one HIP runtime per process)
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
rank, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 8)
W = H = G = 4096
lut = np.load("tests/golden/colormaps_rgba8.npz")["viridis"]
h = np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
t = cabi.Terrain(W, H, G, lut); t.set_height(h); t.set_uniforms(b.camera_uniforms("default", W, H))
if n > 1: t.set_tile_shard(rank, n, 3)
for _ in range(40): t.render()
t.sync()
