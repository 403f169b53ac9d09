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
import vulkan_forge_amd as _vf; lut = _vf.colormap_rgba8("viridis")
h = np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
t = cabi.Terrain(W, H, G, lut); t.set_height(h); t.set_uniforms(b.camera_uniforms("default", W, H))
skew = int(sys.argv[3]) if len(sys.argv) > 3 else 3
if n > 1: t.set_tile_shard(rank, n, skew)
for _ in range(40): t.render()
t.sync()
