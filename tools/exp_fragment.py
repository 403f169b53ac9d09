"""The fragment stage as a launch of its own (vf_terrain_debug_fragment_stage) on C4, both cameras -- run under rocprofv3 (kernel trace or
--pmc FETCH_SIZE / WRITE_SIZE) to see k_resolve's time and HBM traffic."""
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
    t.set_uniforms(b.camera_uniforms(cam, W, H)); t.render(); t.sync()
    ft = t.fragment_stage(repeats=5)
    print(cam, ft, flush=True)
