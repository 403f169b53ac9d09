"""The fragment stage as a launch of its own (vf_terrain_debug_fragment_stage) on C4 -- run under rocprofv3 (kernel trace or
--pmc FETCH_SIZE / WRITE_SIZE) to see k_resolve's time and HBM traffic.  usage: exp_fragment.py [default|fill|both] [exact]"""
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
which = sys.argv[1] if len(sys.argv) > 1 else "both"
if "exact" in sys.argv[2:]: t.set_shade_precision(0)
for cam in (("default", "fill") if which == "both" else (which,)):
    t.set_uniforms(b.camera_uniforms(cam, W, H)); t.render(); t.sync()
    ft = t.fragment_stage(repeats=10)
    print(f"{cam:8s} {ft['resolve_ms']:.4f} ms = {100 * 201326592 / (ft['resolve_ms'] * 1e-3) / 8e12:.1f} % of 8 TB/s on B_frag; covered {ft['covered_pixels']}; equal to the frame: {ft['equal_to_frame']}", flush=True)
