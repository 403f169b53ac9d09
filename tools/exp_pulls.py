"""Blocks pulled from the work lists against blocks still live at the pull (late culling): the library built with -DVF_DBG_PULLS
reports pulls in the item statistics' block column; run once with it and once with the plain library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
W = H = G = 4096
import vulkan_forge_amd as _vf; lut = _vf.colormap_rgba8("viridis")
h = np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
t = cabi.Terrain(W, H, G, lut); t.set_height(h)
for cam in ("default", "fill"):
    t.set_uniforms(b.camera_uniforms(cam, W, H))
    for (r, n) in ((0, 1), (2, 8)):
        if n == 1: t.set_shard(0, 1, 64)
        else: t.set_tile_shard(r, n, 0)
        for _ in range(24): t.render()
        t.enable_timing(True); t.render(); it = t.item_stats(); t.enable_timing(False)
        print(f"{os.path.basename(os.environ.get('VF_HIP_LIB', 'libvf_hip.so')):18s} {cam:8s} {r}/{n}: items {len(it):4d}  block column sum {int(it[:, 1].sum())}")
