"""(Round 6: the knobs this drives live in tools/experiments/r06_kernel_laboratory.patch -- apply it first.)
What depth slices would cost: variants of the library that draw only a slice of every tile's block rows (VF_DBG_SLICE_LO/HI,
in 1/256 of the tile's row list) with every busy tile cut into the same number of strips (VF_DBG_FORCE_LG).  Sum of the slices'
pairs and item times against the whole list = the occlusion culling a slice loses by not seeing the slices in front of it."""
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
        for _ in range(6): t.render()
        t.enable_timing(True); t.render(); tm = t.timings(); it = t.item_stats(); t.enable_timing(False)
        ms = it[:, 2] * 1e-5
        k = int(np.argmax(ms))
        print(f"{os.path.basename(os.environ.get('VF_HIP_LIB','lib')):24s} {cam:8s} {r}/{n}: tile_ms {tm['tile_ms']:.3f} items {len(it)} pairs {it[:,1].sum()} sum item-ms {ms.sum():.1f} max {ms.max():.3f} (blocks {it[k,1]})")
