"""(tile, block) pairs a frame draws and the tile kernel's time, for one library: C4 default / fill camera, one GPU and rank 2 of 8.
usage: [VF_HIP_LIB=...] exp_pairs.py [label]"""
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
out = []
for cam, shard in (("default", None), ("fill", None), ("default", (2, 8)), ("fill", (2, 8))):
    t.set_uniforms(b.camera_uniforms(cam, W, H))
    if shard: t.set_tile_shard(shard[0], shard[1], 0)
    else: t.set_shard(0, 1, 64)
    for _ in range(40): t.render()
    t.enable_timing(True)
    for _ in range(4): t.render()
    tm = t.timings(); t.enable_timing(False)
    out.append(f"{cam}{'' if not shard else ' rank'}: pairs {tm['blocks_rasterised']} tile {tm['tile_ms']:.4f} ms")
print(f"{(sys.argv[1] if len(sys.argv) > 1 else 'default'):10s} " + " | ".join(out), flush=True)
