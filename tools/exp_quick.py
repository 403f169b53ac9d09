"""One library (VF_HIP_LIB or the default), three steady-state frame periods at C4: one GPU default camera, one GPU fill camera, rank
2 of 8 (column stripes) default camera -- 40 settle frames, then wall clock over 100 frames rendered back to back; twice.
usage: exp_quick.py [label]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vulkan_forge_amd as vf
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
W = H = G = 4096
h = np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
t = cabi.Terrain(W, H, G, vf.colormap_rgba8("viridis")); t.set_height(h)
def period(n=100):
    for _ in range(40): t.render()
    best = 1e9
    for _ in range(2):
        t.sync(); t0 = time.perf_counter()
        for _ in range(n): t.render()
        t.sync(); best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best
out = []
for cam, shard in (("default", None), ("fill", None), ("default", (2, 8)), ("fill", (2, 8))):
    t.set_uniforms(b.camera_uniforms(cam, W, H))
    if shard: t.set_tile_shard(shard[0], shard[1], 0)
    else: t.set_shard(0, 1, 64)
    out.append(period())
print(f"{(sys.argv[1] if len(sys.argv) > 1 else 'default'):24s} one GPU default {out[0]:.4f}  fill {out[1]:.4f}   rank 2/8 default {out[2]:.4f}  fill {out[3]:.4f}  ms", flush=True)
