"""Frame by frame on one rank of N (default 2 of 8, column stripes): tile kernel time, items, how the heavy tiles were cut, the longest
item -- does the feedback-driven plan settle, and where?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
rank, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 8)
cam = sys.argv[3] if len(sys.argv) > 3 else "default"
W = H = G = 4096
import vulkan_forge_amd as _vf; lut = _vf.colormap_rgba8("viridis")
h = np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
t = cabi.Terrain(W, H, G, lut); t.set_height(h); t.set_uniforms(b.camera_uniforms(cam, W, H))
if n > 1: t.set_tile_shard(rank, n, 0)
frames = int(os.environ.get("VF_FRAMES", "28"))
tl = []
for f in range(frames):                                   # device times only: the kernels run as untimed
    t.enable_timing(True, stats=False); t.render(); tl.append(t.timings()["tile_ms"])
print("tile_ms per frame, statistics off:", " ".join(f"{x:.3f}" for x in tl))
t.enable_timing(False); t.sync()
import time
t0 = time.perf_counter()
for f in range(200): t.render()
t.sync()
print(f"frame period over 200 untimed frames: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms")
for f in range(frames):
    t.enable_timing(True); t.render(); tm = t.timings(); it = t.item_stats()
    ms = it[:, 2] * 1e-5
    print(f"frame {f:2d}: tile_ms {tm['tile_ms']:.3f} items {len(it):4d} strips {np.bincount((it[:,0] >> 24) & 7, minlength=5).tolist()} slices {np.bincount((it[:,0] >> 29) & 3, minlength=3).tolist()} "
          f"pairs {it[:,1].sum():6d} sum {ms.sum():5.1f} ms (/256 {ms.sum()/256:.3f}) max {ms.max():.3f}")
