"""Frames in flight on one GPU: k handles (each with its own streams, plan states and output) render consecutive frames of the C4
workload round-robin -- as the whole frame and as rank 2 of 8.  A rank of eight is bound by its longest work item (0.20 ms) with half
of the CUs idle; a second frame in flight can use them.  usage: exp_flight.py [max handles]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vulkan_forge_amd as vf
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
W = H = G = 4096
h = np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
kmax = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ts = []
for k in range(kmax):
    t = cabi.Terrain(W, H, G, vf.colormap_rgba8("viridis")); t.set_height(h); ts.append(t)
def period(k, n=120):
    for _ in range(40):
        for t in ts[:k]: t.render()
    best = 1e9
    for _ in range(2):
        for t in ts[:k]: t.sync()
        t0 = time.perf_counter()
        for f in range(n): ts[f % k].render()
        for t in ts[:k]: t.sync()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best
for cam in ("default", "fill"):
    for shard in (None, (2, 8)):
        for t in ts:
            t.set_uniforms(b.camera_uniforms(cam, W, H))
            if shard: t.set_tile_shard(shard[0], shard[1], 0)
            else: t.set_shard(0, 1, 64)
        print(f"{cam:8s} {'whole frame' if not shard else 'rank 2 of 8 '}: " + "  ".join(f"{k} in flight {period(k):.4f} ms/frame" for k in range(1, kmax + 1)), flush=True)
