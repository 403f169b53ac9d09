"""What HIP-event timing costs a frame (round 6): steady-state frame periods and 20-frame bursts (what bench.py times) untimed, with the
events of vf_terrain_enable_timing(t, 2) on every frame, and with those of level 3 on every fourth."""
import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import vulkan_forge_amd as vf
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", "/root/repo/bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
W = H = G = 4096
h = np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
t = cabi.Terrain(W, H, G, vf.colormap_rgba8("viridis")); t.set_height(h)
t.set_uniforms(b.camera_uniforms("default", W, H))
for _ in range(40): t.render()
def period(n=100):
    best = 1e9
    for _ in range(3):
        t.sync(); t0 = time.perf_counter()
        for _ in range(n): t.render()
        t.sync(); best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best
def burst(n=20, reps=10):
    xs = []
    for _ in range(reps):
        t.sync(); t0 = time.perf_counter()
        for _ in range(n): t.render()
        t.sync(); xs.append((time.perf_counter() - t0) / n * 1e3)
    return min(xs), sorted(xs)[len(xs)//2]
print("untimed: 100-frame period %.4f; 20-frame bursts min/median %.4f %.4f" % ((period(),) + burst()))
t.enable_timing(True, stats=False)
print("timing level 2 (HIP events, no stats): 100-frame period %.4f; 20-frame bursts %.4f %.4f" % ((period(),) + burst()))
t.enable_timing(True, stats=False, sampled=True)
print("timing level 3 (HIP events on every 4th frame): 100-frame period %.4f; 20-frame bursts %.4f %.4f" % ((period(),) + burst()))
print("   its own figures:", {k: round(v, 4) for k, v in t.timings().items() if k.endswith("_ms")})
t.enable_timing(False)
print("untimed again: %.4f; bursts %.4f %.4f" % ((period(),) + burst()))
