"""Emulate N-GPU band sharding on one GPU: time every rank's frame separately; max over ranks ~ parallel frame time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
W = H = G = 4096
import vulkan_forge_amd as _vf; lut = _vf.colormap_rgba8("viridis")
h = np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
t = cabi.Terrain(W, H, G, lut, lib=cabi.load(sys.argv[1] if len(sys.argv) > 1 else cabi.DEFAULT_LIB)); t.set_height(h)
for cam in ("default", "fill"):
    t.set_uniforms(b.camera_uniforms(cam, W, H))
    for band in (64, 0):                                  # 0 = interleaved tiles
        for n in (1, 2, 4, 8):
            times = []
            for r in range(n):
                if band: t.set_shard(r, n, band)
                else: t.set_tile_shard(r, n, {1: 1, 2: 1, 4: 3, 8: 3}[n])
                for _ in range(30): t.render()
                t.enable_timing(True); t.render(); t.render(); tm = t.timings(); t.enable_timing(False)
                times.append(tm["total_ms"])
            print(f"{cam:8s} band={band:4d} N={n}: per-rank ms max={max(times):.3f} min={min(times):.3f} sum={sum(times):.3f}  -> speedup vs N=1 (compute only) = {base/max(times) if n>1 else 1.0:.2f}" if n > 1 else f"{cam:8s} band={band:4d} N=1: {times[0]:.3f} ms", flush=True)
            if n == 1: base = times[0]
