"""Where the C4 frame's tile-kernel time sits on the screen: work-item time, (tile, block) pairs and items summed per tile row
(one GPU, default and fill camera).  The far field is the top of the covered rows, the near field the bottom."""
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
    for _ in range(24): t.render()
    t.enable_timing(True); t.render(); tm = t.timings(); it = t.item_stats(); t.enable_timing(False)
    ty = (it[:, 0] & 0xFFFFF) // 64
    ms = it[:, 3] * 1e-5
    tot = ms.sum()
    print(f"{cam}: tile kernel {tm['tile_ms']:.3f} ms, items {len(it)}, pairs {int(it[:, 1].sum())}, item time {tot:.1f} ms (/256 = {tot / 256:.3f})")
    print("   tile row: items  pairs  item-ms  share  us/pair   cumulative share")
    cum = 0.0
    for r in range(64):
        m = ty == r
        if not m.any(): continue
        s = ms[m].sum(); p = int(it[m, 1].sum()); cum += s
        print(f"   {r:8d}: {int(m.sum()):5d} {p:6d} {s:8.2f} {100 * s / tot:6.1f}% {1e3 * s / max(p, 1):7.2f}   {100 * cum / tot:6.1f}%")
