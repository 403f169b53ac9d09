"""How close is k_tile to an ideal schedule of its own work items?  Replays the measured per-item times (launch order)
through (a) greedy list scheduling on 256 CUs, (b) the same per XCD with workgroup i pinned to XCD i % 8 (32 CUs each)."""
import heapq, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
W = H = G = 4096
import vulkan_forge_amd as _vf; lut = _vf.colormap_rgba8("viridis")
h = np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
t = cabi.Terrain(W, H, G, lut); t.set_height(h)

def makespan(times, machines):
    heap = [0.0] * machines
    for x in times:
        heapq.heapreplace(heap, heap[0] + x)
    return max(heap)

for cam in ("default", "fill"):
    t.set_uniforms(b.camera_uniforms(cam, W, H))
    for _ in range(24): t.render()
    t.enable_timing(True); t.render(); tm = t.timings(); it = t.item_stats(); t.enable_timing(False)
    ms = it[:, 3] * 1e-5                                   # raster + fragment, launch order
    ideal = makespan(ms, 256)
    per_xcd = max(makespan(ms[x::8], 32) for x in range(8))
    print(f"{cam}: k_tile {tm['tile_ms']:.3f} ms | items {len(ms)} sum/256 {ms.sum()/256:.3f} longest {ms.max():.3f} | list scheduling on 256 CUs {ideal:.3f} | "
          f"8 XCDs x 32 CUs, item i on XCD i%8: {per_xcd:.3f} (XCD sums/32: {[round(float(ms[x::8].sum()/32),3) for x in range(8)]})")
    order = np.argsort(-ms)
    print(f"   measured times re-sorted exactly: list {makespan(ms[order], 256):.3f}, per-XCD {max(makespan(ms[order][x::8], 32) for x in range(8)):.3f}; "
          f"rank correlation of launch order with true order: {np.corrcoef(np.argsort(np.argsort(-ms)), np.arange(len(ms)))[0,1]:.3f}")
