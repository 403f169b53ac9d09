"""Heaviest work items of the C4 frame: which tiles / strips set the critical path of k_tile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
W = H = G = 4096
SKEW = int(os.environ.get("VF_SKEW", "0"))
import vulkan_forge_amd as _vf; lut = _vf.colormap_rgba8("viridis")
h = np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
t = cabi.Terrain(W, H, G, lut); t.set_height(h)
for cam in ("default", "fill"):
    t.set_uniforms(b.camera_uniforms(cam, W, H))
    for (r, n) in ((0, 1), (2, 8)):
        if n == 1: t.set_shard(0, 1, 64)
        else: t.set_tile_shard(r, n, SKEW)
        lay = cabi.tile_layout(W, H, r, n, SKEW, lib=t.lib) if n > 1 else None
        for _ in range(24): t.render()
        t.enable_timing(True); t.render(); tm = t.timings(); it = t.item_stats(); t.enable_timing(False)
        ms = it[:, 2] * 1e-5
        order = np.argsort(-ms)[:12]
        print(f"{cam} rank {r}/{n}: total {tm['total_ms']:.3f} ms tile {tm['tile_ms']:.3f}; items {len(it)} (strips: {np.bincount((it[:,0] >> 24) & 7, minlength=5).tolist()} by log2, slices: {np.bincount((it[:,0] >> 29) & 3, minlength=3).tolist()}) "
              f"pairs {it[:,1].sum()} sum item-ms {ms.sum():.1f} (/256 = {ms.sum()/256:.3f}) max {ms.max():.3f} p50 {np.median(ms):.3f}")
        for k in order:
            code = int(it[k, 0]); tile = code & 0xFFFFF; part = (code >> 20) & 15; lg = (code >> 24) & 7; sl = (code >> 27) & 3; lgs = (code >> 29) & 3
            tx, ty = (tile % 64, tile // 64) if lay is None else lay[tile]
            print(f"    item {k:5d}: tile ({tx},{ty}) strip {part}/{1 << lg} slice {sl}/{1 << lgs} blocks={it[k,1]} raster_ms={ms[k]:.3f} frag_ms={(int(it[k,3])-int(it[k,2]))*1e-5:.3f} us/block={1e3*ms[k]/max(it[k,1],1):.2f}")
