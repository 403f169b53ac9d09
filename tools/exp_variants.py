"""Time library variants (build/variants/libvf_*.so) on the C4 workload: interleaved rounds in one process."""
import glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
W = H = G = int(os.environ.get("VF_SIZE", 4096))
lut = np.load("tests/golden/colormaps_rgba8.npz")["viridis"]
h = np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
libs = sorted(glob.glob("build/variants/libvf_*.so"))
objs = {}
for p in libs:
    t = cabi.Terrain(W, H, G, lut, lib=cabi.load(p)); t.set_height(h); objs[os.path.basename(p)] = t
for cam in ("default", "fill"):
    u = b.camera_uniforms(cam, W, H)
    res = {k: [] for k in objs}
    for rnd in range(3):
        for k, t in objs.items():
            t.set_uniforms(u); [t.render() for _ in range(12)]; t.enable_timing(True); t.render(); t.render(); t.render(); t.render(); tm = t.timings(); ts = t.tile_stats(); t.enable_timing(False)
            res[k].append(tm["total_ms"])
            if rnd == 0:
                blk, tr, tf = ts[:, 0].astype(float), ts[:, 1] * 1e-5, ts[:, 2] * 1e-5   # ms
                busy = blk > 0
                print(f"   {k}: pairs={int(blk.sum())} busy_tiles={int(busy.sum())} blocks/tile max={blk.max():.0f} p50(busy)={np.median(blk[busy]):.0f} "
                      f"| tile raster ms max={tr.max():.3f} p50(busy)={np.median(tr[busy]):.3f} sum={tr.sum():.1f} | frag ms max={(tf-tr).max():.3f} sum={(tf-tr).sum():.1f}", flush=True)
    for k, v in res.items():
        print(f"{cam:8s} {k:28s} tile_ms min={min(v):8.3f} med={sorted(v)[1]:8.3f}", flush=True)
