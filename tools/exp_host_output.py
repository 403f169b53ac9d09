"""Experiment (round 5): the frame written straight into PINNED HOST memory by the tile kernel's stores (no device buffer, no copy)
against render-to-HBM + one direct D2H copy into pinned memory.  C4, default camera; synchronous calls (what render_rgba does)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vulkan_forge_amd as vf
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
W = H = G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
h = np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
t = cabi.Terrain(W, H, G, vf.colormap_rgba8("viridis")); t.set_height(h)
t.set_uniforms(b.camera_uniforms("default", W, H))
dev = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
pin = torch.empty((H, W, 4), dtype=torch.uint8).pin_memory()
pin2 = torch.empty((H, W, 4), dtype=torch.uint8).pin_memory()
def sync_frames(n, fn):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), sorted(ts)[len(ts) // 2]
t.set_output_device(dev.data_ptr())
for _ in range(20): t.render()
t.sync()
def to_hbm(): t.render(); t.sync()
def to_hbm_copy(): t.render(); t.sync(); pin2.copy_(dev); torch.cuda.synchronize()
print("render -> HBM, sync:                 min %.3f median %.3f ms" % sync_frames(20, to_hbm))
print("render -> HBM, sync, D2H -> pinned:  min %.3f median %.3f ms" % sync_frames(20, to_hbm_copy))
ref = dev.cpu()
t.set_output_device(pin.data_ptr())
for _ in range(5): t.render()
t.sync()
def to_host(): t.render(); t.sync()
print("render -> pinned host memory, sync:  min %.3f median %.3f ms" % sync_frames(20, to_host))
print("same frame:", bool(torch.equal(ref, pin)))
