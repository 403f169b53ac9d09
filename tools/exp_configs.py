"""Frame times of the other BASELINE.json configurations (C2, C3, C5 pose batch) -- context next to bench.py's C4 line."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vulkan_forge_amd as vf
from vulkan_forge_amd import cabi
lut = np.load("tests/golden/colormaps_rgba8.npz")["viridis"]

def uniforms(W, H, eye):
    view = vf.camera_look_at(eye, (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)); proj = vf.camera_perspective(45.0, W / H, 0.1, 100.0, "wgpu")
    u = np.zeros(44, np.float32); u[:16] = view.T.reshape(-1); u[16:32] = proj.T.reshape(-1)
    sun = np.array([0.5, 0.8, 0.6], np.float32); u[32:35] = sun / np.sqrt((sun * sun).sum()); u[35] = 1.0; u[36:39] = 1.0
    return u

def run(name, W, H, G, tex, eyes, frames):
    t = cabi.Terrain(W, H, G, lut)
    if tex is not None: t.set_height(tex)
    us = [uniforms(W, H, e) for e in eyes]
    for k in range(8): t.set_uniforms(us[k % len(us)]); t.render()
    t.sync(); t0 = time.perf_counter()
    for k in range(frames): t.set_uniforms(us[k % len(us)]); t.render()
    t.sync(); dt = (time.perf_counter() - t0) / frames
    t.enable_timing(True); t.render(); tm = t.timings(); t.enable_timing(False)
    print(f"{name}: {dt*1e3:.3f} ms/frame wall (back-to-back, no read-back) = {W*H/dt/1e6:.0f} Mpix/s; GPU kernels of one frame {tm['total_ms']:.3f} ms (tile {tm['tile_ms']:.3f})", flush=True)
    t.close()

run("C2 TerrainSpike 800x600 grid 128 (analytic surface)", 800, 600, 128, np.zeros((1, 1), np.float32), [(3.0, 2.0, 3.0)], 200)
g = 1024
run("C3 Scene 1920x1080 grid 1024, rng(20250815) heights", 1920, 1080, g, np.random.default_rng(20250815).random((g, g), dtype=np.float32) * np.float32(0.5) - np.float32(0.25), [(3.0, 2.0, 3.0)], 200)
g = 2048
eyes = [(3 * math.sqrt(2) * math.cos(2 * math.pi * k / 64), 2.0, 3 * math.sqrt(2) * math.sin(2 * math.pi * k / 64)) for k in range(64)]
run("C5 64 poses, 1920x1080 grid 2048, rng(20250817) heights, one GPU", 1920, 1080, g, np.random.default_rng(20250817).random((g, g), dtype=np.float32) * np.float32(0.5) - np.float32(0.25), eyes, 128)
