"""Does a handle's frame time depend on what it rendered before?  Static camera (1920 x 1080, grid 2048), blocks of 64 frames; every
second trial is preceded by five frames of an unrelated view.  (Face-value strip times gave two fixed points: 0.50 / 0.56 ms.)"""
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
W, H, g = 1920, 1080, 2048
tex = np.random.default_rng(20250817).random((g, g), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
t = cabi.Terrain(W, H, g, lut); t.set_height(tex)
u0 = uniforms(W, H, (3 * math.sqrt(2), 2.0, 0.0))
other = uniforms(W, H, (-3.5, 1.5, 2.0))
for trial in range(6):
    if trial % 2: 
        for _ in range(5): t.set_uniforms(other); t.render()      # disturb the feedback state
    t.set_uniforms(u0)
    out = []
    for block in range(6):
        t.sync(); t0 = time.perf_counter()
        for k in range(64): t.render()
        t.sync(); out.append(round((time.perf_counter() - t0) / 64 * 1e3, 3))
    print("trial", trial, "ms/frame per block of 64 frames:", out, flush=True)
