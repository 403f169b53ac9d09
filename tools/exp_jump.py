"""Frame time when the camera jumps between two unrelated views every 1..8 frames (1920 x 1080, grid 2048): how fast the
feedback-driven plan recovers after a jump."""
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
ua, ub = uniforms(W, H, (3.0, 2.0, 3.0)), uniforms(W, H, (-3.5, 1.5, 2.0))
for hold in (1, 2, 3, 4, 8):
    seq = ([ua] * hold + [ub] * hold) * 40
    for u in seq[:4 * hold]: t.set_uniforms(u); t.render()
    t.sync(); t0 = time.perf_counter()
    for u in seq: t.set_uniforms(u); t.render()
    t.sync(); dt = (time.perf_counter() - t0) / len(seq)
    print(f"two cameras, {hold} frames each in turn: {dt*1e3:.3f} ms/frame", flush=True)
