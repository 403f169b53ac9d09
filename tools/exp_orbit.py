"""Frame time of a camera orbiting the terrain (1920 x 1080, grid 2048) at different speeds: what stale scheduling feedback costs
and where the plan switches from overlapping the previous frame to waiting for its tile times (vf_hip.hip: kFreshFeedbackPx)."""
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
for nposes, rep in ((64, 1), (96, 1), (128, 1), (192, 1), (256, 1), (512, 1), (64, 2), (1, 1)):
    eyes = [(3 * math.sqrt(2) * math.cos(2 * math.pi * k / nposes), 2.0, 3 * math.sqrt(2) * math.sin(2 * math.pi * k / nposes)) for k in range(nposes)]
    us = [uniforms(W, H, e) for e in eyes for _ in range(rep)]
    for k in range(16): t.set_uniforms(us[k % len(us)]); t.render()
    t.sync(); t0 = time.perf_counter(); frames = 256
    for k in range(frames): t.set_uniforms(us[k % len(us)]); t.render()
    t.sync(); dt = (time.perf_counter() - t0) / frames
    print(f"{nposes} poses on the orbit, each rendered {rep}x in a row: {dt*1e3:.3f} ms/frame", flush=True)
