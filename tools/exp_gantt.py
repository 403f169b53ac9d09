"""When does each work item of a frame start and end?  (library built with -DVF_DIAG_ITEM=3: the item statistics' block column holds the
start tick.)  Prints the frame's schedule: items in flight over time, when the last item of each weight class starts, the idle share.
usage: exp_gantt.py [camera] [rank n [stripe_log2]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vulkan_forge_amd as vf
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
W = H = G = 4096
if os.environ.get("VF_C5"): W, H, G = 1920, 1080, 2048            # BASELINE config 5's frame; camera "poseK" = pose K of the orbit, at rest
cam = sys.argv[1] if len(sys.argv) > 1 else "default"
shard = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else None
h = np.random.default_rng(20250817 if os.environ.get("VF_C5") else 20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
t = cabi.Terrain(W, H, G, vf.colormap_rgba8("viridis")); t.set_height(h); t.set_uniforms(b.orbit_uniforms(int(cam[4:]), W, H) if cam.startswith("pose") else b.orbit_uniforms(0, W, H) if cam.startswith("orbit") else b.camera_uniforms(cam, W, H))
if shard: t.set_tile_shard(shard[0], shard[1], (int(sys.argv[4]) << 16) if len(sys.argv) > 4 else 0)
moving = cam.startswith("orbit")                       # "orbitK": the poses 0 .. K of the orbit one after another (a moving camera); the schedule of pose K
if moving:
    for k in range(64 + int(cam[5:])): t.set_uniforms(b.orbit_uniforms(k % 64, W, H)); t.render()
else:
    for _ in range(30): t.render()
for rep in range(1 if moving else 2):
    if moving: t.set_uniforms(b.orbit_uniforms((int(cam[5:]) + 1) % 64, W, H))
    t.enable_timing(True); t.render(); tm = t.timings(); it = t.item_stats(); t.enable_timing(False)
    start = it[:, 1].astype(np.int64); start -= start.min()
    dur = it[:, 3].astype(np.int64)
    end = start + dur
    order = np.argsort(start)
    span = end.max()
    print(f"{cam} {shard}: tile_ms {tm['tile_ms']:.3f}; {len(it)} items; first start 0, last end {span * 1e-5:.3f} ms; sum of items {dur.sum() * 1e-5:.2f} ms = {dur.sum() / span / 256 * 100:.0f} % of 256 workgroups x span")
    # items in flight at 16 sample points
    for k in range(17):
        x = span * k // 16
        print(f"   t = {x * 1e-5:.3f} ms: {int(((start <= x) & (end > x)).sum()):4d} items in flight, {int((start > x).sum()):4d} not started")
    # the items that end last
    last = np.argsort(-end)[:8]
    for k in last:
        code = int(it[k, 0])
        print(f"   ends {end[k] * 1e-5:.3f}: item {k} (queue position) started {start[k] * 1e-5:.3f} ran {dur[k] * 1e-5:.3f} ms, strips 1/{1 << ((code >> 24) & 7)}")
    # how long after the kernel's first item did the k-th item of the queue start?
    print("   start of queue positions 0, 64, 128, 192, 255, 256, 300, 350, 400:", [f"{start[k] * 1e-5:.3f}" for k in (0, 64, 128, 192, 255, 256, 300, 350, 400) if k < len(it)])
