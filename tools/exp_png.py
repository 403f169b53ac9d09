"""End-to-end render_png / render_rgba latency at C4: frame + read-back + PNG encode."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vulkan_forge_amd as vf
W = H = G = int(os.environ.get("VF_SIZE", 4096))
s = vf.Scene(W, H, grid=G)
s.set_height_from_r32f(np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25))
d = tempfile.mkdtemp()
for threads in ("1", "4", "16"):
    os.environ["VF_PNG_THREADS"] = threads
    ts = []
    for k in range(4):
        t0 = time.perf_counter(); s.render_png(os.path.join(d, "a.png")); ts.append(time.perf_counter() - t0)
    print(f"render_png {W}x{H} threads={threads}: {min(ts)*1e3:.1f} ms (file {os.path.getsize(os.path.join(d, 'a.png'))/1e6:.1f} MB)", flush=True)
ts = []
for k in range(4):
    t0 = time.perf_counter(); a = s.render_rgba(); ts.append(time.perf_counter() - t0)
print(f"render_rgba -> numpy: {min(ts)*1e3:.1f} ms")
t0 = time.perf_counter(); b = vf._ext._encode_png_rgba8(a); print(f"host-only encode (CPU filter + deflate, 16 threads): {(time.perf_counter()-t0)*1e3:.1f} ms")
