"""What the first calls of a cold process cost below the host module: render + wait, then the PNG scanline read-back, four times.
(cabi.read_png_scanlines returns a COPY of the 67 MB: 6.7 ms of every read-back figure printed here are that copy; the host module
deflates from the handle's buffer in place.)  Round 5, C4: second render 17-20 ms = the context's two side streams being made,
second read-back 18 ms = page-locking the handle's scanline buffer; from the third call on 0.9 + 1.4 ms."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vulkan_forge_amd as vf
from vulkan_forge_amd import cabi
W=H=G=4096
h = np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py")); b_ = importlib.util.module_from_spec(spec); spec.loader.exec_module(b_)
t = cabi.Terrain(W, H, G, vf.colormap_rgba8("viridis")); t.set_height(h); t.set_uniforms(b_.camera_uniforms("default", W, H))
def ms(t0): return round((time.perf_counter()-t0)*1e3, 2)
for k in range(4):
    t0 = time.perf_counter(); t.render(); t.sync(); a = ms(t0)
    t0 = time.perf_counter(); s = t.read_png_scanlines(); b = ms(t0)
    print(f"call {k}: render+sync {a} ms, read_png_scanlines {b} ms", flush=True)
