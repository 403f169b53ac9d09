import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from vulkan_forge_amd import cabi
lib = cabi.load(sys.argv[1])
luts = np.load("tests/golden/colormaps_rgba8.npz")
for (kind, W, H, G, tex) in ((0, 160, 120, 48, None), (1, 320, 240, 64, None), (1, 640, 360, 256, 256), (0, 800, 600, 128, None)):
    u = oracle.default_uniforms(kind, W, H)
    h = oracle.SPIKE_DUMMY_HEIGHT if kind == 0 else oracle.SCENE_DUMMY_HEIGHT
    if tex: h = np.random.default_rng(1).random((tex, tex), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
    t = cabi.Terrain(W, H, G, luts["viridis"], lib=lib); t.set_uniforms(u); t.set_height(h); t.render()
    rgba = t.read_rgba(); vis = t.read_visibility()
    rr, rv = oracle.render_terrain(u, W, H, G, h, luts["viridis"], nthreads=8)
    bad = vis != rv
    ys, xs = np.nonzero(bad)
    print(kind, W, H, G, "vis mismatches", int(bad.sum()), "rgba maxdiff", int(np.abs(rgba.astype(int) - rr.astype(int)).max()))
    if bad.any():
        print("   x range", xs.min(), xs.max(), "y range", ys.min(), ys.max(), "x%64 hist", np.bincount(xs % 64, minlength=64).tolist())
        print("   got==0:", int((vis[bad] == 0).sum()), "got<ref:", int((vis[bad] < rv[bad]).sum()), "got>ref:", int((vis[bad] > rv[bad]).sum()))
    t.close()
