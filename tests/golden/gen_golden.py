#!/usr/bin/env python3
"""Generate tests/golden/frames_oracle.npz -- small rendered frames (inputs + expected RGBA + visibility).

PROVENANCE: these vectors come from THIS repository's CPU oracle (oracle/vf_oracle.c), not from a run of the
reference: the reference cannot be built here (no Rust toolchain / no Vulkan adapter) and holds no golden
images of its own.  They freeze the oracle's output so that (a) a change to the oracle is
noticed, and (b) the GPU tests have committed expected values that do not depend on the oracle build of
the day.  Regenerate only together with a documented convention change.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))     # tests/golden/ -> repository root
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402

luts = np.load(os.path.join(ROOT, "tests", "golden", "colormaps_rgba8.npz"))

def hm(seed, w, h):
    return np.random.default_rng(seed).random((h, w), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)

CASES = {
    # name: (kind, W, H, grid, colormap, lut_is_srgb, height or None, camera or None)
    "spike_64x48_g16_viridis": (0, 64, 48, 16, "viridis", True, None, None),
    "spike_100x75_g24_magma_unorm": (0, 100, 75, 24, "magma", False, None, None),
    "scene_dummy_80x60_g12_terrain": (1, 80, 60, 12, "terrain", True, None, None),
    "scene_h16_96x64_g16_viridis": (1, 96, 64, 16, "viridis", True, hm(11, 16, 16), None),
    "scene_h9x5_fill_72x72_g20": (1, 72, 72, 20, "viridis", True, hm(12, 9, 5), ((0.0, 2.2, 0.0), (0, 0, 0), (0, 0, -1), 60.0, 0.1, 100.0)),
    "scene_clip_64x64_g10": (1, 64, 64, 10, "magma", True, hm(13, 8, 8), ((0.2, 0.3, 0.4), (0, 0, 0), (0, 1, 0), 70.0, 0.1, 100.0)),
}

out = {}
for name, (kind, W, H, G, cmap, srgb, height, cam) in CASES.items():
    u = O.default_uniforms(kind, W, H) if cam is None else O.look_at_uniforms(kind, W, H, *cam)
    h = height if height is not None else (O.SPIKE_DUMMY_HEIGHT if kind == 0 else O.SCENE_DUMMY_HEIGHT)
    lut = luts[cmap] if srgb else O.lut_to_linear_u8(luts[cmap])
    rgba, vis = O.render_terrain(u, W, H, G, h, lut, lut_is_srgb=srgb)
    out[name + "/meta"] = np.array([kind, W, H, G, int(srgb)], np.int32)
    out[name + "/cmap"] = np.array(cmap)
    out[name + "/uniforms"] = u
    out[name + "/height"] = h
    out[name + "/cam"] = np.array(sum(([*c] if isinstance(c, tuple) else [c] for c in cam), []), np.float32) if cam else np.zeros(0, np.float32)
    out[name + "/rgba"] = rgba
    out[name + "/vis"] = vis
    print(name, "coverage", float((vis > 0).mean()))
out["triangle_33x21/rgba"] = O.render_triangle(33, 21)
xy, uv, idx = O.grid_generate(5, 4, (0.37, 1.9))
out["grid_5x4/xy"], out["grid_5x4/uv"], out["grid_5x4/idx"] = xy, uv, idx
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "frames_oracle.npz"), **out)
print("wrote tests/golden/frames_oracle.npz")
