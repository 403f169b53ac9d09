#!/usr/bin/env python3
"""The reference's actual usage, timed in a COLD process: construct -> set the heights -> render once -> PNG
(src/terrain/mod.rs:259-407 `new`, src/scene/mod.rs:226-276 `set_height_from_r32f`, src/terrain/mod.rs:410-491 `render_png`).

Prints one JSON object (host wall clock, ms).  bench.py runs this as a child process and files it under
api_latency_ms.one_shot; the first object of the process also pays for the HIP runtime and the context (reported apart as
`runtime_init`: a device query made before anything is constructed).

usage: one_shot.py [W H GRID SEED]        (default: 4096 4096 4096 20250816 = BASELINE config 4)"""
import json, os, sys, tempfile, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t_start = time.perf_counter()
import numpy as np   # noqa: E402
import vulkan_forge_amd as vf   # noqa: E402

t_import = time.perf_counter()
W, H, G, seed = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (4096, 4096, 4096, 20250816)
heights = np.random.default_rng(seed).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)


def ms(t0):
    return (time.perf_counter() - t0) * 1e3


out = {"config": f"Scene {W}x{H} grid={G}, R32F {G}x{G} heights rng({seed})", "clock": "host wall clock, one cold process, every step once",
       "import_ms": (t_import - t_start) * 1e3}
t0 = time.perf_counter(); vf.device_probe(); out["runtime_init"] = ms(t0)        # HIP runtime + context: paid once per process
with tempfile.TemporaryDirectory() as tmp:
    png = os.path.join(tmp, "one_shot.png")
    t0 = time.perf_counter(); a = vf.Scene(W, H, grid=G, colormap="viridis"); out["construct"] = ms(t0)
    t0 = time.perf_counter(); a.set_height_from_r32f(heights); out["set_height"] = ms(t0)
    t0 = time.perf_counter(); a.render_png(png); out["first_render_png"] = ms(t0)
    t0 = time.perf_counter(); a.render_png(png); out["second_render_png"] = ms(t0)
    t0 = time.perf_counter(); a.render_png(png); out["third_render_png"] = ms(t0)
    out["png_bytes"] = os.path.getsize(png)
    del a
    # a second object of the same process (the runtime is warm): what its first frame costs as an array
    t0 = time.perf_counter(); b = vf.Scene(W, H, grid=G, colormap="viridis"); out["construct_second_object"] = ms(t0)
    t0 = time.perf_counter(); b.set_height_from_r32f(heights); out["set_height_second_object"] = ms(t0)
    t0 = time.perf_counter(); rgba = b.render_rgba(); out["first_render_rgba"] = ms(t0)
    t0 = time.perf_counter(); rgba2 = b.render_rgba(); out["second_render_rgba"] = ms(t0)
    t0 = time.perf_counter(); rgba3 = b.render_rgba(); out["third_render_rgba"] = ms(t0)
    t0 = time.perf_counter(); rgba4 = b.render_rgba(); out["fourth_render_rgba"] = ms(t0)
    # a caller that KEEPS its frames gets page-locked arrays for three of them (9 ms to make each), ordinary ones after that ...
    kept = [rgba, rgba2, rgba3, rgba4]
    out["render_rgba_keeping_every_frame"] = []
    for _ in range(3):
        t0 = time.perf_counter(); kept.append(b.render_rgba()); out["render_rgba_keeping_every_frame"].append(ms(t0))
    # ... and one that drops them renders into the pool's buffers again and again
    del kept, rgba3, rgba4
    out["render_rgba_dropping_the_frame"] = []
    for _ in range(3):
        t0 = time.perf_counter(); f = b.render_rgba(); out["render_rgba_dropping_the_frame"].append(ms(t0)); del f
    out["frames_equal"] = bool(np.array_equal(rgba, rgba2))
    out["covered_fraction"] = float((rgba.reshape(-1, 4) != rgba[0, 0]).any(axis=1).mean())
    t0 = time.perf_counter(); del b; out["destroy"] = ms(t0)
out["one_shot_total"] = out["construct"] + out["set_height"] + out["first_render_png"]
print(json.dumps(out), flush=True)
