"""First frames of a fresh handle (the reference's one-shot usage): wall time of three synchronous frames, how frame 0 was cut.
VF_NO_STATIC_PLAN=1 switches the static first-frame estimate off (the round-2 behaviour).  usage: exp_cold.py [default|fill] [N:rank]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vulkan_forge_amd as vf
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
W = H = G = 4096
h = np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
cam = sys.argv[1] if len(sys.argv) > 1 else "default"
shard = tuple(int(v) for v in sys.argv[2].split(":")) if len(sys.argv) > 2 else None
warm = cabi.Terrain(W, H, G, vf.colormap_rgba8("viridis")); warm.set_height(h); warm.set_uniforms(b.camera_uniforms(cam, W, H)); warm.render(); warm.render(); warm.sync()   # (the process's first launches are not the handle's)
for rep in range(3):
    t = cabi.Terrain(W, H, G, vf.colormap_rgba8("viridis"), share_ctx=warm); t.set_height(h); t.set_uniforms(b.camera_uniforms(cam, W, H))
    if shard: t.set_tile_shard(shard[1], shard[0], 0)
    t.sync()
    ms, cuts = [], []
    for f in range(4):
        t.enable_timing(True, stats=(rep == 2))
        t0 = time.perf_counter(); t.render(); t1 = time.perf_counter(); t.sync(); ms.append((time.perf_counter() - t0) * 1e3)
        tm = t.timings()
        if rep == 1: print(f"   frame {f}: host launch {(t1 - t0) * 1e3:.3f} ms; device: boxes..setup-launch {tm['ranges_ms']:.3f}, plan {tm['plan_ms']:.3f}, clear+tile {tm['tile_ms']:.3f}, plan start -> frame done {tm['total_ms']:.3f}")
        if rep == 2:
            it = t.item_stats()
            if os.environ.get("VF_DUMP_WEIGHTS") and f < 2:             # library built with -DVF_DIAG_ITEM=2: word 1 is the plan's weight
                np.save(os.path.join(os.path.dirname(__file__), "..", "gpurun_out", f"items_{cam}_f{f}.npy"), it)
                w, tm = it[:, 1].astype(float), it[:, 3].astype(float)
                order = np.argsort(-tm)[:12]
                print(f"   frame {f}: corr(weight, time) = {np.corrcoef(w, tm)[0, 1]:.3f}; sum w {w.sum():.0f} sum t {tm.sum():.0f}; heaviest items (weight, ticks): " + " ".join(f"({int(w[k])},{int(tm[k])})" for k in order))
            cuts.append((len(it), np.bincount((it[:, 0] >> 24) & 7, minlength=5).tolist(), round(float(it[:, 2].max()) * 1e-5, 3)))
        t.enable_timing(False)
    print(f"{cam} rep {rep}: frames " + " ".join(f"{x:.3f}" for x in ms) + (f"   (items, strips by log2, longest item ms): {cuts}" if cuts else ""), flush=True)
    t.close()
