"""Blocks PULLED from the tiles' work lists against blocks drawn (live after the late cull): the library given must be a -DVF_DBG_PULLS build (round 6: apply tools/experiments/r06_kernel_laboratory.patch first)
(its per-item count is pulls), the default library gives the drawn ones.  usage: exp_pulls.py build/variants/libvf_pulls.so"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vulkan_forge_amd as vf
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
W = H = G = 4096
h = np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
for lib, what in ((None, "drawn"), (sys.argv[1], "pulled")):
    t = cabi.Terrain(W, H, G, vf.colormap_rgba8("viridis"), lib=cabi.load(lib) if lib else None); t.set_height(h)
    out = []
    for cam, shard in (("default", None), ("fill", None), ("default", (2, 8)), ("fill", (2, 8))):
        t.set_uniforms(b.camera_uniforms(cam, W, H))
        if shard: t.set_tile_shard(shard[0], shard[1], 0)
        else: t.set_shard(0, 1, 64)
        for _ in range(30): t.render()
        t.enable_timing(True); t.render(); t.render(); it = t.item_stats(); tm = t.timings(); t.enable_timing(False)
        out.append(f"{cam}{' rank' if shard else ''}: {int(it[:, 1].sum())} (pairs {tm['blocks_rasterised']})")
    print(f"{what:7s} " + " | ".join(out), flush=True)
    t.close()
