"""The plan's work list as the tile kernel saw it (library built with -DVF_DIAG_ITEM=2: word 1 of the item statistics is the plan's
weight): are the items in descending order of the sort key (k_plan_sort: exponent + five mantissa bits of the weight)?
usage: VF_HIP_LIB=build/variants/libvf_weight.so python tools/exp_plan_order.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vulkan_forge_amd as vf
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
for (W, H, G, cam) in ((4096, 4096, 4096, "default"), (4096, 4096, 4096, "fill"), (1920, 1080, 2048, "default"), (8192, 8192, 2048, "fill")):
    h = np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
    t = cabi.Terrain(W, H, G, vf.colormap_rgba8("viridis")); t.set_height(h); t.set_uniforms(b.camera_uniforms(cam, W, H))
    for f in range(4):
        t.enable_timing(True, stats=True); t.render(); t.sync()
        it = t.item_stats(); t.enable_timing(False)
        w = np.maximum(it[:, 1].astype(np.float32), 1.0)
        key = (w.view(np.uint32) >> 18).astype(np.int64)
        bad = 0
        for base in range(0, len(key), 4096):
            k = key[base:base + 4096]
            bad += int((np.diff(k) > 0).sum())
        print(f"{W}x{H} grid {G} {cam} frame {f}: {len(it)} items, weights {int(w.max())} .. {int(w.min())}, out-of-order neighbours (by key, within 4096-item runs): {bad}", flush=True)
    t.close()
