"""Per-phase cycle breakdown of k_tile (library built with -DVF_PHASE_PROF) on the C4 workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
W = H = G = int(os.environ.get("VF_SIZE", 4096))
if os.environ.get("VF_C5"): W, H, G = 1920, 1080, 2048            # BASELINE config 5's frame; cameras: orbit poses 8 (the default eye) and 0
import vulkan_forge_amd as _vf; lut = _vf.colormap_rgba8("viridis")
h = np.random.default_rng(20250817 if os.environ.get("VF_C5") else 20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
names = ["setup", "pull/cull", "vertex", "classify", "span raster", "completion/rescan", "chunk-end wait", "fragment"]
t = cabi.Terrain(W, H, G, lut, lib=cabi.load(sys.argv[1])); t.set_height(h)
if len(sys.argv) > 3: t.set_tile_shard(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 0)      # one rank of N
for cam in (("pose8", "pose0") if os.environ.get("VF_C5") else ("default", "fill")):
    t.set_uniforms(b.orbit_uniforms(int(cam[4:]), W, H) if cam.startswith("pose") else b.camera_uniforms(cam, W, H))
    for _ in range(24): t.render()
    t.enable_timing(True); t.render(); tm = t.timings(); ph = t.phase_cycles().astype(float); it = t.item_stats(); t.enable_timing(False)
    lanes = ph[34:38]; vsub2 = ph[32:34]; vsub = ph[30:32]; wv = ph[24:30]; sub = ph[16:24]; cnt = ph[8:16]; ph = ph[:8]; ph[0] += sub.sum(); ph[2] += vsub.sum() + vsub2.sum()
    tot = ph.sum()
    live = max(cnt[7], 1)
    print(f"{cam}: tile_ms={tm['tile_ms']:.3f} pairs={tm['blocks_rasterised']} wave-cycles total={tot:.3e} (= {tot/16/2.4e6:.1f} ms of workgroup time at 2.4 GHz)")
    for n, c in zip(names, ph):
        print(f"   {n:20s} {100*c/tot:6.2f} %   {c/max(tm['blocks_rasterised'],1):9.0f} cycles/pair")
    use, _ = t.raster_groups()
    print(f"   line loop ({'with line groups: group-test trips' if use else 'plain: line trips'}={wv[0]/live:.2f}), wave-level executions per pair: lines reaching stage 1={wv[1]/live:.2f}  stage 2={wv[2]/live:.2f}  paint steps={wv[3]/live:.2f};  classification: triangles reaching the occlusion loop={wv[5]/live:.1f}, its wave-level iterations={wv[4]/live:.2f}")
    # round 6: how many of a wave's 64 lanes each part of the block loop keeps busy -- lane-level events / wave-level executions
    div = lambda x, y: x / y if y else float("nan")
    print(f"   LANES busy per wave-level execution: pass A (classification) {div(lanes[1], lanes[2]):.1f} of 64;  pass B set-up (lanes with a triangle) {div(lanes[3], cnt[1]):.1f};  "
          f"line loop: {'group tests' if use else 'line trips'} {div(lanes[0], wv[0]):.1f}, stage 1 {div(cnt[6], wv[1]):.1f}, stage 2 {div(cnt[4], wv[2]):.1f}, paint {div(cnt[5], wv[3]):.1f};  occlusion loop of the classification {div(wv[5], wv[4]):.1f} (triangles per iteration)")
    for n, c in zip(["row list", "candidate tests", "wait for slowest wave", "scan + list fill", "hand-over + pull", "item record + tile state", "row mask", "-"], sub):
        print(f"      set-up: {n:22s} {100*c/tot:6.2f} %")
    print(f"      vertex: record wait {100*vsub[0]/tot:.2f} %  (one empty time stamp: {100*vsub[1]/tot:.2f} % = {vsub[1]/live:.0f} cycles per pair; every phase above holds one per boundary)  wait for the vertex records {100*vsub2[0]/tot:.2f} % = {vsub2[0]/live:.0f} cycles per pair  LDS staging {100*vsub2[1]/tot:.2f} %  alive-list compaction {100*(ph[2]-vsub.sum()-vsub2.sum())/tot:.2f} %")
    print(f"   shader clock during the items: {tot / 16 / (it[:, 3].astype(float).sum() * 1e-8) / 1e9:.3f} GHz (wave cycles / 16 waves / item time)")
    print(f"   live pairs={cnt[7]:.0f}  survivors/pair={cnt[0]/live:.1f}  passB wave-iterations/pair={cnt[1]/live:.2f}  pairs without any open line={100*cnt[2]/live:.1f}%"
          f"  lines/pair={cnt[3]/live:.1f}  solved lines/pair={cnt[4]/live:.1f}  painted px/pair={cnt[5]/live:.1f}  lines with an open pixel in the bbox range/pair={cnt[6]/live:.1f}")
