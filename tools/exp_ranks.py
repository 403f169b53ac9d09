"""Steady-state frame period of every rank of an N-GPU tile shard, one rank after another on this GPU (frames rendered back to back, wall
clock over 40 frames, the better of two runs): max over ranks ~ the parallel frame time without the exchange.  usage: exp_ranks.py [camera] [N:skew[:stripe_log2[:b]] ...]
(stripe_log2 "d" = the default of vulkan_forge_amd/dist.py::default_stripe_log2: a period of eight tile columns; a trailing ":b" = the
stripes dealt by their measured times -- the whole frame's per-tile times summed per stripe, vf_balance_stripes -- instead of round-robin)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vulkan_forge_amd as vf
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
W = H = G = 4096
h = np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
t = cabi.Terrain(W, H, G, vf.colormap_rgba8("viridis")); t.set_height(h)
cam = sys.argv[1] if len(sys.argv) > 1 else "default"
from vulkan_forge_amd import dist as vdist
def parse(a):
    f = a.split(":")
    n, skew = int(f[0]), int(f[1])
    sh = 0 if len(f) < 3 else (vdist.default_stripe_log2(n, W // 64) if f[2] == "d" else int(f[2]))
    return n, skew, sh, len(f) > 3 and f[3] == "b"
layouts = [parse(a) for a in sys.argv[2:]] or [(1, 0, 0, False), (2, 0, 2, False), (4, 0, 1, False), (8, 0, 0, False), (8, 3, 0, False)]
t.set_uniforms(b.camera_uniforms(cam, W, H))
base = None
col_ms = None
for n, skew, sh, bal in layouts:
    per = []
    word = vdist.layout_code(skew, sh)
    if bal:                                               # what the N ranks would agree on: per-stripe times of the frame (here: from the one-GPU handle)
        if col_ms is None:
            t.set_shard(0, 1, 64)
            for _ in range(30): t.render()
            col_ms = t.tile_times().reshape(H // 64, W // 64).sum(axis=0)
        word = vdist.balanced_layout(col_ms.reshape(-1, 1 << sh).sum(axis=1), n, sh)
    for r in range(n):
        if n == 1: t.set_shard(0, 1, 64)
        else: t.set_tile_shard(r, n, word)
        for _ in range(30): t.render()
        best = 1e9
        for _ in range(2):                                 # the better of two runs of 40 frames: now and then a run reads 20-30 % high on this pool (round 5: 0.548 ms once for a rank that takes 0.43; not reproduced)
            t.sync(); t0 = time.perf_counter()
            for _ in range(40): t.render()
            t.sync(); best = min(best, (time.perf_counter() - t0) / 40 * 1e3)
        per.append(best)
    if n == 1: base = per[0]
    print(f"{cam:8s} N={n} skew={skew} stripes of {1 << sh}{' dealt by measured times' if bal else ''}: frame period per rank max {max(per):.3f} min {min(per):.3f} ms" + (f"  -> {base / max(per):.2f}x one GPU (compute only)" if base and n > 1 else ""), flush=True)
