"""Parity soak on the GPU box: many seeded random frames (sizes, grids, cameras, clip planes, exaggeration, colormaps, both
shade modes) rendered by the HIP path through the C-ABI and by the CPU oracle; reports every case whose visibility is not
bit-exact, whose EXACT-precision RGBA differs from the oracle's at all beyond the stated 1 LSB, or whose FAST-precision (default)
RGBA is more than 1 LSB from it -- with the histogram of the FAST path's differences.  Each case renders several frames on one
handle, so the compared frames are planned with scheduling feedback (strips, heaviest-first order).  Test infrastructure, like
tests/: the oracle is the checker here, nothing of it is shipped.  A script (python tests/soak_parity.py) and, for a bounded
slice, a function tests/test_gpu_soak.py calls under pytest -m gpu.

usage: soak_parity.py [first_seed] [cases] [time_budget_s]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def run(first=5000, cases=200, budget=400.0, huge=False, verbose=True):
    import oracle
    from vulkan_forge_amd import cabi
    luts = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "colormaps_rgba8.npz"))
    t0 = time.time()
    bad, done, worst, worst_fast, shards = [], 0, 0, 0, 0
    hist = np.zeros(4, np.int64)                               # FAST vs oracle, channel values differing by 0, 1, 2, > 2 LSB
    say = (lambda *a: print(*a, flush=True)) if verbose else (lambda *a: None)
    for seed in range(first, first + cases):
        if time.time() - t0 > budget:
            break
        rng = np.random.default_rng(seed)
        big = rng.random() < 0.15
        W, H = (int(rng.integers(300, 1100)), int(rng.integers(200, 900))) if big else (int(rng.integers(1, 400)), int(rng.integers(1, 300)))
        G = int(rng.choice([128, 200, 256, 384]) if big else rng.choice([2, 3, 5, 9, 16, 17, 33, 64, 96, 130]))
        if huge:                                               # VF_SOAK_HUGE=1: frames and grids of the BASELINE configurations' size
            W, H = int(rng.integers(1500, 4097)), int(rng.integers(1000, 4097))
            G = int(rng.choice([512, 1024, 1500, 2048, 3000, 4096]))
        tex = (int(rng.integers(1, 300)), int(rng.integers(1, 300)))
        h = (rng.random(tex, dtype=np.float32) - np.float32(0.5)) * np.float32(rng.choice([0.0, 0.2, 0.5, 1.0, 3.0]))
        r = float(rng.choice([0.05, 0.6, 2.0, 3.0, 4.5, 9.0]))
        th, ph = rng.uniform(0, 2 * math.pi), rng.uniform(-0.6, 1.5)
        eye = (r * math.cos(th) * math.cos(ph), r * math.sin(ph), r * math.sin(th) * math.cos(ph))
        target = tuple(float(v) for v in rng.uniform(-0.4, 0.4, 3))
        fovy = float(rng.choice([20.0, 45.0, 60.0, 120.0, 170.0]))
        znear = float(rng.choice([1e-3, 0.1, 0.5 * r]))
        zfar = float(rng.choice([r + 0.3, 100.0, 1e4]))
        try:
            u = oracle.look_at_uniforms(1, W, H, eye, target, (0.0, 1.0, 0.0), fovy, znear, zfar)
        except Exception:
            continue                                           # degenerate camera (eye == target direction parallel to up, ...)
        u[38] = float(rng.choice([1.0, 1.0, 0.0, 8.0, -2.0]))
        u[36] = float(rng.choice([1.0, 1.0, 0.3, 2.5]))
        cmap = str(rng.choice(["viridis", "magma", "terrain"]))
        mode = int(rng.random() < 0.2)
        ref_rgba, ref_vis = oracle.render_terrain(u, W, H, G, h, luts[cmap], nthreads=min(16, oracle.max_threads()), shade_mode=mode)
        t = cabi.Terrain(W, H, G, luts[cmap])
        try:
            t.set_uniforms(u); t.set_shade_mode(mode); t.set_height(h)
            t.set_raster_groups(int(rng.integers(-1, 2)))  # either line loop of the raster stage, or the handle's own choice
            for _ in range(8 if huge else 4): t.render()       # default precision (FAST)
            fast = t.read_rgba()
            t.set_shade_precision(0)                           # EXACT
            t.render()
            rgba = t.read_rgba(); vis = t.read_visibility()
            shard_bad = ""
            if rng.random() < 0.25:                            # the same frame from N ranks' shards, one after another on this GPU
                n = int(rng.choice([2, 3, 4, 5, 8]))
                whole = rgba
                if rng.random() < 0.5:                         # ... in the default (FAST) arithmetic half of the time: the same bytes from every cut
                    t.set_shade_precision(1); whole = fast
                if rng.random() < 0.5:
                    band = int(rng.choice([64, 128]))
                    out = np.zeros_like(rgba)
                    for rk in range(n):
                        t.set_shard(rk, n, band)
                        for _ in range(3): t.render()
                        rows = np.flatnonzero(((np.arange(H) // band) % n) == rk)
                        loc = t.read_rgba()
                        out[rows] = loc
                    if not np.array_equal(out, whole): shard_bad = f"bands n={n} band={band} {'FAST' if whole is fast else 'EXACT'}"
                else:
                    skew = int(rng.choice([0, 0, 1, 3, 5, 7]))
                    out = np.zeros_like(rgba)
                    for rk in range(n):
                        t.set_tile_shard(rk, n, skew)
                        for _ in range(3): t.render()
                        tiles = t.read_tiles()
                        for k, (tx, ty) in enumerate(cabi.tile_layout(W, H, rk, n, skew, lib=t.lib)):
                            hh, ww = min(64, H - ty * 64), min(64, W - tx * 64)
                            out[ty * 64:ty * 64 + hh, tx * 64:tx * 64 + ww] = tiles[k][:hh, :ww]
                    if not np.array_equal(out, whole): shard_bad = f"tiles n={n} skew={skew} {'FAST' if whole is fast else 'EXACT'}"
                shards += 1
        finally:
            t.close()
        if shard_bad:
            bad.append((seed, W, H, G, -1, -1))
            say(f"SHARD MISMATCH seed={seed} {W}x{H} grid={G}: {shard_bad}")
        nv = int((vis != ref_vis).sum())
        d = int(np.abs(rgba.astype(np.int16) - ref_rgba.astype(np.int16)).max(initial=0))
        df = np.abs(fast.astype(np.int16) - ref_rgba.astype(np.int16))
        hist += np.bincount(np.minimum(df, 3).ravel(), minlength=4)
        dfm = int(df.max(initial=0))
        worst = max(worst, d); worst_fast = max(worst_fast, dfm)
        done += 1
        if nv or d > 1 or dfm > 1:
            bad.append((seed, W, H, G, nv, max(d, dfm)))
            say(f"MISMATCH seed={seed} {W}x{H} grid={G} mode={mode}: visibility differs at {nv} pixels, RGBA max diff exact {d} / fast {dfm}")
        if done % (5 if huge else 25) == 0:
            say(f"{done} cases, {len(bad)} mismatches, worst RGBA diff exact {worst} / fast {worst_fast} LSB, {time.time()-t0:.0f} s")
    tot = max(int(hist.sum()), 1)
    summary = (f"soak: {done} cases from seed {first} ({shards} of them also rendered as 2..8 band or tile shards and stitched): {len(bad)} mismatches; "
               f"worst RGBA difference EXACT {worst} LSB, FAST {worst_fast} LSB; FAST vs oracle over {tot} channel values: 0 LSB {hist[0]}, "
               f"1 LSB {hist[1]} ({100.0 * hist[1] / tot:.4f} %), 2 LSB {hist[2]}, more {hist[3]}; {time.time()-t0:.0f} s")
    say(summary)
    return {"cases": done, "bad": bad, "worst_exact": worst, "worst_fast": worst_fast, "hist": hist.tolist(), "shards": shards, "summary": summary}


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    budget = float(sys.argv[3]) if len(sys.argv) > 3 else 400.0
    res = run(first, cases, budget, huge=os.environ.get("VF_SOAK_HUGE") == "1")
    sys.exit(1 if res["bad"] else 0)
