"""First GPU sanity run: parity of a handful of configs vs the oracle + crude timing (not a test, a probe)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
import vulkan_forge_amd as vf

print(vf.enumerate_adapters())
luts = np.load("tests/golden/colormaps_rgba8.npz")

def check(kind, W, H, G, height=None, cam=None, cmap="viridis"):
    cls = vf.TerrainSpike if kind == 0 else vf.Scene
    obj = cls(W, H, grid=G, colormap=cmap)
    u = oracle.default_uniforms(kind, W, H)
    hh = oracle.SPIKE_DUMMY_HEIGHT if kind == 0 else oracle.SCENE_DUMMY_HEIGHT
    if height is not None:
        obj.set_height_from_r32f(height); hh = height
    if cam is not None:
        obj.set_camera_look_at(*cam); u = oracle.look_at_uniforms(kind, W, H, *cam)
    assert np.array_equal(u, obj.debug_uniforms_f32()), (u, obj.debug_uniforms_f32())
    t = time.time(); rgba = obj.render_rgba(); vis = obj.debug_visibility(); tg = time.time() - t
    t = time.time(); r_rgba, r_vis = oracle.render_terrain(u, W, H, G, hh, luts[cmap], nthreads=8); tc = time.time() - t
    nv = int((vis != r_vis).sum()); d = np.abs(rgba.astype(int) - r_rgba.astype(int)); 
    print(f"kind={kind} {W}x{H} g={G} cov={(r_vis>0).mean():.3f} vis_mismatch={nv} rgba_maxdiff={d.max()} n_diff_px={(d.max(axis=2)>0).sum()} gpu={tg*1e3:.1f}ms cpu={tc*1e3:.1f}ms", flush=True)
    return nv == 0 and d.max() == 0

ok = True
ok &= check(0, 160, 120, 48)
ok &= check(0, 800, 600, 128)
ok &= check(1, 320, 240, 64)
ok &= check(0, 64, 48, 2)
ok &= check(0, 640, 480, 8, cmap="magma")
rng = np.random.default_rng(20250815)
h = (rng.random((256, 256), dtype=np.float32) * 0.5 - 0.25)
ok &= check(1, 640, 360, 256, height=h)
ok &= check(1, 640, 360, 256, height=h, cam=((0.0, 2.2, 0.0), (0, 0, 0), (0, 0, -1), 60.0, 0.1, 100.0), cmap="terrain")
# camera inside the terrain bounds: near-plane clipping + big triangles
ok &= check(1, 320, 240, 32, height=h, cam=((0.2, 0.3, 0.4), (0.0, 0.0, 0.0), (0, 1, 0), 70.0, 0.1, 100.0))
ok &= check(0, 320, 240, 16, cam=((0.5, 0.05, 0.5), (0.0, 0.2, 0.0), (0, 1, 0), 90.0, 0.05, 50.0))
tri = vf.render_triangle_rgba(256, 256); rt = oracle.render_triangle(256, 256)
print("triangle equal:", np.array_equal(tri, rt)); ok &= np.array_equal(tri, rt)
xy, uv, idx = vf.grid_generate(257, 129, (0.37, 1.9)); oxy, ouv, oidx = oracle.grid_generate(257, 129, (0.37, 1.9))
g_ok = np.array_equal(xy.view(np.uint32), oxy.view(np.uint32)) and np.array_equal(uv.view(np.uint32), ouv.view(np.uint32)) and np.array_equal(idx, oidx)
print("grid_generate bit-exact:", g_ok); ok &= g_ok
print("ALL OK" if ok else "MISMATCHES")

# crude timing at C3 / C4 sizes
for (W, H, G, seed) in ((1920, 1080, 1024, 20250815), (4096, 4096, 4096, 20250816)):
    rng = np.random.default_rng(seed)
    hh = rng.random((G, G), dtype=np.float32) * 0.5 - 0.25
    s = vf.Scene(W, H, grid=G); s.set_height_from_r32f(hh); s.enable_timing(True)
    for cam in (None, ((0.0, 2.2, 0.0), (0, 0, 0), (0, 0, -1), 60.0, 0.1, 100.0)):
        if cam: s.set_camera_look_at(*cam)
        for _ in range(3): s.render_rgba()
        print(W, H, G, "cam", "default" if cam is None else "fill", s.last_timings(), flush=True)
