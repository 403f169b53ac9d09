"""BASELINE config 5 at its stated size: 64 camera look-ats on the default camera's orbit over one grid=2048 terrain,
1920x1080.  Eight of the 64 poses against the oracle, rendered back to back on ONE handle in orbit order
with the poses between them rendered too, so that what is tested is the moving-camera plan path (feedback from the previous
pose, dilated weights, strips) the pose batch really runs through -- not eight cold frames."""
import math

import numpy as np
import pytest

from conftest import heightmap

pytestmark = pytest.mark.gpu


def orbit_pose(k):
    th = 2 * math.pi * k / 64
    return ((3 * math.sqrt(2) * math.cos(th), 2.0, 3 * math.sqrt(2) * math.sin(th)), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 45.0, 0.1, 100.0)


def test_c5_orbit_poses_full_size(oracle, luts):
    from vulkan_forge_amd import cabi
    W, H, G = 1920, 1080, 2048
    h = heightmap(20250817, G)
    t = cabi.Terrain(W, H, G, luts["viridis"])
    try:
        t.set_height(h)
        t.set_shade_precision(0)                                     # EXACT: the poses are compared byte for byte with the oracle
        t.enable_timing(True)
        checked, strips = 0, 0
        for k in range(0, 60):                                       # poses 0..59 in order; every 8th one (from 3) is compared
            u = oracle.look_at_uniforms(1, W, H, *orbit_pose(k))
            t.set_uniforms(u)
            t.render()
            if k % 8 == 3:
                rgba = t.read_rgba()
                strips += int((t.item_stats()[:, 0] >> 24).astype(bool).sum())
                ref_rgba, ref_vis = oracle.render_terrain(u, W, H, G, h, luts["viridis"], nthreads=min(16, oracle.max_threads()))
                assert np.array_equal(rgba, ref_rgba), k
                # the same pose with the default (FAST) arithmetic, directly against the oracle: within 1 LSB
                t.set_shade_precision(1); t.render(); fast = t.read_rgba(); t.set_shade_precision(0)
                assert int(np.abs(fast.astype(np.int16) - ref_rgba.astype(np.int16)).max()) <= 1, k
                t.enable_timing(False)
                assert np.array_equal(t.read_visibility(), ref_vis), k
                t.enable_timing(True)
                checked += 1
        assert checked == 8
        assert strips > 0                                            # the feedback-driven split took part
    finally:
        t.close()


def test_c5_batch_entry_point_full_size(oracle, luts):
    """vf_terrain_render_batch (BASELINE config 5 as ONE call): the 64 poses of the orbit queued back to back, every pose planned from
    the pose before last through the camera motion (no pose waits for its predecessor); 8 of the 64 outputs against the oracle at
    full size, EXACT arithmetic byte for byte."""
    import ctypes as C
    from vulkan_forge_amd import cabi
    W, H, G = 1920, 1080, 2048
    h = heightmap(20250817, G)
    t = cabi.Terrain(W, H, G, luts["viridis"])
    hip = C.CDLL("libamdhip64.so.7")                              # the runtime libvf_hip.so already loaded (matched by soname)
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; hip.hipFree.argtypes = [C.c_void_p]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    frame_bytes = W * H * 4
    slab = C.c_void_p()
    assert hip.hipMalloc(C.byref(slab), 64 * frame_bytes) == 0
    def frame(k):
        out = np.empty((H, W, 4), np.uint8)
        assert hip.hipMemcpy(out.ctypes.data, slab.value + k * frame_bytes, frame_bytes, 2) == 0      # hipMemcpyDeviceToHost
        return out
    try:
        t.set_height(h)
        t.set_shade_precision(0)
        us = np.stack([oracle.look_at_uniforms(1, W, H, *orbit_pose(k)) for k in range(64)])
        t.render_batch(us, [slab.value + k * frame_bytes for k in range(64)])
        t.sync()
        for k in range(5, 64, 8):
            ref_rgba, _ = oracle.render_terrain(us[k], W, H, G, h, luts["viridis"], nthreads=min(16, oracle.max_threads()), want_vis=False)
            assert np.array_equal(frame(k), ref_rgba), k
        # the same pose as a call of its own gives the same bytes (the batch is the loop, planned ahead)
        t.set_uniforms(us[21]); t.set_output_device(slab.value); t.render(); t.sync()
        assert np.array_equal(frame(0), frame(21))
    finally:
        t.close()
        hip.hipFree(slab)


def test_c5_batch_with_read_back(oracle, luts):
    """vf_terrain_render_batch_host: every frame read back while the next poses are drawn (three device frames in flight)."""
    from vulkan_forge_amd import cabi
    W, H, G = 640, 360, 256
    h = heightmap(20250817, G)
    t = cabi.Terrain(W, H, G, luts["viridis"])
    try:
        t.set_height(h)
        t.set_shade_precision(0)
        us = np.stack([oracle.look_at_uniforms(1, W, H, *orbit_pose(k)) for k in range(0, 64, 4)])
        got = t.render_batch_host(us)
        assert got.shape == (16, H, W, 4)
        for k in (0, 5, 15):
            ref_rgba, _ = oracle.render_terrain(us[k], W, H, G, h, luts["viridis"], want_vis=False)
            assert np.array_equal(got[k], ref_rgba), k
    finally:
        t.close()


def test_scene_render_batch_matches_the_per_pose_loop(tmp_path, oracle, luts):
    """Scene.render_batch(poses) == set_camera_look_at + render_rgba per pose (src/scene/mod.rs:208-224, :278-335); with paths: PNG files."""
    import vulkan_forge_amd as vf
    W, H, G = 1920, 1080, 512                                     # frame-sized: the arrays live in the pinned pool
    h = heightmap(20250817, G)
    sc = vf.Scene(W, H, grid=G, colormap="viridis")
    sc.set_height_from_r32f(h)
    poses = [orbit_pose(k) for k in (0, 9, 18, 40)]
    frames = sc.render_batch(poses)
    assert len(frames) == 4 and frames[0].shape == (H, W, 4) and frames[0].dtype == np.uint8
    assert np.array_equal(sc.render_rgba(), frames[3])            # the camera of the last pose stays
    for k, pose in enumerate(poses):
        sc.set_camera_look_at(*pose)
        assert np.array_equal(sc.render_rgba(), frames[k]), k
    ref, _ = oracle.render_terrain(oracle.look_at_uniforms(1, W, H, *poses[1]), W, H, G, h, luts["viridis"], nthreads=min(16, oracle.max_threads()), want_vis=False)
    assert int(np.abs(frames[1].astype(np.int16) - ref.astype(np.int16)).max()) <= 1            # (default FAST arithmetic: within 1 LSB)
    paths = [str(tmp_path / f"pose{k}.png") for k in range(4)]
    assert sc.render_batch(poses, paths) is None
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(paths[2]).convert("RGBA")), frames[2])
    with pytest.raises(ValueError):
        sc.render_batch(poses, paths[:2])
    assert sc.render_batch([]) == []


def test_bench_c5_workload_runs(tmp_path):
    """bench.py --workload c5 prints one JSON line with the contract's fields (short run, no CPU baseline)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "c5", "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--no-extra"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["scaling"] == "replicas" and d["n_gpus"] == 1 and d["steps"] == 8 and d["unit"] == "Mpix/s"
    assert d["config"]["grid"] == 2048 and (d["config"]["width"], d["config"]["height"]) == (1920, 1080)
    assert d["value"] > 0 and d["ms_per_pose"] > 0 and d["roofline"]["bound"] == "hbm"
