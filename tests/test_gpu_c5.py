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
