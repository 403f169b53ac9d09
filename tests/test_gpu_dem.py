"""Renderer DEM path on the GPU: the reference's tests re-expressed (tests/test_height_upload.py,
test_dem_stats.py, test_dem_normalization.py, test_tonemap.py:26-31) plus parity with the oracle.

Tolerances: min/max, the ingest, minmax normalisation and the texture round trip are bit-exact.  mean/std (and hence
zscore) are float reductions: the reference adds in f32 in index order, the HIP path carries the sums in FP64, so they
agree within N * 2^-24 relative (stated per test)."""
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import vulkan_forge as vf          # noqa: E402


def ramp(dtype, shape):
    h, w = shape
    return np.linspace(0.0, 1.0, num=h * w, dtype=dtype).reshape(shape)


def test_add_terrain_float32_and_stats():
    r = vf.Renderer(16, 16)
    r.add_terrain(ramp(np.float32, (4, 4)), (1.0, 1.0), 2.0, colormap="viridis")
    mn, mx, mean, std = r.terrain_stats()
    assert mn == pytest.approx(0.0) and mx == pytest.approx(2.0) and mean == pytest.approx(1.0)
    assert std == pytest.approx(np.std(np.linspace(0.0, 2.0, num=16, dtype=np.float32)), rel=1e-3)


def test_add_terrain_float64_and_normalize():
    r = vf.Renderer(8, 8)
    r.add_terrain(ramp(np.float64, (3, 3)), (1.0, 1.0), 1.0, colormap="magma")
    r.normalize_terrain("minmax", range=(10.0, 20.0), eps=None)
    mn, mx, _, _ = r.terrain_stats()
    assert mn == pytest.approx(10.0, rel=1e-5) and mx == pytest.approx(20.0, rel=1e-5)
    r.normalize_terrain("zscore", range=None, eps=1e-6)
    _, _, mean, std = r.terrain_stats()
    assert abs(mean) < 1e-5 and std == pytest.approx(1.0, rel=1e-3)


def test_upload_readback_patch_and_errors():
    r = vf.Renderer(32, 32)
    with pytest.raises(RuntimeError, match=re.escape("no terrain uploaded; call add_terrain() first")):
        r.upload_height_r32f()
    with pytest.raises(RuntimeError, match="no terrain uploaded"):
        r.terrain_stats()
    assert (r.debug_read_height_patch(0, 0, 3, 2) == 0).all()             # no texture yet: zeros (src/lib.rs:580-588)
    r.add_terrain(ramp(np.float32, (5, 5)), (1.0, 1.0), 1.0, colormap="terrain")
    with pytest.raises(RuntimeError, match="no height texture uploaded"):
        r.read_full_height_texture()
    r.upload_height_r32f()
    full = r.read_full_height_texture()
    assert full.shape == (5, 5) and full.dtype == np.float32 and np.array_equal(full, ramp(np.float32, (5, 5)))
    assert np.array_equal(r.debug_read_height_patch(1, 1, 3, 3), full[1:4, 1:4])
    r.upload_height_r32f()
    assert np.array_equal(r.read_full_height_texture(), full)
    with pytest.raises(RuntimeError, match=re.escape("requested patch exceeds texture bounds in x: x+w (7) > width (5)")):
        r.debug_read_height_patch(4, 0, 3, 4)
    with pytest.raises(RuntimeError, match="exceeds texture bounds in y"):
        r.debug_read_height_patch(0, 3, 4, 3)
    with pytest.raises(RuntimeError, match="patch dimensions must be > 0"):
        r.debug_read_height_patch(0, 0, 0, 1)
    r.normalize_terrain("minmax", range=(10.0, 20.0), eps=None)
    r.upload_height_r32f()
    after = r.read_full_height_texture()
    assert after.shape == full.shape and after.min() == pytest.approx(10.0) and after.max() == pytest.approx(20.0)


@pytest.mark.parametrize("w,h", [(7, 5), (64, 48), (255, 3), (33, 33), (61, 17)])
def test_height_roundtrip_odd_sizes(w, h):
    r = vf.Renderer(max(w, 16), max(h, 16))
    hm = np.random.RandomState(42).rand(h, w).astype(np.float32)
    r.add_terrain(hm, spacing=(1.0, 1.0), exaggeration=1.0, colormap="viridis")
    r.upload_height_r32f()
    back = r.read_full_height_texture()
    assert back.shape == (h, w) and np.array_equal(back, hm)              # rtol 1e-6 in the reference; exact here


def test_argument_errors():
    r = vf.Renderer(16, 16)
    hm = ramp(np.float32, (4, 4))
    for args, msg in ((((hm, (0.0, 1.0), 1.0, "viridis")), "spacing components must be > 0"),
                      ((hm, (1.0, 1.0), 0.0, "viridis"), "exaggeration must be > 0"),
                      ((hm.astype(np.int32), (1.0, 1.0), 1.0, "viridis"), "heightmap must be a 2-D NumPy array of dtype float32 or float64"),
                      # a non-contiguous float32 array fails the reference's f32 attempt AND its f64 downcast: the dtype message
                      # (src/lib.rs:351-372); a non-contiguous float64 array reports the layout (:374-376)
                      ((hm[:, ::2], (1.0, 1.0), 1.0, "viridis"), "heightmap must be a 2-D NumPy array of dtype float32 or float64"),
                      ((hm.astype(np.float64)[:, ::2], (1.0, 1.0), 1.0, "viridis"), "heightmap must be C-contiguous (row-major)"),
                      ((hm, (1.0, 1.0), 1.0, "invalid_colormap"), "Unknown colormap 'invalid_colormap'. Supported: viridis, magma, terrain")):
        with pytest.raises(RuntimeError, match=re.escape(msg)):
            r.add_terrain(*args)
    r.add_terrain(hm, (1.0, 1.0), 1.0, "viridis")
    with pytest.raises(RuntimeError, match="mode must be 'minmax' or 'zscore'"):
        r.normalize_terrain("x")
    r.normalize_terrain("MinMax")                                          # case-insensitive (src/lib.rs:480)
    r.set_height_range(-5.0, 40.0)                                         # tests/test_dem_stats.py:12-18
    for bad in ((1.0, 1.0), (2.0, -3.0), (float("nan"), 1.0)):
        with pytest.raises(ValueError):
            r.set_height_range(*bad)
    r.set_sun(45.0, 30.0)                                                  # tests/test_tonemap.py:26-31
    with pytest.raises(ValueError, match="exposure must be > 0"):
        r.set_exposure(0.0)
    with pytest.raises(ValueError, match="angles must be finite"):
        r.set_sun(float("inf"), 0.0)
    r.set_exposure(1.25)
    for cm in ("viridis", "magma", "terrain"):                             # tests/test_colormap.py:105-126
        r.add_terrain(np.random.rand(64, 64).astype(np.float32), (1.0, 1.0), 1.0, cm)
        assert len(r.terrain_stats()) == 4


def test_failed_add_terrain_keeps_the_previous_terrain():
    """src/lib.rs:398-416: an unknown colormap is reported after the height range was stored but before self.terrain is replaced."""
    r = vf.Renderer(16, 16)
    first = ramp(np.float32, (6, 5))
    r.add_terrain(first, (1.0, 1.0), 1.0, "viridis")
    before = r.terrain_stats()
    with pytest.raises(RuntimeError, match="Unknown colormap"):
        r.add_terrain(np.full((9, 9), 77.0, np.float32), (1.0, 1.0), 1.0, "nope")
    assert r.terrain_stats() == before
    r.upload_height_r32f()
    assert np.array_equal(r.read_full_height_texture(), first)
    fresh = vf.Renderer(16, 16)
    with pytest.raises(RuntimeError, match="Unknown colormap"):
        fresh.add_terrain(first, (1.0, 1.0), 1.0, "nope")
    with pytest.raises(RuntimeError, match="no terrain uploaded"):
        fresh.terrain_stats()


@pytest.mark.parametrize("shape,dtype,ex", [((64, 64), np.float32, 1.0), ((300, 1000), np.float32, 2.5), ((129, 257), np.float64, 0.5),
                                            ((2048, 2048), np.float32, 1.0)])
def test_parity_with_oracle(oracle, shape, dtype, ex):
    rng = np.random.default_rng(shape[0])
    hm = (rng.random(shape) * 3 - 1).astype(dtype)
    ref = oracle.dem_ingest(hm, ex)
    r = vf.Renderer(16, 16)
    r.add_terrain(hm, (1.0, 1.0), ex, "viridis")
    r.upload_height_r32f()
    assert np.array_equal(r.read_full_height_texture().view(np.uint32), ref.view(np.uint32))      # ingest: bit-exact
    st, ost = r.terrain_stats(), oracle.dem_stats(ref)
    n = ref.size
    assert st[0] == ost[0] and st[1] == ost[1]                                                    # min / max: exact
    tol = max(n * 2.0 ** -24, 1e-6)                                                               # sequential f32 sum error bound
    assert st[2] == pytest.approx(float(ref.astype(np.float64).mean()), abs=1e-6 * (1 + abs(ost[2])))
    assert st[2] == pytest.approx(ost[2], abs=tol * max(abs(ost[0]), abs(ost[1])))
    assert st[3] == pytest.approx(float(ref.astype(np.float64).std()), rel=1e-5)
    assert st[3] == pytest.approx(ost[3], rel=max(tol, 1e-5))
    r.normalize_terrain("minmax", range=(-2.0, 7.0))
    r.upload_height_r32f()
    got = r.read_full_height_texture()
    want = oracle.dem_normalize(ref, "minmax", out_range=(-2.0, 7.0))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))                              # minmax: bit-exact
