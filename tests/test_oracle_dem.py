"""Oracle pins for the Renderer DEM path: the known answers of the reference's own tests
(tests/test_height_upload.py:20-49, tests/test_dem_normalization.py, tests/test_dem_stats.py)."""
import numpy as np
import pytest


def ramp(dtype, shape):
    h, w = shape
    return np.linspace(0.0, 1.0, num=h * w, dtype=dtype).reshape(shape)


def test_ingest_and_stats_known_answers(oracle):
    h = oracle.dem_ingest(ramp(np.float32, (4, 4)), 2.0)                 # test_height_upload.py:20-32
    mn, mx, mean, std = oracle.dem_stats(h)
    assert mn == pytest.approx(0.0) and mx == pytest.approx(2.0) and mean == pytest.approx(1.0)
    assert std == pytest.approx(np.std(np.linspace(0.0, 2.0, num=16, dtype=np.float32)), rel=1e-3)
    h64 = oracle.dem_ingest(ramp(np.float64, (3, 3)), 1.0)
    assert h64.dtype == np.float32 and np.array_equal(h64, ramp(np.float64, (3, 3)).astype(np.float32))


def test_normalize_known_answers(oracle):
    h = oracle.dem_ingest(ramp(np.float64, (3, 3)), 1.0)                 # test_height_upload.py:35-49
    n = oracle.dem_normalize(h, "minmax", eps=1e-8, out_range=(10.0, 20.0))
    mn, mx, _, _ = oracle.dem_stats(n)
    assert mn == pytest.approx(10.0, rel=1e-5) and mx == pytest.approx(20.0, rel=1e-5)
    z = oracle.dem_normalize(n, "zscore", eps=1e-6)
    _, _, mean, std = oracle.dem_stats(z)
    assert abs(mean) < 1e-5 and std == pytest.approx(1.0, rel=1e-3)
    flat = oracle.dem_normalize(np.ones((4, 4), np.float32), "minmax")  # denom = max(|0|, eps)
    assert (flat == 0).all()


def test_percentile_range(oracle):
    h = np.arange(1000, dtype=np.float32)[::-1].reshape(25, 40).copy()
    assert oracle.dem_percentile_range(h) == (10.0, 990.0)               # buf[(len*0.01)], buf[(len*0.99)] after sort
    big = np.arange(300 * 1000, dtype=np.float32).reshape(300, 1000)     # > 65536: stride sample, step = n // 65536 = 4
    p1, p99 = oracle.dem_percentile_range(big)
    samp = np.sort(big.reshape(-1)[::4])
    assert p1 == samp[int(np.float32(samp.size) * np.float32(0.01))] and p99 == samp[int(np.float32(samp.size) * np.float32(0.99))]
