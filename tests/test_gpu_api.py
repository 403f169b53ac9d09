"""The reference's own integration tests, re-expressed against the drop-in module (needs the GPU):
tests/test_t31_integration.py, test_t41_scene.py, test_colormap.py, test_determinism.py, test_api.py,
test_api_validation.py, test_camera.py::TestTerrainSpikeIntegration."""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import vulkan_forge as vfpkg                      # noqa: E402
import vulkan_forge._vulkan_forge as vf           # noqa: E402


def test_t31_uniform_lanes_layout():
    spike = vf.TerrainSpike(256, 192, grid=64, colormap="viridis")
    u = spike.debug_uniforms_f32()
    assert isinstance(u, np.ndarray) and u.dtype == np.float32 and u.shape == (44,)
    assert abs(u[36] - 1) < 1e-6 and abs(u[37] - 1) < 1e-6 and abs(u[38] - 1) < 1e-6 and abs(u[39]) < 1e-6


def test_t31_render_png_smoke(tmp_path):
    out = tmp_path / "terrain_smoke.png"
    vf.TerrainSpike(320, 240, grid=64, colormap="viridis").render_png(str(out))
    assert out.exists() and out.stat().st_size > 4096


def test_t41_scene_renders_png_and_height_upload_changes_output(tmp_path):
    from PIL import Image
    o1, o2 = tmp_path / "s1.png", tmp_path / "s2.png"
    scn = vf.Scene(320, 240, grid=64, colormap="viridis")
    scn.render_png(str(o1))
    assert o1.stat().st_size > 4096
    h = (np.sin(np.linspace(0, 4 * np.pi, 128))[:, None] * np.cos(np.linspace(0, 4 * np.pi, 128))[None, :]).astype("float32") * 0.25
    scn.set_height_from_r32f(h)
    scn.render_png(str(o2))
    assert o1.stat().st_size != o2.stat().st_size
    a, b = np.asarray(Image.open(o1)), np.asarray(Image.open(o2))
    assert a.shape == (240, 320, 4) and not np.array_equal(a, b)


def test_png_holds_exactly_the_rendered_pixels(tmp_path, oracle, luts):
    from PIL import Image
    W, H, G = 200, 120, 32
    out = tmp_path / "p.png"
    s = vf.TerrainSpike(W, H, grid=G, colormap="magma")
    s.render_png(str(out))
    png = np.asarray(Image.open(out).convert("RGBA"))
    ref, _ = oracle.render_terrain(oracle.default_uniforms(0, W, H), W, H, G, oracle.SPIKE_DUMMY_HEIGHT, luts["magma"])
    assert np.array_equal(png, s.render_rgba())
    assert np.abs(png.astype(int) - ref.astype(int)).max() <= 1


def test_png_scanlines_are_filtered_on_the_gpu_like_the_host_encoder(tmp_path, luts):
    """vf_terrain_read_png_scanlines = the host encoder's adaptive filter, row for row and byte for byte
    (None/Sub/Up/Average/Paeth, smallest sum of absolute residuals, first on ties), so render_png only deflates."""
    import zlib
    from PIL import Image
    from test_host_api import _png_scanlines
    from vulkan_forge_amd import cabi
    import oracle as O
    for W, H, G in ((200, 120, 32), (1, 1, 2), (257, 65, 16), (1920, 1080, 256)):
        t = cabi.Terrain(W, H, G, luts["terrain"])
        try:
            t.set_uniforms(O.default_uniforms(1, W, H))
            t.set_height(np.random.default_rng(W).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25))
            t.render()
            rgba, scan = t.read_rgba(), t.read_png_scanlines()
            assert scan.shape == (H, 4 * W + 1)
            host, _ = _png_scanlines(bytes(vf._encode_png_rgba8(rgba)))
            assert scan.tobytes() == host
            if H > 64:                                         # rank 0 of 2 then holds only part of the frame
                t.set_shard(0, 2, 64)
                t.render()
                with pytest.raises(cabi.VfError, match="whole frame"):
                    t.read_png_scanlines()
        finally:
            t.close()
    out = tmp_path / "big.png"
    s = vf.Scene(1920, 1080, grid=512)
    s.set_height_from_r32f(np.random.default_rng(3).random((512, 512), dtype=np.float32) * np.float32(0.5) - np.float32(0.25))
    s.render_png(str(out))
    assert np.array_equal(np.asarray(Image.open(out).convert("RGBA")), s.render_rgba())


def test_height_argument_errors():
    scn = vf.Scene(64, 64, grid=8)
    with pytest.raises(RuntimeError, match="height must be C-contiguous"):
        scn.set_height_from_r32f(np.zeros((8, 16), np.float32)[:, ::2])
    with pytest.raises(TypeError):
        scn.set_height_from_r32f(np.zeros((8, 8), np.float64))
    with pytest.raises(TypeError):
        scn.set_height_from_r32f(np.zeros(8, np.float32))
    with pytest.raises(TypeError):
        scn.set_height_from_r32f([[0.0, 1.0]])
    assert not hasattr(vf.TerrainSpike(64, 64, grid=8), "set_height_from_r32f")   # TerrainSpike has no height setter


def test_camera_integration():
    spike = vf.TerrainSpike(512, 512)
    u0 = spike.debug_uniforms_f32()
    assert len(u0) == 44
    spike.set_camera_look_at((1.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 60.0, 0.1, 100.0)
    assert not np.allclose(u0, spike.debug_uniforms_f32())
    with pytest.raises(RuntimeError, match=r"fovy_deg must be finite and in \(0, 180\)"):
        spike.set_camera_look_at((0, 0, 3), (0, 0, 0), (0, 1, 0), 0.0, 0.1, 100.0)
    with pytest.raises(RuntimeError, match="up vector must not be colinear"):
        spike.set_camera_look_at((0, 0, 3), (0, 0, 0), (0, 0, -1), 45.0, 0.1, 100.0)
    spike.set_camera_look_at((0, 0, 3), (0, 0, 0), (0, 1, 0), 45.0, 0.1, 100.0)
    u = spike.debug_uniforms_f32()
    np.testing.assert_allclose(u[:16].reshape(4, 4, order="F"), vf.camera_look_at((0, 0, 3), (0, 0, 0), (0, 1, 0)), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(u[16:32].reshape(4, 4, order="F"), vf.camera_perspective(45.0, 1.0, 0.1, 100.0, "wgpu"), rtol=1e-5, atol=1e-6)
    t = vf.TerrainSpike(128, 96, grid=32)
    proj = t.debug_uniforms_f32()[16:32].reshape(4, 4, order="F")
    assert np.allclose(proj, vf.camera_perspective(45.0, 128 / 96, 0.1, 100.0, clip_space="wgpu"), atol=1e-6)


def test_colormaps_render_and_format_selection(tmp_path, monkeypatch):
    for cm in ("viridis", "magma", "terrain"):
        out = tmp_path / f"{cm}.png"
        vf.TerrainSpike(128, 128, grid=32, colormap=cm).render_png(str(out))
        assert out.stat().st_size > 1000
    assert vf.TerrainSpike(128, 128, grid=32).debug_lut_format() in ("Rgba8UnormSrgb", "Rgba8Unorm")
    monkeypatch.setenv("VF_FORCE_LUT_UNORM", "1")
    t = vf.TerrainSpike(128, 128, grid=32, colormap="viridis")
    assert t.debug_lut_format() == "Rgba8Unorm"
    out = tmp_path / "unorm.png"
    t.render_png(str(out))
    assert out.stat().st_size > 1000


def test_unorm_fallback_matches_oracle(monkeypatch, oracle, luts):
    monkeypatch.setenv("VF_FORCE_LUT_UNORM", "1")
    W, H, G = 160, 120, 32
    t = vf.Scene(W, H, grid=G, colormap="terrain")
    ref, _ = oracle.render_terrain(oracle.default_uniforms(1, W, H), W, H, G, oracle.SCENE_DUMMY_HEIGHT,
                                   oracle.lut_to_linear_u8(luts["terrain"]), lut_is_srgb=False)
    assert np.abs(t.render_rgba().astype(int) - ref.astype(int)).max() <= 1


def test_triangle_determinism_and_api(tmp_path):
    shas = set()
    for _ in range(3):
        a = vfpkg.Renderer(64, 64).render_triangle_rgba()
        assert a.shape == (64, 64, 4) and a.dtype == np.uint8
        shas.add(hashlib.sha256(a.tobytes()).hexdigest())
    assert len(shas) == 1
    assert vfpkg.Renderer(16, 16).info() == "Renderer 16x16, format=Rgba8UnormSrgb"
    a = vfpkg.render_triangle_rgba(32, 24)
    assert a.shape == (24, 32, 4)
    out = tmp_path / "tri.png"
    vfpkg.render_triangle_png(str(out), 32, 24)
    assert out.exists() and out.stat().st_size > 0
    t = vfpkg.make_terrain(64, 48, 16)
    o2 = tmp_path / "t.png"
    t.render_png(str(o2))
    assert o2.exists() and o2.stat().st_size > 0


def test_diagnostics():
    ads = vf.enumerate_adapters()
    assert ads and ads[0]["backend"] == "HIP" and "gfx" in ads[0]["features"]
    p = vf.device_probe()
    assert p["status"] == "ok" and p["millis"] >= 0 and p["backend_request"] == "AUTO"
    assert vf.device_probe("vulkan")["status"] == "unsupported"


# ---- CLI tools -----------------------------------------------------------------------------------
def test_cli_tools_on_the_gpu(tmp_path):
    import json
    from PIL import Image
    from vulkan_forge_amd.tools import determinism_harness, device_diagnostics, perf_sanity, terrain_spike
    assert terrain_spike.main(["--width", "200", "--height", "120", "--grid", "48", "--out", str(tmp_path / "t.png")]) == 0
    assert Image.open(tmp_path / "t.png").size == (200, 120)
    for workload in ("triangle", "terrain", "scene"):
        rep_path = tmp_path / f"perf_{workload}.json"
        assert perf_sanity.main(["--width", "160", "--height", "96", "--runs", "5", "--warmups", "1", "--workload", workload,
                                 "--grid", "32", "--json", str(rep_path)]) == 0
        rep = json.loads(rep_path.read_text())
        assert rep["runs"] == 5 and len(rep["steady"]["samples_ms"]) == 5 and rep["init_ms"] > 0
        assert rep["steady"]["min_ms"] <= rep["steady"]["median_ms"] <= rep["steady"]["p95_ms"] <= rep["steady"]["max_ms"]
    for workload, procs in (("triangle", 0), ("terrain", 0), ("terrain", 2)):
        out = tmp_path / f"det_{workload}_{procs}"
        assert determinism_harness.main(["--width", "96", "--height", "64", "--runs", "3", "--processes", str(procs), "--png",
                                         "--workload", workload, "--grid", "24", "--out-dir", str(out)]) == 0
        rep = json.loads((out / "determinism_report.json").read_text())
        assert rep["all_equal"] and len(rep["hashes"]) == 3 and len(rep["unique"]) == 1 and "png" in rep
    assert device_diagnostics.main(["--json", str(tmp_path / "diag.json")]) == 0
    diag = json.loads((tmp_path / "diag.json").read_text())
    assert diag["backends"]["hip"]["adapters"] and diag["errors"] == []


# ---- C-ABI contract: status codes + vf_last_error, never a crash -----------------------------------------------------
def test_cabi_misuse_reports_errors(luts):
    import ctypes as C
    from vulkan_forge_amd import cabi
    lib = cabi.load()
    ctx = C.c_void_p()
    assert lib.vf_ctx_create(0, C.byref(ctx)) == cabi.VF_OK
    t = C.c_void_p()
    lut = np.ascontiguousarray(luts["viridis"], np.uint8).reshape(1024)
    for w, h, g in ((0, 10, 8), (10, 0, 8), (20000, 10, 8), (10, 10, 9000)):      # (grid < 2 is raised to 2 like src/terrain/mod.rs:260)
        assert lib.vf_terrain_create(ctx, w, h, g, lut.ctypes.data, 1, C.byref(t)) == cabi.VF_ERR_INVALID, (w, h, g)
        assert lib.vf_last_error()
    assert lib.vf_terrain_create(ctx, 64, 48, 8, None, 1, C.byref(t)) == cabi.VF_ERR_INVALID
    assert lib.vf_terrain_create(None, 64, 48, 8, lut.ctypes.data, 1, C.byref(t)) == cabi.VF_ERR_INVALID
    assert lib.vf_terrain_create(ctx, 64, 48, 8, lut.ctypes.data, 1, C.byref(t)) == cabi.VF_OK
    buf = np.zeros((48, 64, 4), np.uint8)
    assert lib.vf_terrain_render(t, None) == cabi.VF_ERR_INVALID                      # uniforms not set
    assert b"uniforms" in lib.vf_last_error()
    assert lib.vf_terrain_read_rgba(t, buf.ctypes.data, 0, 48) == cabi.VF_ERR_INVALID  # nothing rendered
    u = np.zeros(44, np.float32); u[0] = u[5] = u[10] = u[15] = 1.0; u[16] = u[21] = u[26] = u[31] = 1.0
    assert lib.vf_terrain_set_uniforms(t, u.ctypes.data) == cabi.VF_OK
    assert lib.vf_terrain_set_uniforms(t, None) == cabi.VF_ERR_INVALID
    assert lib.vf_terrain_set_height(t, None, 4, 4) == cabi.VF_ERR_INVALID
    assert lib.vf_terrain_set_height(t, buf.ctypes.data, 0, 4) == cabi.VF_ERR_INVALID
    assert lib.vf_terrain_set_shard(t, 2, 2, 64) == cabi.VF_ERR_INVALID               # rank >= nranks
    assert lib.vf_terrain_set_shard(t, 0, 2, 48) == cabi.VF_ERR_INVALID               # band_h not a power of two
    assert lib.vf_terrain_set_shard(t, 0, 2, 32) == cabi.VF_ERR_INVALID               # band_h below the tile height
    assert lib.vf_terrain_set_tile_shard(t, 3, 3, 1) == cabi.VF_ERR_INVALID
    assert lib.vf_terrain_set_shade_mode(t, 7) == cabi.VF_ERR_INVALID
    assert lib.vf_terrain_render(t, None) == cabi.VF_OK                               # identity matrices: still a valid frame
    assert lib.vf_terrain_read_rgba(t, buf.ctypes.data, 40, 9) == cabi.VF_ERR_INVALID  # rows outside the frame
    assert lib.vf_terrain_read_rgba(t, buf.ctypes.data, 0, 48) == cabi.VF_OK
    n = C.c_uint32()
    assert lib.vf_tile_layout(0, 10, 0, 1, 1, None, 0, C.byref(n)) == cabi.VF_ERR_INVALID
    assert lib.vf_tile_layout(100, 100, 0, 1, 1, None, 0, None) == cabi.VF_ERR_INVALID
    stats = np.zeros(64, np.uint32)
    assert lib.vf_terrain_debug_item_stats(t, stats.ctypes.data, 16, C.byref(n)) == cabi.VF_ERR_INVALID   # timing not enabled
    assert lib.vf_terrain_debug_phase_cycles(t, stats.ctypes.data, 8) == cabi.VF_ERR_INVALID              # not a profiling build
    lib.vf_terrain_destroy(t); lib.vf_terrain_destroy(None)
    lib.vf_ctx_destroy(ctx); lib.vf_ctx_destroy(None)
    assert lib.vf_ctx_create(99, C.byref(ctx)) != cabi.VF_OK                          # no such device


@pytest.mark.parametrize("threads", ["1", "3"])
def test_large_readback_goes_through_the_pinned_ring_unchanged(luts, monkeypatch, threads):
    """Frames of 8 MiB and more are read back through a ring of pinned chunks by a few host threads: every byte must arrive,
    also when the last chunk is short, when there are more chunks than ring slots, and for a row range."""
    from vulkan_forge_amd import cabi
    monkeypatch.setenv("VF_COPY_THREADS", threads)
    W, H, G = 3000, 3301, 96                                      # 39.6 MB: five chunks, the last one short
    t = cabi.Terrain(W, H, G, luts["viridis"])
    rng = np.random.default_rng(5)
    t.set_height((rng.random((G, G), dtype=np.float32) - np.float32(0.5)) * np.float32(0.4))
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    t.set_uniforms(b.camera_uniforms("fill", W, H))
    t.render(); t.sync()
    want = np.empty((H, W, 4), np.uint8)                          # reference: pieces below 8 MiB take the plain copy
    for y0 in range(0, H, 600):
        rows = min(600, H - y0)
        t._check(t.lib.vf_terrain_read_rgba(t.t, want[y0:].ctypes.data, y0, rows))
    assert len(np.unique(want.reshape(-1, 4), axis=0)) > 100      # a real picture, not a cleared frame
    got = np.empty((H, W, 4), np.uint8)
    t._check(t.lib.vf_terrain_read_rgba(t.t, got.ctypes.data, 0, H))
    assert np.array_equal(got, want)
    part = np.full((1500, W, 4), 7, np.uint8)
    t._check(t.lib.vf_terrain_read_rgba(t.t, part.ctypes.data, 1001, 1500))
    assert np.array_equal(part, want[1001:2501])


def test_readback_into_page_locked_memory_by_stores_and_by_the_copy_engine(luts):
    """vf_host_alloc destinations are written by the device's own stores (k_copy_to_host: 16 bytes per lane + a byte tail) when source
    and destination are 16-byte aligned, by the copy engine otherwise; hipHostMalloc-sized requests (< 4 MiB) come from the runtime,
    larger ones from huge pages registered with it.  Every byte must arrive either way, and nothing beyond the frame is touched."""
    import ctypes as C
    from vulkan_forge_amd import cabi
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    for W, H in ((1001, 1101), (257, 63)):                        # 4 408 404 B (huge pages; a 4-byte tail) / 64 764 B (hipHostMalloc; 12-byte tail)
        t = cabi.Terrain(W, H, 64, luts["magma"])
        try:
            t.set_height((np.random.default_rng(W).random((64, 64), dtype=np.float32) - np.float32(0.5)) * np.float32(0.4))
            t.set_uniforms(b.camera_uniforms("fill", W, H))
            t.render(); t.sync()
            want = t.read_rgba()                                   # ordinary memory: the plain / staged copy
            assert len(np.unique(want.reshape(-1, 4), axis=0)) > 50
            n = W * H * 4
            p = C.c_void_p()
            t._check(t.lib.vf_host_alloc(C.c_size_t(n + 64), C.byref(p)))
            try:
                buf = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n + 64,))
                buf[:] = 0xAB
                t._check(t.lib.vf_terrain_read_rgba(t.t, p, 0, H))                      # aligned both ends: stores
                assert np.array_equal(buf[:n].reshape(H, W, 4), want) and (buf[n:] == 0xAB).all()
                buf[:] = 0xCD
                t._check(t.lib.vf_terrain_read_rgba(t.t, C.c_void_p(p.value + 4), 3, H - 5))   # W odd, y0 = 3: neither end aligned -> copy engine
                m = W * (H - 5) * 4
                assert np.array_equal(buf[4:4 + m].reshape(H - 5, W, 4), want[3:H - 2]) and (buf[:4] == 0xCD).all() and (buf[4 + m:] == 0xCD).all()
            finally:
                t.lib.vf_host_free(p)
        finally:
            t.close()


def test_frame_sized_render_rgba_arrays_live_in_the_pinned_pool(oracle, luts, monkeypatch):
    """Round 5: frame-sized render_rgba results are NumPy arrays over page-locked buffers of a pool (one DMA, no host copy); every call
    returns its own array, a dead array's buffer is reused, and the pageable path (VF_RGBA_PAGEABLE) gives the same bytes."""
    import gc
    W, H, G = 1920, 1080, 256
    s = vf.TerrainSpike(W, H, grid=G, colormap="viridis")
    first = s.render_rgba()                                      # (the first frame too: huge pages + registration cost less than one copy's page faults)
    assert not first.flags["OWNDATA"]
    a = s.render_rgba()
    b = s.render_rgba()
    assert np.array_equal(first, a)
    assert a.shape == (H, W, 4) and a.flags["C_CONTIGUOUS"] and a.flags["WRITEABLE"] and not a.flags["OWNDATA"]
    assert a.ctypes.data != b.ctypes.data and np.array_equal(a, b)
    ref, _ = oracle.render_terrain(oracle.default_uniforms(0, W, H), W, H, G, oracle.SPIKE_DUMMY_HEIGHT, luts["viridis"], nthreads=min(16, oracle.max_threads()), want_vis=False)
    assert np.abs(a.astype(int) - ref.astype(int)).max() <= 1
    keep = a.copy()
    where = a.ctypes.data
    a[:] = 0                                                     # the caller owns what it got
    assert np.array_equal(b, keep)
    del a
    gc.collect()
    c = s.render_rgba()                                          # the dead array's buffer comes back from the pool
    assert c.ctypes.data == where and np.array_equal(c, keep)
    monkeypatch.setenv("VF_RGBA_PAGEABLE", "1")
    d = s.render_rgba()
    assert d.flags["OWNDATA"] and np.array_equal(d, keep)
