"""Host-side logic of the drop-in module on a machine WITHOUT a GPU: camera functions, validation and error
classes/strings, argument guards, PNG encoder, DEM helpers -- expectations re-expressed from the reference's
tests (tests/test_camera.py, test_api_validation.py, test_grid_generate.py, test_colormap.py, test_dem_*.py)."""
import io
import re

import numpy as np
import pytest

import vulkan_forge as vf
import vulkan_forge._vulkan_forge as ext


def test_public_exports_exist():
    for name in ("Renderer", "TerrainSpike", "Scene", "render_triangle_rgba", "render_triangle_png", "make_terrain",
                 "grid_generate", "generate_grid", "colormap_supported", "camera_look_at", "camera_perspective",
                 "camera_view_proj", "dem_stats", "dem_normalize", "enumerate_adapters", "device_probe", "__version__"):
        assert hasattr(vf, name), name
    for name in ("Renderer", "TerrainSpike", "Scene", "enumerate_adapters", "device_probe", "grid_generate",
                 "colormap_supported", "camera_look_at", "camera_perspective", "camera_view_proj"):
        assert hasattr(ext, name), name            # src/lib.rs:961-976
    from vshade import Renderer as R2
    assert R2 is vf.Renderer                       # tests/test_api.py:12-15


def test_camera_functions_match_oracle_exactly(oracle):
    eye, tgt, up = (1.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)
    a = vf.camera_look_at(eye, tgt, up)
    assert a.shape == (4, 4) and a.dtype == np.float32 and a.flags.c_contiguous
    assert np.array_equal(a, oracle.camera_look_at(eye, tgt, up))
    for clip in ("wgpu", "gl"):
        assert np.array_equal(vf.camera_perspective(60.0, 16 / 9, 0.1, 100.0, clip), oracle.camera_perspective(60.0, 16 / 9, 0.1, 100.0, clip))
        assert np.array_equal(vf.camera_view_proj(eye, tgt, up, 60.0, 1.5, 0.1, 100.0, clip),
                              oracle.camera_view_proj(eye, tgt, up, 60.0, 1.5, 0.1, 100.0, clip))
    assert abs(vf.camera_look_at((0, 0, 3), (0, 0, 0), (0, 1, 0))[2, 3] + 3.0) < 1e-6
    assert np.array_equal(vf.camera_perspective(45.0, 1.0, 0.1, 100.0), vf.camera_perspective(45.0, 1.0, 0.1, 100.0, clip_space="wgpu"))


@pytest.mark.parametrize("call,msg", [
    (lambda: vf.camera_look_at((float("inf"), 0, 0), (0, 0, 0), (0, 1, 0)), "eye/target/up components must be finite"),
    (lambda: vf.camera_look_at((0, 0, 3), (0, 0, 0), (0, 0, -1)), "up vector must not be colinear with view direction"),
    (lambda: vf.camera_perspective(0.0, 1.0, 0.1, 100.0), "fovy_deg must be finite and in (0, 180)"),
    (lambda: vf.camera_perspective(45.0, 0.0, 0.1, 100.0), "aspect must be finite and > 0"),
    (lambda: vf.camera_perspective(45.0, 1.0, 0.0, 100.0), "znear must be finite and > 0"),
    (lambda: vf.camera_perspective(45.0, 1.0, 0.1, 0.05), "zfar must be finite and > znear"),
    (lambda: vf.camera_perspective(45.0, 1.0, 0.1, 100.0, "invalid"), "clip_space must be 'wgpu' or 'gl'"),
    (lambda: vf.camera_view_proj((0, 0, 3), (0, 0, 0), (0, 1, 0), 0.0, 1.0, 0.1, 100.0), "fovy_deg must be finite and in (0, 180)"),
])
def test_camera_errors_are_runtime_errors_with_exact_text(call, msg):
    with pytest.raises(RuntimeError, match=re.escape(msg)):
        call()


@pytest.mark.parametrize("args,msg", [
    ((1, 3), "nx and nz must be >= 2"), ((3, 1), "nx and nz must be >= 2"),
    ((3, 3, (0.0, 1.0)), "spacing components must be finite and > 0"),
    ((3, 3, (float("inf"), 1.0)), "spacing components must be finite and > 0"),
    ((3, 3, (1.0, 1.0), "corner"), "origin must be 'center'"),
])
def test_grid_generate_validation_is_value_error(args, msg):       # checked before any device work
    with pytest.raises(ValueError, match=re.escape(msg)):
        vf.grid_generate(*args)


def test_colormap_registry():
    assert vf.colormap_supported() == ["viridis", "magma", "terrain"]
    for bad in ("VIRIDIS", "invalid_colormap"):
        with pytest.raises(RuntimeError, match=re.escape(f"Unknown colormap '{bad}'. Supported: viridis, magma, terrain")):
            vf.TerrainSpike(64, 64, grid=8, colormap=bad)
        with pytest.raises(RuntimeError, match="Unknown colormap"):
            vf.Scene(64, 64, grid=8, colormap=bad)


def test_argument_guards(tmp_path):
    from vulkan_forge import _validate as V
    assert V.size_wh(32, 24) == (32, 24) and V.grid(128) == 128
    for bad in ((0, 10), (10, -1), (8193, 10)):
        with pytest.raises(ValueError):
            V.size_wh(*bad)
    with pytest.raises(ValueError, match="must be an integer"):
        V.size_wh("x", 3)
    for bad in (1, 4097):
        with pytest.raises(ValueError):
            V.grid(bad)
    assert V.png_path(tmp_path / "a.PNG").endswith("a.PNG")
    with pytest.raises(ValueError, match="must end with .png"):
        V.png_path(tmp_path / "a.jpg")
    with pytest.raises(ValueError, match="directory does not exist"):
        V.png_path(tmp_path / "nope" / "a.png")
    with pytest.raises(ValueError):
        vf.render_triangle_png(tmp_path / "x.png", 0, 10)           # tests/test_api_validation.py:17-21
    with pytest.raises(ValueError):
        vf.render_triangle_png(tmp_path / "x.jpg", 10, 10)
    with pytest.raises(ValueError):
        vf.make_terrain(64, 64, 1)                                  # :31-34


def test_dem_helpers():
    a = np.arange(12, dtype=np.float32).reshape(3, 4)
    mn, mx, mean, std = vf.dem_stats(a)
    assert (mn, mx) == (0.0, 11.0) and abs(mean - 5.5) < 1e-6 and abs(std - a.std()) < 1e-5
    out, st = vf.dem_normalize(a, mode="minmax", out_range=(0.0, 2.0), return_stats=True)
    assert out.min() == 0.0 and abs(out.max() - 2.0) < 1e-6 and st[0] == 0.0
    z = vf.dem_normalize(a, mode="zscore")
    assert abs(z.mean()) < 1e-6
    assert (vf.dem_normalize(np.ones((2, 2), np.float32)) == 0).all()
    with pytest.raises(ValueError, match="mode must be"):
        vf.dem_normalize(a, mode="x")
    with pytest.raises(RuntimeError, match="heightmap must be 2-D"):
        vf.dem_stats(np.zeros(4, np.float32))
    with pytest.raises(RuntimeError):
        vf.dem_stats(np.zeros((4, 4), np.int32))


def test_png_encoder_round_trip():
    from PIL import Image
    rng = np.random.default_rng(1)
    for shape in ((1, 1), (7, 5), (48, 64), (33, 257)):
        img = rng.integers(0, 256, size=(*shape, 4), dtype=np.uint8)
        img[: shape[0] // 2] = [39, 39, 48, 255]
        png = ext._encode_png_rgba8(img)
        assert png[:8] == b"\x89PNG\r\n\x1a\n"
        back = np.asarray(Image.open(io.BytesIO(png)).convert("RGBA"))
        assert np.array_equal(back, img)


def _png_scanlines(png: bytes) -> bytes:
    """concatenate the IDAT payloads and inflate them: the filtered scanlines the encoder produced"""
    import struct
    import zlib
    at, idat, kinds = 8, b"", []
    while at < len(png):
        n, kind = struct.unpack(">I4s", png[at:at + 8])
        body = png[at + 8:at + 8 + n]
        assert zlib.crc32(kind + body) == struct.unpack(">I", png[at + 8 + n:at + 12 + n])[0]
        kinds.append(kind)
        if kind == b"IDAT":
            idat += body
        at += 12 + n
    assert kinds[0] == b"IHDR" and kinds[-1] == b"IEND" and set(kinds[1:-1]) == {b"IDAT"}
    return zlib.decompress(idat), kinds.count(b"IDAT")


def test_png_encoder_is_row_parallel_and_deterministic(monkeypatch):
    """rows are deflated in independent runs (one IDAT each, Adler-32 combined); the bytes do not depend
    on the number of worker threads, every chunk CRC is right and the stream inflates to the adaptive-filtered rows."""
    from PIL import Image
    H, W = 700, 900                                            # 3601-byte rows -> 291 rows per ~1 MiB run -> 3 IDAT chunks
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([(xx * 255 // W), (yy * 255 // H), ((xx * 3 + yy) & 255), np.full_like(xx, 255)], axis=2).astype(np.uint8)
    img[::5, ::7, :3] ^= 0x5A
    out = {}
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("VF_PNG_THREADS", threads)
        out[threads] = bytes(ext._encode_png_rgba8(img))
    assert out["1"] == out["3"] == out["8"]
    raw, nidat = _png_scanlines(out["1"])
    assert nidat == 3 and len(raw) == H * (4 * W + 1)
    rows = np.frombuffer(raw, np.uint8).reshape(H, 4 * W + 1)
    assert set(np.unique(rows[:, 0])) <= {0, 1, 2, 3, 4}
    sub = rows[rows[:, 0] == 1]                                # Sub-filtered rows undo with a running sum per channel
    if len(sub):
        y = int(np.flatnonzero(rows[:, 0] == 1)[0])
        px = np.cumsum(rows[y, 1:].reshape(W, 4).astype(np.uint32), axis=0).astype(np.uint8)
        assert np.array_equal(px, img[y])
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(out["1"])).convert("RGBA")), img)


def test_no_gpu_means_loud_failure():
    """The product has no CPU fallback: without a HIP device anything that renders raises the reference's
    'No suitable GPU adapter' (src/terrain/mod.rs:285) instead of silently computing on the host."""
    if ext.enumerate_adapters():
        pytest.skip("a HIP device is present")
    assert ext.device_probe()["status"] == "unsupported" and ext.device_probe()["message"] == "No suitable GPU adapter"
    for make in (lambda: vf.TerrainSpike(64, 64), lambda: vf.Scene(64, 64), lambda: vf.Renderer(16, 16),
                 lambda: vf.grid_generate(3, 3), lambda: vf.render_triangle_rgba(16, 16)):
        with pytest.raises(RuntimeError, match="No suitable GPU adapter"):
            make()
