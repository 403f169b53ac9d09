"""Pin the CPU oracle against every known-answer fact the reference's own tests hold for this path
(SURVEY.md 8(c)): grid_generate, camera matrices, uniform block, clear colour, LUT bytes, error strings."""
import hashlib
import re

import numpy as np
import pytest

RTOL, ATOL = 1e-5, 1e-6   # reference tests/test_camera.py:36-37


# ---- grid_generate: reference tests/test_grid_generate.py + src/terrain/mesh.rs:92-129 ----------------
def test_grid_basic_shapes_dtypes(oracle):
    xy, uv, idx = oracle.grid_generate(4, 3, (2.0, 1.0))          # test_grid_generate.py:9-20
    assert xy.shape == (12, 2) and uv.shape == (12, 2) and idx.shape == (36,)
    assert xy.dtype == np.float32 and uv.dtype == np.float32 and idx.dtype == np.uint32


def test_grid_uv_corners(oracle):
    _, uv, _ = oracle.grid_generate(4, 3, (2.0, 1.0))             # :23-41, mesh.rs:96-106 (exact equality)
    assert uv[0].tolist() == [0.0, 0.0] and uv[3].tolist() == [1.0, 0.0]
    assert uv[8].tolist() == [0.0, 1.0] and uv[11].tolist() == [1.0, 1.0]


def test_grid_first_triangle_ccw(oracle):
    xy, _, idx = oracle.grid_generate(3, 3, (1.0, 1.0))           # :44-62, mesh.rs:108-121
    p0, p1, p2 = xy[idx[0]], xy[idx[1]], xy[idx[2]]
    e1, e2 = p1 - p0, p2 - p0
    assert e1[0] * e2[1] - e1[1] * e2[0] > 0


def test_grid_256_u32_count(oracle):
    xy, uv, idx = oracle.grid_generate(256, 256)                  # :65-79
    assert idx.dtype == np.uint32 and idx.shape == (390150,) and xy.shape == (65536, 2) and uv.shape == (65536, 2)


def test_grid_index_width_switch(oracle):
    assert not oracle.grid_uses_u16(256, 256)                     # mesh.rs:123-129: 65536 vertices -> u32
    assert oracle.grid_uses_u16(255, 255)                         # 65025 -> u16


def test_grid_index_pattern(oracle):
    _, _, idx = oracle.grid_generate(4, 3)                        # mesh.rs:64-73: [i0,i1,i2, i2,i1,i3]
    assert idx[:6].tolist() == [0, 1, 4, 4, 1, 5]
    assert idx[-6:].tolist() == [6, 7, 10, 10, 7, 11]


@pytest.mark.parametrize("args,msg", [
    ((1, 3), "nx and nz must be >= 2"), ((3, 1), "nx and nz must be >= 2"),
    ((3, 3, (0.0, 1.0)), "spacing components must be finite and > 0"),
    ((3, 3, (1.0, -1.0)), "spacing components must be finite and > 0"),
    ((3, 3, (float("inf"), 1.0)), "spacing components must be finite and > 0"),
    ((3, 3, (1.0, 1.0), "corner"), "origin must be 'center'"),
])
def test_grid_validation_strings(oracle, args, msg):               # :82-104
    with pytest.raises(ValueError, match=re.escape(msg)):
        oracle.grid_generate(*args)


def test_grid_centered(oracle):
    xy, _, _ = oracle.grid_generate(3, 3, (2.0, 2.0))             # :107-121
    exp = [[-2, -2], [0, -2], [2, -2], [-2, 0], [0, 0], [2, 0], [-2, 2], [0, 2], [2, 2]]
    assert np.array_equal(xy, np.array(exp, np.float32))


# ---- camera: reference tests/test_camera.py ----------------------------------------------------------
def test_look_at_known_answer(oracle):
    v = oracle.camera_look_at((0, 0, 3), (0, 0, 0), (0, 1, 0))    # :56-68
    assert v.shape == (4, 4) and v.dtype == np.float32 and v.flags.c_contiguous
    assert abs(v[2, 3] - (-3.0)) < ATOL


def test_perspective_discriminates_the_transposed_gl_to_wgpu(oracle):
    """SURVEY.md finding 2 / 8(c): src/camera.rs:14-21 feeds a row-major-looking literal to a column-major
    constructor.  The reference's rows are [f,0,0,0],[0,f,0,0],[0,0,-.501001,-.1001001],[0,0,-1.501001,-.1001001]."""
    p = oracle.camera_perspective(45.0, 1.0, 0.1, 100.0, "wgpu")
    f = 1.0 / np.tan(np.radians(45.0) / 2)
    exp = np.array([[f, 0, 0, 0], [0, f, 0, 0], [0, 0, -0.501001, -0.1001001], [0, 0, -1.501001, -0.1001001]], np.float32)
    np.testing.assert_allclose(p, exp, rtol=RTOL, atol=ATOL)
    gl = oracle.camera_perspective(45.0, 1.0, 0.1, 100.0, "gl")   # :111-124: x,y rows identical, z rows differ
    np.testing.assert_allclose(gl[:2], p[:2], rtol=RTOL, atol=ATOL)
    assert not np.allclose(gl, p)
    assert np.array_equal(oracle.camera_perspective(45.0, 1.0, 0.1, 100.0), p)   # default clip space :103-108


def test_view_proj_is_proj_times_view(oracle):
    eye, tgt, up = (0, 0, 3), (0, 0, 0), (0, 1, 0)                # :185-201
    vp = oracle.camera_view_proj(eye, tgt, up, 45.0, 16 / 9, 0.1, 100.0, "wgpu")
    exp = oracle.camera_perspective(45.0, 16 / 9, 0.1, 100.0, "wgpu") @ oracle.camera_look_at(eye, tgt, up)
    np.testing.assert_allclose(vp, exp, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("fn,args,msg", [
    ("camera_look_at", ((float("inf"), 0, 0), (0, 0, 0), (0, 1, 0)), "eye/target/up components must be finite"),
    ("camera_look_at", ((0, 0, 3), (float("nan"), 0, 0), (0, 1, 0)), "eye/target/up components must be finite"),
    ("camera_look_at", ((0, 0, 3), (0, 0, 0), (0, float("inf"), 0)), "eye/target/up components must be finite"),
    ("camera_look_at", ((0, 0, 3), (0, 0, 0), (0, 0, -1)), "up vector must not be colinear with view direction"),
    ("camera_perspective", (0.0, 1.0, 0.1, 100.0), "fovy_deg must be finite and in (0, 180)"),
    ("camera_perspective", (180.0, 1.0, 0.1, 100.0), "fovy_deg must be finite and in (0, 180)"),
    ("camera_perspective", (float("inf"), 1.0, 0.1, 100.0), "fovy_deg must be finite and in (0, 180)"),
    ("camera_perspective", (45.0, 0.0, 0.1, 100.0), "aspect must be finite and > 0"),
    ("camera_perspective", (45.0, float("inf"), 0.1, 100.0), "aspect must be finite and > 0"),
    ("camera_perspective", (45.0, 1.0, 0.0, 100.0), "znear must be finite and > 0"),
    ("camera_perspective", (45.0, 1.0, float("nan"), 100.0), "znear must be finite and > 0"),
    ("camera_perspective", (45.0, 1.0, 0.1, 0.05), "zfar must be finite and > znear"),
    ("camera_perspective", (45.0, 1.0, 0.1, float("inf")), "zfar must be finite and > znear"),
    ("camera_perspective", (45.0, 1.0, 0.1, 100.0, "invalid"), "clip_space must be 'wgpu' or 'gl'"),
    ("camera_view_proj", ((0, 0, 3), (0, 0, 0), (0, 1, 0), 0.0, 1.0, 0.1, 100.0), "fovy_deg must be finite and in (0, 180)"),
    ("camera_view_proj", ((0, 0, 3), (0, 0, 0), (0, 0, -1), 45.0, 1.0, 0.1, 100.0), "up vector must not be colinear with view direction"),
])
def test_camera_error_strings(oracle, fn, args, msg):              # :28-34, 70-92, 128-167, 203-219
    with pytest.raises(RuntimeError, match=re.escape(msg)):
        getattr(oracle, fn)(*args)


# ---- uniform block: tests/test_t31_integration.py:13-28, tests/test_camera.py:265-328, src/terrain/mod.rs:699-732
@pytest.mark.parametrize("kind", [0, 1])
def test_uniform_lanes(oracle, kind):
    u = oracle.default_uniforms(kind, 256, 192)
    assert u.shape == (44,) and u.dtype == np.float32              # 176 bytes
    assert u[36:40].tolist() == [1.0, 1.0, 1.0, 0.0] and not u[40:].any()
    assert u[35] == 1.0
    sun = np.array([0.5, 1.0, 0.3] if kind == 0 else [0.5, 0.8, 0.6])
    np.testing.assert_allclose(u[32:35], sun / np.linalg.norm(sun), rtol=1e-6)


def test_default_proj_is_perspective_wgpu(oracle):
    W, H = 128, 96                                                 # tests/test_camera.py:298-328
    u = oracle.default_uniforms(0, W, H)
    proj = u[16:32].reshape(4, 4, order="F")
    assert np.allclose(proj, oracle.camera_perspective(45.0, W / H, 0.1, 100.0, "wgpu"), atol=1e-6)
    view = u[:16].reshape(4, 4, order="F")
    assert np.allclose(view, oracle.camera_look_at((3, 2, 3), (0, 0, 0), (0, 1, 0)), atol=1e-6)


def test_look_at_uniforms_match_camera_functions(oracle):
    cam = ((0.0, 0.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 45.0, 0.1, 100.0)   # tests/test_camera.py:265-295
    u = oracle.look_at_uniforms(0, 512, 512, *cam)
    np.testing.assert_allclose(u[:16].reshape(4, 4, order="F"), oracle.camera_look_at(*cam[:3]), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(u[16:32].reshape(4, 4, order="F"), oracle.camera_perspective(45.0, 1.0, 0.1, 100.0), rtol=RTOL, atol=ATOL)
    with pytest.raises(RuntimeError, match=re.escape("fovy_deg must be finite and in (0, 180)")):
        oracle.look_at_uniforms(0, 512, 512, cam[0], cam[1], cam[2], 0.0, 0.1, 100.0)
    with pytest.raises(RuntimeError, match=re.escape("up vector must not be colinear with view direction")):
        oracle.look_at_uniforms(0, 512, 512, cam[0], cam[1], (0, 0, -1), 45.0, 0.1, 100.0)


def test_default_view_known_answer(oracle):
    """eye (3,2,3): s=(.7071,0,-.7071), u=(-.3015,.9045,-.3015), translation (0,~0,-4.6904)."""
    v = oracle.camera_look_at((3, 2, 3), (0, 0, 0), (0, 1, 0))
    np.testing.assert_allclose(v[0, :3], [0.70710677, 0, -0.70710677], atol=1e-6)
    np.testing.assert_allclose(v[1, :3], [-0.30151135, 0.904534, -0.30151135], atol=1e-6)
    np.testing.assert_allclose(v[2, :3], [0.6396021, 0.42640144, 0.6396021], atol=1e-6)
    np.testing.assert_allclose(v[:3, 3], [0, 0, -4.6904154], atol=2e-6)


# ---- LUT bytes and clear colour --------------------------------------------
def test_lut_bytes_match_reference_assets(luts):
    want = {"viridis": "431bb41fd79f015472ceea7282e466c9a4a844399fe06d69719ab849774dc081",
            "magma": "eaf7eb1a81ab34e247bd8156494dd9b843f1655e6efd06ce80f758d66a41dabe",
            "terrain": "8a6d29bd220adb2620741e847ff531a19f51db6b1c0fe615a6b3ca90813105d3"}
    for k, h in want.items():
        assert hashlib.sha256(luts[k].tobytes()).hexdigest() == h   # == sha256 of data/<k>_256.rgba in the reference
    assert luts["viridis"][0, :3].tolist() == [68, 1, 84] and luts["viridis"][255, :3].tolist() == [253, 231, 36]


def test_clear_colour(oracle, luts):
    rgba, vis = oracle.render_terrain(oracle.default_uniforms(0, 32, 24), 32, 24, 4, oracle.SPIKE_DUMMY_HEIGHT, luts["viridis"])
    assert rgba[0, 0].tolist() == [39, 39, 48, 255] and vis[0, 0] == 0   # linear (0.02,0.02,0.03,1) src/terrain/mod.rs:421


def test_render_mesh_layout(oracle):
    verts, idx = oracle.build_grid_xyuv(3)                          # src/terrain/mod.rs:553-598
    assert verts[0].tolist() == [-1.5, -1.5, 0.0, 0.0] and verts[8].tolist() == [1.5, 1.5, 1.0, 1.0]
    assert idx[:6].tolist() == [0, 3, 1, 1, 3, 4]                   # [a,c,b, b,c,d]
