"""The conventions the oracle fixes for behaviour the reference delegates to the GPU driver (DESIGN.md
"Raster conventions"): deterministic sin/cos, sRGB transfer functions, fill rule, culling, clipping,
painter's order, threading and shard invariance."""
import numpy as np
import pytest

from conftest import FILL_CAMERA, heightmap


def test_sincos_accuracy(oracle):
    x = np.concatenate([np.linspace(-8, 8, 400001), np.linspace(-200, 200, 100001)]).astype(np.float32)
    s, c = oracle.sincos(x)
    assert np.abs(s - np.sin(x.astype(np.float64))).max() < 2.5e-7     # WGSL asks for 2^-11 on [-pi, pi]
    assert np.abs(c - np.cos(x.astype(np.float64))).max() < 2.5e-7


def test_srgb_tables_and_encode(oracle):
    dec, thr = oracle.srgb_tables()
    k = np.arange(256) / 255.0
    ref = np.where(k <= 0.04045, k / 12.92, ((k + 0.055) / 1.055) ** 2.4)
    np.testing.assert_allclose(dec, ref, rtol=1e-6)
    # encode == round(255 * oetf(c)) away from the 255 decision thresholds
    c = np.linspace(0, 1, 200001).astype(np.float32)
    c64 = c.astype(np.float64)
    oetf = np.where(c64 <= 0.0031308, 12.92 * c64, 1.055 * c64 ** (1 / 2.4) - 0.055)
    ideal = np.floor(255 * oetf + 0.5)
    got = oracle.srgb_encode(c)
    frac = 255 * oetf + 0.5 - np.floor(255 * oetf + 0.5)
    away = (frac > 1e-3) & (frac < 1 - 1e-3)
    assert np.array_equal(got[away], ideal[away].astype(np.uint8))
    assert np.abs(got.astype(int) - ideal).max() <= 1
    # decode -> encode round-trips every byte; out-of-range and NaN clamp
    assert np.array_equal(oracle.srgb_encode(dec), np.arange(256, dtype=np.uint8))
    assert oracle.srgb_encode(np.array([-1.0, 0.0, 1.0, 7.0, np.nan], np.float32)).tolist() == [0, 0, 255, 255, 0]
    assert np.all(np.diff(thr[1:]) > 0)


def test_lut_unorm_fallback(oracle, luts):
    lin = oracle.lut_to_linear_u8(luts["viridis"])                 # src/colormap/mod.rs:59-79
    s = luts["viridis"][:, :3].astype(np.float64) / 255
    ref = np.where(s <= 0.04045, s / 12.92, ((s + 0.055) / 1.055) ** 2.4)
    assert np.abs(lin[:, :3].astype(int) - np.floor(ref * 255 + 0.5)).max() <= 1
    assert np.array_equal(lin[:, 3], luts["viridis"][:, 3])


def _tri(*pts, z=0.5, w=1.0):
    return [[x, y, z, w] for x, y in pts]


def test_fill_rule_shared_edge_partitions_pixels(oracle):
    """Two front-facing triangles sharing the diagonal of a quad aligned to pixel centres: every pixel centre is
    covered exactly once (top-left rule), whichever triangle comes first."""
    W = H = 16
    # quad spanning pixel centres 2.5 .. 10.5 -> NDC; CCW in Y-up NDC is front-facing
    def ndc(px, py):
        return (2 * px / W - 1, 1 - 2 * py / H)
    a, b, c, d = ndc(2.5, 2.5), ndc(10.5, 2.5), ndc(2.5, 10.5), ndc(10.5, 10.5)
    t0, t1 = _tri(a, c, b), _tri(b, c, d)
    v01 = oracle.raster_triangles(np.array([t0, t1], np.float32), W, H)
    v10 = oracle.raster_triangles(np.array([t1, t0], np.float32), W, H)
    cov = v01 > 0
    assert np.array_equal(cov, v10 > 0)
    # left/top edges (x = 2.5, y = 2.5) are inside, right/bottom (10.5) are not
    exp = np.zeros((H, W), bool); exp[2:10, 2:10] = True
    assert np.array_equal(cov, exp)
    # each triangle alone: disjoint, union = quad
    c0 = oracle.raster_triangles(np.array([t0], np.float32), W, H) > 0
    c1 = oracle.raster_triangles(np.array([t1], np.float32), W, H) > 0
    assert not (c0 & c1).any() and np.array_equal(c0 | c1, exp)


def test_backface_and_degenerate_culled(oracle):
    W = H = 8
    ccw = _tri((-0.9, -0.9), (0.9, -0.9), (0.0, 0.9))
    cw = [ccw[0], ccw[2], ccw[1]]
    line = _tri((-0.9, -0.9), (0.0, 0.0), (0.9, 0.9))
    assert (oracle.raster_triangles(np.array([ccw], np.float32), W, H) > 0).any()
    assert not oracle.raster_triangles(np.array([cw], np.float32), W, H).any()
    assert not oracle.raster_triangles(np.array([line], np.float32), W, H).any()


def test_painters_order_last_wins(oracle):
    W = H = 8
    big = _tri((-1, -1), (1, -1), (-1, 1))
    vis = oracle.raster_triangles(np.array([big, big, big], np.float32), W, H)
    assert set(np.unique(vis)) <= {0, 3} and (vis == 3).any()


def test_near_far_clipping(oracle):
    W = H = 32
    # entirely behind the near plane (z < 0) or beyond far (z > w): nothing
    assert not oracle.raster_triangles(np.array([_tri((-1, -1), (1, -1), (0, 1), z=-0.1)], np.float32), W, H).any()
    assert not oracle.raster_triangles(np.array([_tri((-1, -1), (1, -1), (0, 1), z=1.5)], np.float32), W, H).any()
    # one vertex behind the near plane: the visible part is a quad; its coverage must be a subset of the unclipped one
    full = _tri((-0.8, -0.8), (0.8, -0.8), (0.0, 0.8))
    part = [full[0], full[1], [0.0, 0.8, -0.5, 1.0]]
    cf = oracle.raster_triangles(np.array([full], np.float32), W, H) > 0
    cp = oracle.raster_triangles(np.array([part], np.float32), W, H) > 0
    assert cp.any() and (cp & ~cf).sum() == 0 and cp.sum() < cf.sum()
    # non-finite coordinates drop the primitive
    bad = [full[0], full[1], [np.nan, 0.8, 0.5, 1.0]]
    assert not oracle.raster_triangles(np.array([bad], np.float32), W, H).any()


def test_triangle_path_known_pixels(oracle):
    img = oracle.render_triangle(256, 256)                          # src/lib.rs:72-91, clear WHITE :19
    assert img.shape == (256, 256, 4) and img.dtype == np.uint8
    assert img[0, 0].tolist() == [255, 255, 255, 255] and img[255, 255].tolist() == [255, 255, 255, 255]
    assert (img[..., 3] == 255).all()
    inside = img[128, 128, :3].astype(int)
    assert inside.min() > 100 and not (inside == 255).all()        # centroid: roughly equal mix, sRGB-encoded
    # bottom-left corner region is red-dominated, bottom-right green-dominated, top blue-dominated
    assert img[225, 35, 0] > img[225, 35, 1] and img[225, 220, 1] > img[225, 220, 0] and img[40, 128, 2] > img[40, 128, 0]
    covered = (img[..., :3] != 255).any(axis=2).mean()
    assert 0.30 < covered < 0.34                                    # area 0.5*1.6*1.6/4 = 0.32 of the frame


@pytest.mark.parametrize("kind,cam", [(0, None), (1, None), (1, FILL_CAMERA)])
def test_threads_and_shards_do_not_change_the_frame(oracle, luts, kind, cam):
    W, H, G = 200, 136, 40
    h = heightmap(7, 33, 17) if kind else oracle.SPIKE_DUMMY_HEIGHT
    u = oracle.default_uniforms(kind, W, H) if cam is None else oracle.look_at_uniforms(kind, W, H, *cam)
    r1, v1 = oracle.render_terrain(u, W, H, G, h, luts["magma"], nthreads=1)
    r4, v4 = oracle.render_terrain(u, W, H, G, h, luts["magma"], nthreads=4)
    assert np.array_equal(r1, r4) and np.array_equal(v1, v4)
    assert (v1 > 0).mean() > 0.03
    # band-sharded renders stitch to the full frame
    for nranks, band in ((2, 64), (3, 64), (2, 128)):
        out = np.zeros_like(r1)
        for r in range(nranks):
            rr, _ = oracle.render_terrain(u, W, H, G, h, luts["magma"], rank=r, nranks=nranks, band_h=band)
            rows = ((np.arange(H) // band) % nranks) == r
            out[rows] = rr[rows]
            assert (rr[~rows] == np.array([39, 39, 48, 255], np.uint8)).all()
        assert np.array_equal(out, r1)


# ---- SPEC_T32: the documented-but-unimplemented fragment stage ----------------------------------
def test_spec_t32_fragment_mode_properties(oracle, luts):
    """No reference code exists for this mode (ROADMAP.md:421-436, README.md:128,174-175), so the oracle is checked
    against the properties the documents state: same geometry, normals from the height TEXTURE (forward differences),
    'sun from the east lights the east slopes', Reinhard x/(1+x) in linear before the sRGB store."""
    W, H, G = 160, 120, 48
    lut = luts["viridis"]
    cam = ((0.0, 4.0, 0.01), (0.0, 0.0, 0.0), (0.0, 0.0, -1.0), 45.0, 0.1, 100.0)       # top-down: +x is east on screen
    u = oracle.look_at_uniforms(1, W, H, *cam)
    flat = np.zeros((8, 8), np.float32)
    ref, vis = oracle.render_terrain(u, W, H, G, flat, lut)
    spec, vis2 = oracle.render_terrain(u, W, H, G, flat, lut, shade_mode=oracle.SHADE_SPEC_T32)
    assert np.array_equal(vis, vis2)                                   # the mode only changes colours
    cov = vis > 0
    # flat texture: n = (0,1,0) everywhere -> shade is one constant; colour = reinhard(lut(height) * shade)
    L = np.float32(u[33]) / np.sqrt(np.float32((u[32:35] ** 2).sum()))
    shade = np.float32(0.15) * (np.float32(1) - L) + L
    eotf = lambda b: np.where(b <= 0.04045 * 255, b / 255 / 12.92, ((b / 255 + 0.055) / 1.055) ** 2.4)
    # REFERENCE shades with the analytic normal, SPEC with the flat one: undo each and compare the LUT colour
    lin_spec = eotf(spec[cov][:, :3].astype(np.float64))
    unre = lin_spec / np.maximum(1.0 - lin_spec, 1e-6) / shade          # inverse Reinhard, inverse shade
    assert unre.max() < 1.05 and lin_spec.max() < 0.5                  # x/(1+x) of values <= 1 stays below 1/2
    # east-facing ramp: heights rise towards -x (west), so the slope faces east (+x)
    ramp = np.repeat(np.linspace(1.5, -1.5, 8, dtype=np.float32)[None, :], 8, axis=0)     # -0.43 per texel: a 23 degree slope
    east = u.copy(); west = u.copy()
    east[32:35] = [1.0, 0.35, 0.0]; west[32:35] = [-1.0, 0.35, 0.0]
    a, va = oracle.render_terrain(east, W, H, G, ramp, lut, shade_mode=oracle.SHADE_SPEC_T32)
    b, vb = oracle.render_terrain(west, W, H, G, ramp, lut, shade_mode=oracle.SHADE_SPEC_T32)
    assert np.array_equal(va, vb)
    m = va > 0
    assert a[m][:, :3].astype(int).sum() > 1.5 * b[m][:, :3].astype(int).sum()     # lit vs. ambient only
    # mirrored ramp (slope faces west): the same two suns swap roles
    c, vc = oracle.render_terrain(west, W, H, G, ramp[:, ::-1].copy(), lut, shade_mode=oracle.SHADE_SPEC_T32)
    d, vd = oracle.render_terrain(east, W, H, G, ramp[:, ::-1].copy(), lut, shade_mode=oracle.SHADE_SPEC_T32)
    assert c[vc > 0][:, :3].astype(int).sum() > 1.3 * d[vd > 0][:, :3].astype(int).sum()
    # degenerate 1x1 texture: forward differences vanish, no out-of-range fetch
    one, _ = oracle.render_terrain(u, W, H, G, np.zeros((1, 1), np.float32), lut, shade_mode=oracle.SHADE_SPEC_T32)
    assert np.array_equal(one, spec)
