"""Second opinion on the oracle: an independent float64 NumPy evaluation of the same shader + raster maths
(true sin/cos/pow, no sub-pixel snapping, no fixed-point) must agree with the oracle everywhere except on
pixels whose centre lies within a hair of a triangle edge (where snapping legitimately decides), and
colours must agree within 1 LSB.  This bounds what the deterministic approximations in the oracle
(polynomial sin/cos, 24.8 snapping, threshold sRGB store) can change."""
import numpy as np
import pytest

from conftest import heightmap


def ideal_render(u, W, H, n, height, lut_rgba8):
    u = u.astype(np.float64)
    view = u[:16].reshape(4, 4, order="F"); proj = u[16:32].reshape(4, 4, order="F")
    sun = u[32:35]; exposure = u[35]; spacing = max(u[36], 1e-8); h_range = max(u[37], 1e-8); exag = u[38]
    th, tw = height.shape
    f32 = np.float32
    step = f32(3.0) / (f32(n) - f32(1))
    ii = np.arange(n, dtype=np.float32)
    xs = (f32(-1.5) + ii * step).astype(np.float64)                 # inputs are the same float32 grid values
    uvs = (ii / (f32(n) - f32(1)))
    tx = np.clip(np.floor(uvs * f32(tw)).astype(int), 0, tw - 1); ty = np.clip(np.floor(uvs * f32(th)).astype(int), 0, th - 1)
    X, Z = np.meshgrid(xs, xs)                                        # [j, i]
    Htex = height.astype(np.float64)[ty[:, None], tx[None, :]]
    Hh = Htex + np.sin(X * np.float64(f32(1.3))) * 0.25 + np.cos(Z * np.float64(f32(1.1))) * 0.25
    world = np.stack([X * spacing, Hh * exag, Z * spacing, np.ones_like(X)], -1)
    clip = world @ view.T @ proj.T
    sx = (clip[..., 0] / clip[..., 3] + 1) * W / 2; sy = (1 - clip[..., 1] / clip[..., 3]) * H / 2
    rw = 1 / clip[..., 3]
    dec = lut_rgba8[:, :3].astype(np.float64) / 255
    lut = np.where(dec <= 0.04045, dec / 12.92, ((dec + 0.055) / 1.055) ** 2.4)
    vis = np.zeros((H, W), np.int64); margin = np.full((H, W), np.inf)
    attrs = np.zeros((H, W, 3))
    py, px = np.mgrid[0:H, 0:W]; cx = px + 0.5; cy = py + 0.5
    for j in range(n - 1):
        for i in range(n - 1):
            for t, vid in enumerate((((i, j), (i, j + 1), (i + 1, j)), ((i + 1, j), (i, j + 1), (i + 1, j + 1)))):
                P = [(sx[b, a], sy[b, a]) for a, b in vid]
                area2 = (P[1][0] - P[0][0]) * (P[2][1] - P[0][1]) - (P[1][1] - P[0][1]) * (P[2][0] - P[0][0])
                if area2 >= 0:
                    continue
                x0 = max(int(np.floor(min(p[0] for p in P))) - 1, 0); x1 = min(int(np.ceil(max(p[0] for p in P))) + 1, W)
                y0 = max(int(np.floor(min(p[1] for p in P))) - 1, 0); y1 = min(int(np.ceil(max(p[1] for p in P))) + 1, H)
                if x0 >= x1 or y0 >= y1:
                    continue
                gx, gy = cx[y0:y1, x0:x1], cy[y0:y1, x0:x1]
                def E(a, b):
                    return -((P[b][0] - P[a][0]) * (gy - P[a][1]) - (P[b][1] - P[a][1]) * (gx - P[a][0]))
                e0, e1, e2 = E(1, 2), E(2, 0), E(0, 1)
                edge_len = [np.hypot(P[b][0] - P[a][0], P[b][1] - P[a][1]) + 1e-30 for a, b in ((1, 2), (2, 0), (0, 1))]
                dist = np.minimum(np.minimum(e0 / edge_len[0], e1 / edge_len[1]), e2 / edge_len[2])   # signed px distance to the nearest edge
                inside = dist > 0
                sub_m = margin[y0:y1, x0:x1]
                # pixels this primitive (nearly) touches become ambiguous if within 0.02 px of an edge
                near = np.abs(dist) < 0.02
                prim = 2 * (j * (n - 1) + i) + t
                l = np.stack([e0, e1, e2], -1) / (-area2)
                q = l * np.array([rw[b, a] for a, b in vid])
                Q = q.sum(-1, keepdims=True)
                va = np.array([[Hh[b, a], X[b, a], Z[b, a]] for a, b in vid])
                at = (q @ va) / Q
                sv = vis[y0:y1, x0:x1]; sa = attrs[y0:y1, x0:x1]
                sv[inside] = prim + 1; sa[inside] = at[inside]
                sub_m[inside] = np.where(near[inside], 0.0, np.inf)     # a later clean cover clears older ambiguity
                sub_m[near & ~inside] = 0.0
    # fragment stage
    hgt, x, z = attrs[..., 0], attrs[..., 1], attrs[..., 2]
    t = np.clip(0.5 + hgt / (2 * h_range), 0, 1)
    c = t * 256 - 0.5; i0 = np.floor(c); f = c - i0
    a0 = np.clip(i0, 0, 255).astype(int); a1 = np.clip(i0 + 1, 0, 255).astype(int)
    col = lut[a0] * (1 - f[..., None]) + lut[a1] * f[..., None]
    k13, k11 = np.float64(f32(1.3)), np.float64(f32(1.1))
    dhdx = k13 * np.cos(x * k13) * 0.25; dhdz = -k11 * np.sin(z * k11) * 0.25
    nrm = np.stack([-dhdx, np.ones_like(x), -dhdz], -1); nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    L = sun / np.linalg.norm(sun)
    lam = np.clip(nrm @ L, 0, 1)
    rgb = np.clip(col * exposure * (0.15 * (1 - lam) + lam)[..., None], 0, 1)
    srgb = np.where(rgb <= 0.0031308, 12.92 * rgb, 1.055 * rgb ** (1 / 2.4) - 0.055)
    out = np.empty((H, W, 4), np.float64); out[..., :3] = srgb * 255; out[..., 3] = 255
    clear = np.array([0.02, 0.02, 0.03]); cs = np.where(clear <= 0.0031308, 12.92 * clear, 1.055 * clear ** (1 / 2.4) - 0.055) * 255
    out[vis == 0, :3] = cs
    return out, vis, margin == 0.0


@pytest.mark.parametrize("kind,W,H,G,cmap", [(0, 96, 64, 10, "viridis"), (1, 80, 60, 12, "terrain"), (1, 64, 64, 7, "magma")])
def test_oracle_matches_float64_evaluation(oracle, luts, kind, W, H, G, cmap):
    h = heightmap(3, 9, 6) * 0.5 if kind else oracle.SPIKE_DUMMY_HEIGHT
    u = oracle.default_uniforms(kind, W, H)
    rgba, vis = oracle.render_terrain(u, W, H, G, h, luts[cmap])
    ideal, ivis, ambiguous = ideal_render(u, W, H, max(G, 2), h, luts[cmap])
    clean = ~ambiguous
    assert clean.mean() > 0.97
    assert np.array_equal(vis[clean], ivis[clean].astype(np.uint32))
    same = clean & (vis == ivis)
    # colour: |oracle byte - ideal real value| < 1 everywhere the same fragment is shaded
    d = np.abs(rgba[..., :3].astype(np.float64) - ideal[..., :3])[same]
    assert d.max() < 1.0, d.max()
    assert (rgba[..., 3] == 255).all()
    assert (vis > 0).mean() > 0.03
