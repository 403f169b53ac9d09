"""Triangle smoke path of a given library against the oracle at several sizes, with the shape of any difference.
usage (GPU box): python tests/tri_check.py path/to/libvf.so"""
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
import oracle
from vulkan_forge_amd import cabi
lib = cabi.load(sys.argv[1])
ctx = C.c_void_p(); assert lib.vf_ctx_create(0, C.byref(ctx)) == 0
for W, H in ((1920, 1080), (256, 256), (800, 600), (4096, 4096), (1000, 700)):
    a = np.zeros((H, W, 4), np.uint8)
    assert lib.vf_triangle_render(ctx, W, H, a.ctypes.data) == 0
    ref = oracle.render_triangle(W, H)
    d = (a != ref).any(axis=2)
    ys, xs = np.nonzero(d)
    print(os.path.basename(sys.argv[1]), W, H, "differing pixels:", int(d.sum()), [(int(x), int(y), a[y, x].tolist(), ref[y, x].tolist()) for x, y in list(zip(xs, ys))[:4]])
    if d.any():
        inside = (ref[..., :3] != 255).any(axis=2)
        print("   covered pixels:", int(inside.sum()), " wrong AND covered:", int((d & inside).sum()), " wrong AND white in ref:", int((d & ~inside).sum()))
        print("   bbox of wrong pixels: x", int(xs.min()), int(xs.max()), " y", int(ys.min()), int(ys.max()), "  apex x", W // 2)
        # which side of the apex column, and whole 64-pixel groups?
        print("   wrong left of centre:", int((xs < W // 2).sum()), " right:", int((xs >= W // 2).sum()))
        flat = d.reshape(-1); ins = inside.reshape(-1)
        n64 = (flat.size // 64) * 64
        g = flat[:n64].reshape(-1, 64); gi = ins[:n64].reshape(-1, 64)
        full = ((g == gi) & gi.any(axis=1, keepdims=True)).all(axis=1) & g.any(axis=1)
        print("   64-pixel groups with any wrong pixel:", int(g.any(axis=1).sum()), " of which every covered pixel is wrong:", int(full.sum()))
        for y in (int(ys.min()), int((ys.min() + ys.max()) // 2), int(ys.max())):
            row = d[y]; rin = inside[y]
            xw = np.nonzero(row)[0]; xi = np.nonzero(rin)[0]
            print(f"   row {y}: covered x {xi.min()}..{xi.max()}  wrong x {xw.min() if xw.size else None}..{xw.max() if xw.size else None} ({xw.size} px)")
