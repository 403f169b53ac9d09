#!/usr/bin/env python3
"""what would an image-order (per pixel, descending primitive id) visibility walk cost?

Analysis tool (under tests/: it uses the CPU oracle, which tools/ may not): builds tests/walk_model/walk_model.c (which includes the CPU oracle's vertex arithmetic), renders the frame with the
oracle, walks chosen 64 x 64 tiles pixel by pixel, checks every winner against the oracle's visibility buffer and prints the walk's step
counts per pixel.  Runs on the CPU only.

    python tests/walk_model/walk_step0.py --camera default --tiles 34,26 33,26 28,25 --sample 12
"""
import argparse
import ctypes as C
import math
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (analysis only)

NAMES = ["srow", "stest", "brow", "btest", "crow", "ctest", "exact", "tri"]


def build():
    src = os.path.join(ROOT, "tests", "walk_model", "walk_model.c")
    out = os.path.join(ROOT, "build", "libwalk_model.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-std=c11", "-O2", "-fPIC", "-ffp-contract=off", "-fopenmp", "-mavx2", "-mfma", "-shared", src, "-o", out, "-lm"])
    lib = C.CDLL(out)
    f32p, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    lib.wm_setup.argtypes = [f32p, C.c_uint32, C.c_uint32, C.c_uint32, f32p, C.c_uint32, C.c_uint32, C.c_double]
    lib.wm_walk_rect.argtypes = [C.c_int32] * 4 + [u32p, u32p]
    lib.wm_info.argtypes = [C.POINTER(C.c_double)]
    return lib


def uniforms(camera, W, H, pose=None):
    if camera == "fill":
        return oracle.look_at_uniforms(oracle.KIND_SCENE, W, H, (0.0, 2.2, 0.0), (0, 0, 0), (0.0, 0.0, -1.0), 60.0, 0.1, 100.0)
    if camera == "orbit":
        th = 2.0 * math.pi * pose / 64
        eye = (3.0 * math.sqrt(2.0) * math.cos(th), 2.0, 3.0 * math.sqrt(2.0) * math.sin(th))
        return oracle.look_at_uniforms(oracle.KIND_SCENE, W, H, eye, (0, 0, 0), (0, 1, 0), 45.0, 0.1, 100.0)
    return oracle.look_at_uniforms(oracle.KIND_SCENE, W, H, (3.0, 2.0, 3.0), (0, 0, 0), (0, 1, 0), 45.0, 0.1, 100.0)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--camera", default="default", choices=["default", "fill", "orbit"])
    ap.add_argument("--pose", type=int, default=61)
    ap.add_argument("--size", default="4096x4096")
    ap.add_argument("--grid", type=int, default=4096)
    ap.add_argument("--seed", type=int, default=20250816)
    ap.add_argument("--tiles", nargs="*", default=[])
    ap.add_argument("--sample", type=int, default=0, help="also walk every k-th tile in both directions")
    ap.add_argument("--eps", type=float, default=1.0 / 32)
    ap.add_argument("--smooth", action="store_true", help="no noise texture: the analytic surface only")
    args = ap.parse_args(argv)
    W, H = (int(v) for v in args.size.split("x"))
    G = args.grid
    lib = build()
    rng = np.random.default_rng(args.seed)
    height = rng.random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
    if args.smooth:
        height = np.zeros((1, 1), np.float32)
    u = uniforms(args.camera, W, H, args.pose)
    lut = np.zeros((256, 4), np.uint8)
    t0 = time.time()
    _, vis = oracle.render_terrain(u, W, H, G, height, lut, nthreads=os.cpu_count())
    t1 = time.time()
    f32p, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    lib.wm_setup(u.ctypes.data_as(f32p), W, H, G, height.ctypes.data_as(f32p), height.shape[1], height.shape[0], args.eps)
    info = (C.c_double * 8)()
    lib.wm_info(info)
    print(f"oracle frame {t1 - t0:.1f} s; covered {np.count_nonzero(vis) / vis.size:.3f}; rho = {info[0]:.2e} {info[1]:.2e} {info[2]:.2e} world units, cell {info[5]:.2e}, heights [{info[3]:.3f}, {info[4]:.3f}]")
    tiles = [tuple(int(v) for v in t.split(",")) for t in args.tiles]
    if args.sample:
        tiles += [(tx, ty) for ty in range(args.sample // 2, (H + 63) // 64, args.sample) for tx in range(args.sample // 2, (W + 63) // 64, args.sample)]
    tot = np.zeros(len(NAMES), np.float64)
    npx = 0
    bad = 0
    print("tile        covered  " + "  ".join(f"{n:>8s}" for n in NAMES) + "   (mean per pixel)   max ctest  mismatches")
    for tx, ty in tiles:
        x0, y0, x1, y1 = tx * 64, ty * 64, min(tx * 64 + 64, W), min(ty * 64 + 64, H)
        out = np.zeros((y1 - y0, x1 - x0), np.uint32)
        cnt = np.zeros((y1 - y0, x1 - x0, len(NAMES)), np.uint32)
        lib.wm_walk_rect(x0, y0, x1, y1, out.ctypes.data_as(u32p), cnt.ctypes.data_as(u32p))
        ref = vis[y0:y1, x0:x1]
        mism = int(np.count_nonzero(out != ref))
        bad += mism
        m = cnt.reshape(-1, len(NAMES)).mean(axis=0)
        tot += cnt.reshape(-1, len(NAMES)).sum(axis=0)
        npx += out.size
        print(f"({tx:2d},{ty:2d})   {np.count_nonzero(ref) / ref.size:7.3f}  " + "  ".join(f"{v:8.1f}" for v in m) + f"   {cnt[..., 5].max():9d}  {mism}")
    if npx:
        print("all        " + " " * 9 + "  ".join(f"{v:8.1f}" for v in tot / npx) + f"   mismatches {bad} of {npx}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
