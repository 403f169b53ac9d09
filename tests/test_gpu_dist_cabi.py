"""The multi-GPU exchange behind the C-ABI (include/vf_hip.h: vf_dist_*), on the real backend as far as one GPU allows: a
one-rank RCCL communicator made with vf_dist_unique_id + vf_dist_comm_init, the tile shard of "rank 0 of 1" sent through
ncclSend / ncclRecv to itself by vf_dist_gather_tiles, then stitched -- the result must be the unsharded frame; the same through
vf_dist_exchange_bands (column stripes -> all-to-all + band stitch + in-place band gather: bench.py's default layout).  Runs in a
fresh interpreter without torch: the library resolves RCCL by itself (ROCm's librccl.so.1)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["VF_ROOT"])
from vulkan_forge_amd import cabi
import ctypes as C
lib = cabi.load()
assert lib.vf_dist_available() == 1, lib.vf_last_error()
hip = C.CDLL("libamdhip64.so.7")                        # the runtime libvf_hip.so already loaded (matched by soname)
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
def dmalloc(n):
    p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), n) == 0; assert hip.hipMemset(p, 0, n) == 0; return p.value
luts = np.load(os.path.join(os.environ["VF_ROOT"], "tests", "golden", "colormaps_rgba8.npz"))
W, H, G = 1000, 700, 192                                    # edges cut tiles
rng = np.random.default_rng(11)
h = (rng.random((G, G), dtype=np.float32) - np.float32(0.5)) * np.float32(0.5)
u = np.load(os.environ["VF_UNIFORMS"])
t = cabi.Terrain(W, H, G, luts["viridis"])
t.set_height(h); t.set_uniforms(u)
t.render(); whole = t.read_rgba()
# ---- tiles ----
comm = t.dist_comm_init(t.dist_unique_id(), 0, 1)
ntx, nty = (W + 63) // 64, (H + 63) // 64
stride = ntx * nty + 3                                      # a stride larger than the shard
slab, gathered, image = dmalloc(ntx * nty * 16384), dmalloc(stride * 16384), dmalloc(W * H * 4)
t.set_tile_shard(0, 1, 1)
for rep in range(2):                                        # twice: buffer reuse
    t.set_output_device(slab); t.render()
    t.dist_gather_tiles(comm, 0, gathered, stride)          # own slab -> RCCL -> slot 0 (it did not render in place)
    t.stitch_tiles(gathered, image, 1, 1, stride)
    t.sync()
    out = np.empty((H, W, 4), np.uint8)
    assert hip.hipMemcpy(out.ctypes.data, image, W * H * 4, 2) == 0
    assert np.array_equal(out, whole), rep
# rendered straight into its slot: nothing moves, the result is the same
assert hip.hipMemset(gathered, 0, stride * 16384) == 0
t.set_output_device(gathered); t.render()
t.dist_gather_tiles(comm, 0, gathered, stride)
t.stitch_tiles(gathered, image, 1, 1, stride); t.sync()
assert hip.hipMemcpy(out.ctypes.data, image, W * H * 4, 2) == 0
assert np.array_equal(out, whole)
# argument checks: wrong root, stride too small
for bad in (lambda: t.dist_gather_tiles(comm, 1, gathered, stride), lambda: t.dist_gather_tiles(comm, 0, gathered, 3)):
    try: bad(); raise SystemExit("expected an error")
    except cabi.VfError: pass
# ---- bands ----
t.set_shard(0, 1, 64); t.set_output_device(0)
t.render()
assert hip.hipMemset(image, 0, W * H * 4) == 0
t.dist_gather_bands(comm, 0, image); t.sync()
assert hip.hipMemcpy(out.ctypes.data, image, W * H * 4, 2) == 0
assert np.array_equal(out, whole)
# ---- column stripes -> all-to-all + band stitch + in-place band gather (vf_dist_exchange_bands), on a frame of whole tiles ----
t.close()
W2, H2 = 1024, 640
u2 = np.load(os.environ["VF_UNIFORMS2"])
t = cabi.Terrain(W2, H2, G, luts["viridis"]); t.set_height(h); t.set_uniforms(u2)
t.render(); whole2 = t.read_rgba()
slab2, image2 = dmalloc((W2 // 64) * (H2 // 64) * 16384), dmalloc(W2 * H2 * 4)
t.set_output_device(slab2)
out2 = np.empty((H2, W2, 4), np.uint8)
for rep, layout in enumerate((0, 0, 0, 1 << 16)):           # the handle's staging buffers are reused; last: stripes of two tiles (layout word)
    t.set_tile_shard(0, 1, layout)
    assert hip.hipMemset(image2, 0, W2 * H2 * 4) == 0
    t.render(); t.dist_exchange_bands(comm, 0, image2); t.sync()
    assert hip.hipMemcpy(out2.ctypes.data, image2, W2 * H2 * 4, 2) == 0
    assert np.array_equal(out2, whole2), rep
t.set_tile_shard(0, 1, 0); t.render()
assert t.dist_version() > 20000
# argument checks, made on every rank before anything is posted: wrong root, a skewed shard, a frame that cuts tiles
try: t.dist_exchange_bands(comm, 1, image2); raise SystemExit("expected an error")
except cabi.VfError: pass
t.set_tile_shard(0, 1, 3); t.render()
try: t.dist_exchange_bands(comm, 0, image2); raise SystemExit("expected an error")
except cabi.VfError as e: assert "skew 0" in str(e)
t.close()
t = cabi.Terrain(W, H, G, luts["viridis"]); t.set_height(h); t.set_uniforms(u); t.set_tile_shard(0, 1, 0); t.render()
try: t.dist_exchange_bands(comm, 0, image); raise SystemExit("expected an error")
except cabi.VfError as e: assert "whole tiles" in str(e)
t.dist_comm_destroy(comm)
t.close()
print("CABI GATHER OK")
'''


def test_rccl_gather_through_the_c_abi(oracle, tmp_path):
    import numpy as np
    from conftest import FILL_CAMERA
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    upath = tmp_path / "u.npy"
    np.save(upath, oracle.look_at_uniforms(1, 1000, 700, *FILL_CAMERA))
    upath2 = tmp_path / "u2.npy"
    np.save(upath2, oracle.look_at_uniforms(1, 1024, 640, *FILL_CAMERA))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VF_ROOT=root, VF_UNIFORMS=str(upath), VF_UNIFORMS2=str(upath2))
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "CABI GATHER OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])


def test_eight_virtual_ranks_rehearse_the_drivers_n8_frame():
    """The driver's `bench.py --gpus 8` layout at C4 size -- stripe_log2 0, eight bands of eight tile rows, 64-tile chunks -- as eight
    handles of ONE process (tools/rehearse_virtual.py: the pool's process guard allows six GPU processes per card, so the N = 8
    path cannot be rehearsed as processes): sharding, slabs, chunked all-to-all, per-rank band stitch, in-place gather; the stitched
    frame equals the single-rank frame byte for byte."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "rehearse_virtual.py"), "8", "--frames", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 8 and len(d["ranks"]) == 8 and d["gathered_frame_equals_single_rank_frame"] is True
    assert d["stripe_log2"] == 0 and d["band_rows"] == 512 and d["chunk_tiles"] == 64
    assert all(rk["local_tiles"] == 512 for rk in d["ranks"])
