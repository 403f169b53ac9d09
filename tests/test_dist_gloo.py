"""N > 1 path on CPU: two gloo ranks shard the frame by bands, exchange them with the same gather the GPU bench
uses (vulkan_forge_amd/dist.py), and rank 0 must hold the single-rank frame byte for byte.  The per-rank
pixels come from the oracle here (no GPU in this tier); the band logic and the exchange are the product's."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, W, H, G, band, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    import oracle as O
    from vulkan_forge_amd import dist as vdist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lut = np.load(os.path.join(ROOT, "tests", "golden", "colormaps_rgba8.npz"))["viridis"]
    u = O.default_uniforms(O.KIND_SCENE, W, H)
    h = np.random.default_rng(5).random((24, 24), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
    full, _ = O.render_terrain(u, W, H, G, h, lut, rank=rank, nranks=world, band_h=band)
    # pack the owned rows densely in band order, exactly like the HIP path's local-row buffer
    rows = np.concatenate([np.arange(y0, y0 + n) for r, y0, n, _ in vdist.bands(H, world, band) if r == rank])
    assert len(rows) == vdist.local_rows(H, rank, world, band)
    local = torch.from_numpy(np.ascontiguousarray(full[rows]))
    image = torch.zeros((H, W, 4), dtype=torch.uint8) if rank == 0 else None
    dist.barrier()
    vdist.gather_bands(local, image, H, band, dst=0)
    dist.barrier()
    if rank == 0:
        ref, _ = O.render_terrain(u, W, H, G, h, lut)
        q.put(bool(np.array_equal(image.numpy(), ref)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H,band", [(2, 256, 64), (2, 200, 64), (3, 320, 64)])
def test_band_shards_gather_to_the_single_rank_frame(world, H, band):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 160, H, 32, band, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_band_layout_properties():
    from vulkan_forge_amd import dist as vdist
    for H, n, b in ((4096, 8, 64), (4096, 8, 128), (1080, 4, 64), (100, 3, 64), (64, 2, 64)):
        lay = vdist.bands(H, n, b)
        assert sum(r[2] for r in lay) == H and lay[0][1] == 0
        assert all(lay[k][1] + lay[k][2] == lay[k + 1][1] for k in range(len(lay) - 1))
        assert sum(vdist.local_rows(H, r, n, b) for r in range(n)) == H
    assert vdist.local_rows(4096, 3, 8, 64) == 512


# ---- interleaved tiles (the bench's N > 1 layout) -------------------------------------------------------------
def _tile_worker(rank, world, port, W, H, G, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    import oracle as O
    from vulkan_forge_amd import dist as vdist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lut = np.load(os.path.join(ROOT, "tests", "golden", "colormaps_rgba8.npz"))["viridis"]
    u = O.default_uniforms(O.KIND_SCENE, W, H)
    h = np.random.default_rng(5).random((24, 24), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
    full, _ = O.render_terrain(u, W, H, G, h, lut)                      # the pixels come from the oracle (no GPU in this tier)
    ex = vdist.TileExchange(W, H, "cpu", depth=2)
    lay = vdist.tile_layout(W, H, rank, world, ex.skew)
    pad = np.zeros((((H + 63) // 64) * 64, ((W + 63) // 64) * 64, 4), np.uint8)
    ok = True
    nframes = 5
    done = []

    def finish(slot, frame):
        g = ex.finish(slot)
        if rank == 0 and frame >= 0:
            img = np.zeros_like(pad)
            for r in range(world):
                slab = g[r].numpy().view(np.uint8).reshape(-1, 64, 64, 4)
                for k, (tx, ty) in enumerate(vdist.tile_layout(W, H, r, world, ex.skew)):
                    img[ty * 64:(ty + 1) * 64, tx * 64:(tx + 1) * 64] = slab[k]
            done.append(bool(np.array_equal(img[:H, :W], full ^ np.uint8(frame))))

    in_slot = [-1, -1]
    for f in range(nframes):                                            # frame f = the oracle frame xor f: slots must not mix frames
        slot = f % 2
        finish(slot, in_slot[slot])
        pad[:H, :W] = full ^ np.uint8(f)
        slab = np.zeros((ex.stride, 64, 64, 4), np.uint8)
        for k, (tx, ty) in enumerate(lay):
            slab[k] = pad[ty * 64:(ty + 1) * 64, tx * 64:(tx + 1) * 64]
        ex.output(slot).copy_(torch.from_numpy(slab.reshape(-1).view(np.int32)))
        ex.start(slot)
        in_slot[slot] = f
    for k in range(2):
        slot = (nframes + k) % 2
        finish(slot, in_slot[slot])
    dist.barrier()
    if rank == 0:
        q.put(len(done) == nframes and all(done))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H", [(2, 160, 256), (3, 200, 150)])
def test_tile_shards_double_buffered_exchange(world, W, H):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tile_worker, args=(r, world, port, W, H, 32, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_tile_layout_properties():
    from vulkan_forge_amd import dist as vdist
    for W, H, n, skew in ((4096, 4096, 8, None), (4096, 4096, 4, None), (1920, 1080, 8, None), (200, 150, 3, None), (64, 64, 2, None), (100, 100, 1, None),
                          (130, 70, 6, None), (4096, 4096, 8, 3), (1920, 1080, 8, 5), (200, 150, 3, 1),
                          (4096, 4096, 2, vdist.layout_code(0, 2)), (4096, 4096, 4, vdist.layout_code(0, 1)), (1920, 1080, 3, vdist.layout_code(0, 2)),
                          (700, 300, 2, vdist.layout_code(1, 1))):
        skew = vdist.default_skew(n) if skew is None else skew     # default: column stripes (0); skewed maps stay supported
        sh, sk = skew >> 16, skew & 0xFFFF                          # the layout word: stripe width 1 << sh tiles, row-to-row shift sk
        ntx, nty = (W + 63) // 64, (H + 63) // 64
        seen = np.full((nty, ntx), -1)
        sizes = []
        for r in range(n):
            lay = vdist.tile_layout(W, H, r, n, skew)
            sizes.append(len(lay))
            for tx, ty in lay:
                assert seen[ty, tx] == -1 and ((tx >> sh) + sk * ty) % n == r
                seen[ty, tx] = r
            assert [tuple(x) for x in lay] == sorted((tuple(x) for x in lay), key=lambda p: (p[1], p[0]))   # row-major storage order
        assert (seen >= 0).all()
        assert vdist.stride_tiles(W, H, n, skew) == max(sizes)
        if ntx >= n:
            assert max(sizes) - min(sizes) <= nty << sh            # balanced to within one stripe per tile row
    assert vdist.default_skew(8) == 0 and vdist.default_skew(3) == 0 and vdist.default_skew(2) == 0
    assert len(vdist.tile_layout(4096, 4096, 3, 8, 3)) == 512
    assert [vdist.default_stripe_log2(n, 64) for n in (1, 2, 4, 8, 16)] == [0, 2, 1, 0, 0] and vdist.default_stripe_log2(2, 4) == 1


# ---- tile shards stitched in parallel: all-to-all + one band per rank + in-place band gather (dist.BandStitchExchange) ----------
def _band_worker(rank, world, port, W, H, G, q, stripe=0, owner=None):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    import oracle as O
    from vulkan_forge_amd import dist as vdist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lut = np.load(os.path.join(ROOT, "tests", "golden", "colormaps_rgba8.npz"))["viridis"]
    u = O.default_uniforms(O.KIND_SCENE, W, H)
    h = np.random.default_rng(5).random((24, 24), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
    full, _ = O.render_terrain(u, W, H, G, h, lut)                      # the pixels come from the oracle (no GPU in this tier)
    assert vdist.band_exchange_applies(W, H, world, stripe)
    ex = vdist.BandStitchExchange(W, H, "cpu", depth=2, stripe_log2=stripe)
    assert ex.skew == vdist.layout_code(0, stripe)
    if owner is not None:                                               # a load-balanced deal of the same stripes (registered stripe map)
        from vulkan_forge_amd import cabi
        ex.set_layout(cabi.register_stripe_map(owner, stripe, world))
        assert ex.skew & (1 << 20) and vdist.layout_stripe_log2(ex.skew) == stripe
    lay = vdist.tile_layout(W, H, rank, world, ex.skew)
    assert len(lay) == ex.stride

    def stitch(recv, band, rows):                                       # what vf_stitch_tiles_device does for a frame of `rows` rows
        img = band.numpy()
        for r in range(world):
            slab = recv[r].numpy().view(np.uint8).reshape(-1, 64, 64, 4)
            for k, (tx, ty) in enumerate(vdist.tile_layout(W, rows, r, world, ex.skew)):
                img[ty * 64:(ty + 1) * 64, tx * 64:(tx + 1) * 64] = slab[k]

    image = torch.zeros((H, W, 4), dtype=torch.uint8) if rank == 0 else None
    ok = True
    for f in range(5):                                                  # frame f = the oracle frame xor f: slots must not mix frames
        slot = f % 2
        frame = full ^ np.uint8(f)
        slab = np.zeros((ex.stride, 64, 64, 4), np.uint8)
        for k, (tx, ty) in enumerate(lay):
            slab[k] = frame[ty * 64:(ty + 1) * 64, tx * 64:(tx + 1) * 64]
        ex.output(slot).copy_(torch.from_numpy(slab.reshape(-1).view(np.int32)))
        ex.exchange(slot, stitch, image)
        if rank == 0:
            ok = ok and bool(np.array_equal(image.numpy(), frame))
    dist.barrier()
    if rank == 0:
        q.put(ok)
    dist.destroy_process_group()


# (8, 512, 512, 0): the layout of the driver's 8-GPU run in small -- single-column stripes, eight bands of whole tile rows, 64-tile frame
@pytest.mark.parametrize("world,W,H,stripe,owner", [(2, 256, 128, 0, None), (4, 256, 256, 0, None), (2, 512, 128, 2, None), (4, 512, 256, 1, None), (8, 512, 512, 0, None),
                                                     (4, 512, 256, 0, [3, 0, 1, 2, 2, 1, 0, 3]), (2, 512, 128, 1, [1, 0, 0, 1])])
def test_tile_shards_band_stitch_exchange(world, W, H, stripe, owner):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_band_worker, args=(r, world, port, W, H, 32, q, stripe, owner)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_band_exchange_applicability():
    from vulkan_forge_amd import dist as vdist
    assert vdist.band_exchange_applies(4096, 4096, 8) and vdist.band_exchange_applies(4096, 4096, 2) and vdist.band_exchange_applies(4096, 4096, 4)
    assert not vdist.band_exchange_applies(1920, 1080, 8) and not vdist.band_exchange_applies(4096, 4096, 3) and not vdist.band_exchange_applies(200, 150, 2)
    assert vdist.band_exchange_applies(4096, 4096, 2, 2) and vdist.band_exchange_applies(4096, 4096, 4, 1) and not vdist.band_exchange_applies(256, 256, 2, 2)


def test_balanced_stripe_maps():
    """vf_balance_stripes + vf_tile_layout_register_map (host arithmetic): every rank keeps its number of stripes, the deal is a function
    of the times alone, a map's layout word partitions the frame like any other layout and keeps the storage order."""
    from vulkan_forge_amd import cabi, dist as vdist
    rng = np.random.default_rng(11)
    for nstripes, n in ((64, 8), (32, 4), (16, 2), (64, 2), (8, 8)):
        ms = rng.random(nstripes).astype(np.float32) ** 3
        owner = cabi.balance_stripes(ms, n)
        assert np.array_equal(owner, cabi.balance_stripes(ms.copy(), n))
        assert np.bincount(owner, minlength=n).tolist() == [nstripes // n] * n
        loads = np.bincount(owner, weights=ms, minlength=n)
        naive = np.bincount(np.arange(nstripes) % n, weights=ms, minlength=n)
        assert loads.max() <= naive.max() + 1e-6
    with pytest.raises(cabi.VfError):
        cabi.balance_stripes(np.ones(10, np.float32), 4)                  # must divide evenly
    assert cabi.balance_stripes(np.array([np.nan, 1, 2, np.inf], np.float32), 2).tolist() in ([0, 1, 0, 1], [1, 1, 0, 0], [1, 0, 0, 1], [0, 0, 1, 1], [1, 0, 1, 0], [0, 1, 1, 0])
    owner = np.array([3, 0, 1, 2, 2, 1, 0, 3], np.uint8)
    word = cabi.register_stripe_map(owner, 1, 4)
    assert word == cabi.register_stripe_map(owner.copy(), 1, 4) and word & (1 << 20) and vdist.layout_stripe_log2(word) == 1
    W, H = 1024, 200                                                     # 16 tile columns = 8 stripes of 2
    seen = np.full((4, 16), -1)
    for r in range(4):
        lay = vdist.tile_layout(W, H, r, 4, word)
        assert len(lay) == 4 * 4 and [tuple(x) for x in lay] == sorted((tuple(x) for x in lay), key=lambda p: (p[1], p[0]))
        for tx, ty in lay:
            assert owner[tx >> 1] == r and seen[ty, tx] == -1
            seen[ty, tx] = r
    assert (seen >= 0).all()
    with pytest.raises(cabi.VfError):
        vdist.tile_layout(W, H, 0, 2, word)                               # made for four ranks
    with pytest.raises(cabi.VfError):
        vdist.tile_layout(4096, H, 0, 4, word)                            # a wider frame has stripes the table does not name
    with pytest.raises(cabi.VfError):
        cabi.register_stripe_map([0, 5], 0, 2)
    with pytest.raises(cabi.VfError):
        vdist.tile_layout(W, H, 0, 4, (1 << 20) | (63 << 21))            # no such map
    st = vdist.stripe_times(np.arange(8, dtype=np.float32), [(0, 0), (1, 0), (4, 0), (5, 0), (0, 1), (1, 1), (4, 1), (5, 1)], 4, 1)
    assert st.tolist() == [0 + 1 + 4 + 5, 0.0, 2 + 3 + 6 + 7, 0.0]
