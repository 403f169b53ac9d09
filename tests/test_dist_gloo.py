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
