"""Screen-space band sharding across the GPUs of one node (one process per GPU, torch.distributed).

The reference is single-device; this is the multi-GPU layer SURVEY.md 8(e) asks for.  Pixel row y belongs
to rank ((y // band_h) % nranks); every rank renders only its bands (the tile kernel never launches a
workgroup for a foreign tile, and blocks that cannot reach an owned tile are never rasterised), keeps them
densely packed ("local rows", band order) and one exchange step moves them to rank 0: each band is a
contiguous band_h*W*4-byte slab of the final image, so rank 0 receives every remote band directly into
its final position -- point-to-point sends that run over the xGMI links to rank 0 in parallel, no staging
copy, no ring.  torch.distributed's "nccl" backend is RCCL on ROCm; "gloo" runs the same code on CPU.
"""
from __future__ import annotations

from typing import List, Tuple


def bands(height: int, nranks: int, band_h: int) -> List[Tuple[int, int, int, int]]:
    """[(owner rank, y0, rows, local_y0)] for every band of the frame, top to bottom."""
    out, local = [], [0] * nranks
    b = 0
    while b * band_h < height:
        y0 = b * band_h
        rows = min(band_h, height - y0)
        r = b % nranks
        out.append((r, y0, rows, local[r]))
        local[r] += rows
        b += 1
    return out


def local_rows(height: int, rank: int, nranks: int, band_h: int) -> int:
    return sum(rows for r, _, rows, _ in bands(height, nranks, band_h) if r == rank)


def gather_bands(local, image, height: int, band_h: int, dst: int = 0, group=None):
    """Move every rank's local rows (tensor [local_rows, W, 4] uint8) into `image` ([H, W, 4]) on rank `dst`.

    All ranks call this.  `image` is only used on `dst`.  Returns the list of outstanding work handles
    already waited on (None when world size is 1)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    layout = bands(height, world, band_h)
    if rank == dst:
        for r, y0, rows, ly0 in layout:
            if r == dst:
                image[y0:y0 + rows].copy_(local[ly0:ly0 + rows], non_blocking=True)
    if world == 1:
        return None
    ops = []
    for r, y0, rows, ly0 in layout:
        if r == dst:
            continue
        if rank == dst:
            ops.append(dist.P2POp(dist.irecv, image[y0:y0 + rows], r, group))
        elif rank == r:
            ops.append(dist.P2POp(dist.isend, local[ly0:ly0 + rows], dst, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return ops
