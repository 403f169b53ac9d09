"""Screen-space sharding across the GPUs of one node (one process per GPU, torch.distributed).

Two shard shapes (both new: the reference is single-device):

* interleaved 64 x 64 tiles (`TileExchange`, the bench's N > 1 path): tile (tx, ty) belongs to rank
  (tx + skew * ty) % nranks, so the expensive tiles of a frame (they cluster along the terrain's silhouette) spread over
  all ranks.  A rank keeps its tiles densely packed (tile-major); one exchange step moves each rank's slab to rank 0
  (point-to-point, every sender on its own xGMI link, no ring), where `vf_stitch_tiles_device` writes the frame.  The
  exchange is asynchronous and double-buffered: frame k travels while frame k+1 renders.
* 64-row bands (`gather_bands`): band b belongs to rank b % nranks; bands are contiguous slabs of the final image, so
  rank 0 receives them in place without a stitch pass.  Balanced only when the work is spread over the frame's height.

Band sharding in detail: pixel row y belongs
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


# ---- interleaved tiles ------------------------------------------------------------------------------------------
TILE = 64
TILE_WORDS = TILE * TILE          # RGBA8 pixels (32-bit words) per tile slot


def default_skew(nranks: int) -> int:
    """Row-to-row shift of the tile -> rank map.  0: rank r owns every nranks-th tile COLUMN (vertical stripes, interleaved).
    A terrain block projects onto a tall, narrow streak, so stripes keep most blocks out of most ranks -- the per-frame set-up pass
    (vertex stage, culling) only runs for blocks that reach a rank's tiles, and with stripes that is ~1/5 of the grid at 8 ranks where
    a skewed map (3, 5, ...) leaves ~3/4 -- while interleaving at single-tile width still spreads the silhouette's heavy tiles over
    all ranks (measured on C4, one rank of eight: 0.28 ms per frame with stripes, 0.39 with skew 3; tools/exp_ranks.py)."""
    return 0


def layout_code(skew: int = 0, stripe_log2: int = 0) -> int:
    """The layout word the C-ABI takes as `skew` (VF_TILE_LAYOUT): tile (tx, ty) -> rank ((tx >> stripe_log2) + skew * ty) % nranks."""
    return int(skew) | (int(stripe_log2) << 16)


def default_stripe_log2(nranks: int, ntx: int = 0) -> int:
    """log2 of the width, in tiles, of the column stripes dealt to the ranks: a period of eight tile columns -- stripes of 4 tiles for 2
    ranks, 2 for 4, 1 from 8 ranks on (C4, frame period of the slowest rank against one GPU: 2 ranks 1.55 -> 1.68x, 4 ranks 2.47 ->
    2.63x; with 8 ranks wider stripes lose, 3.7 -> 3.6x: profiles/r04_stripes.log).  A wider stripe keeps more of a block's tall, narrow
    footprint inside one rank (fewer blocks set up and drawn by several ranks); a narrower one spreads the silhouette's heavy tiles
    better.  `ntx` (tile columns of the frame), when given, caps the width so that every rank still owns a stripe."""
    sh = 2 if nranks <= 2 else 1 if nranks <= 4 else 0
    if nranks <= 1:
        return 0
    while sh > 0 and ntx and (nranks << sh) > ntx:
        sh -= 1
    return sh


def layout_stripe_log2(layout: int) -> int:
    """log2 of the stripe width of a layout word (a plain VF_TILE_LAYOUT word or a registered stripe map's)."""
    return (int(layout) >> 16) & 0xF


def stripe_times(tile_ms, tiles, nstripes: int, stripe_log2: int):
    """Per-stripe sums of a rank's per-tile times (`tiles`: its (tx, ty) list in storage order, `tile_ms` in the same order)."""
    import numpy as np
    tiles = np.asarray(tiles).reshape(-1, 2)
    return np.bincount(tiles[:, 0] >> stripe_log2, weights=np.asarray(tile_ms, np.float64)[:len(tiles)], minlength=nstripes).astype(np.float32)[:nstripes]


def balanced_layout(stripe_ms, nranks: int, stripe_log2: int) -> int:
    """Per-stripe times (summed over all ranks: identical on every rank) -> the layout word of the load-balanced stripe map:
    vf_balance_stripes (heaviest stripe first to the least loaded rank with room; every rank keeps the same number of stripes, so the
    exchange's sizes do not change) + vf_tile_layout_register_map."""
    from . import cabi
    return cabi.register_stripe_map(cabi.balance_stripes(stripe_ms, nranks), stripe_log2, nranks)


def tile_layout(width: int, height: int, rank: int, nranks: int, skew: int):
    """(n, 2) array of (tx, ty): the tiles of `rank` in storage order (the library's vf_tile_layout; host arithmetic)."""
    from . import cabi
    return cabi.tile_layout(width, height, rank, nranks, skew)


def stride_tiles(width: int, height: int, nranks: int, skew: int) -> int:
    """Tile slots per rank in the gather buffer = the largest shard."""
    return max(len(tile_layout(width, height, r, nranks, skew)) for r in range(nranks))


class TileExchange:
    """Double-buffered gather of tile shards to rank `dst`.

    Every rank renders frame k into `output(k % depth)` (int32 tensor, stride*4096 words; on `dst` this is its own slot of
    the gather buffer, so the root's tiles are never copied), calls `start(slot)` and goes on with the next frame;
    `finish(slot)` -- called before the slot is rendered into again, and for every slot at the end -- makes the current
    stream wait for that exchange and returns the gather buffer [nranks, stride*4096] on `dst` (None elsewhere) for the
    caller to stitch.  Works on CUDA tensors with the nccl (RCCL) backend and on CPU tensors with gloo.
    """

    def __init__(self, width, height, device, depth=2, dst=0, skew=None, group=None):
        import torch
        import torch.distributed as dist
        self.dist, self.group, self.dst = dist, group, dst
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.skew = default_skew(self.world) if skew is None else skew
        self.stride = stride_tiles(width, height, self.world, self.skew)
        self.depth = depth
        words = self.stride * TILE_WORDS
        if self.rank == dst:
            self.gathered = [torch.zeros((self.world, words), dtype=torch.int32, device=device) for _ in range(depth)]
            self.local = [g[dst] for g in self.gathered]
        else:
            self.gathered = [None] * depth
            self.local = [torch.zeros(words, dtype=torch.int32, device=device) for _ in range(depth)]
        self.pending = [None] * depth

    def output(self, slot):
        return self.local[slot]

    def start(self, slot):
        if self.world == 1:
            return
        dist = self.dist
        if self.rank == self.dst:
            ops = [dist.P2POp(dist.irecv, self.gathered[slot][r], r, self.group) for r in range(self.world) if r != self.dst]
        else:
            ops = [dist.P2POp(dist.isend, self.local[slot], self.dst, self.group)]
        self.pending[slot] = dist.batch_isend_irecv(ops)

    def finish(self, slot):
        if self.pending[slot] is not None:
            for w in self.pending[slot]:
                w.wait()
            self.pending[slot] = None
        return self.gathered[slot]


# ---- tile shards, stitched in parallel: all-to-all + per-rank band stitch + in-place band gather ------------------------
def band_exchange_applies(width: int, height: int, nranks: int, stripe_log2: int = 0) -> bool:
    """`BandStitchExchange` needs column stripes that divide evenly and bands of whole tile rows: ntx a multiple of nranks * stripe width,
    nty a multiple of nranks."""
    ntx, nty = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    return nranks >= 1 and width % TILE == 0 and height % TILE == 0 and ntx % (nranks << stripe_log2) == 0 and nty % nranks == 0


class BandStitchExchange:
    """Tile shards (column stripes, skew 0) to a row-major frame on rank `dst` WITHOUT a whole-frame stitch on that rank.

    `TileExchange` ends in `vf_stitch_tiles_device` on rank `dst`: a copy of the whole frame (128 MiB of HBM traffic at
    4096 x 4096) that only that rank pays -- +38 us on a rank of eight whose frame takes 0.22 ms, wherever the copy is queued
    (tools/exp_rank0_stitch.py).  Here the stitch is sharded like the rendering:

      1. all-to-all: the frame is cut into `nranks` horizontal bands of whole tile rows; a rank's slab (its tiles, row-major by
         (ty, tx)) holds the tiles of band b contiguously, and sends that chunk to rank b -- an eighth of a slab per peer, every
         xGMI link busy in both directions, nobody's links a hot spot;
      2. every rank stitches ITS band (a frame of H / nranks rows, the same kernel): 1 / nranks of the copy each;
      3. the bands are contiguous slabs of the final image: rank `dst` receives them in place (`dist.gather` into views of the
         image), no further pass.

    Double-buffered like `TileExchange`; `stitch(recv, band_image, band_rows)` is the caller's (the C-ABI's stitch kernel on the
    GPU, NumPy in the CPU tests).  Works on CUDA tensors with nccl (RCCL) and on CPU tensors with gloo."""

    def __init__(self, width, height, device, depth=2, dst=0, group=None, stripe_log2=0, layout=None):
        import torch
        import torch.distributed as dist
        self.dist, self.group, self.dst, self.torch = dist, group, dst, torch
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if layout is not None:
            stripe_log2 = layout_stripe_log2(layout)       # a registered stripe map (balanced_layout): same sizes, other owners
        if not band_exchange_applies(width, height, self.world, stripe_log2):
            raise ValueError("BandStitchExchange needs stripes that divide the tile columns evenly among the ranks and tile rows that divide by their number")
        # `skew`: the layout word (column stripes); may be replaced by a registered map's word of the same stripe width (set_layout)
        self.W, self.H, self.skew, self.depth = width, height, (layout_code(0, stripe_log2) if layout is None else int(layout)), depth
        ntx, nty = width // TILE, height // TILE
        self.band_rows = height // self.world                         # pixel rows per band
        self.chunk_tiles = (nty // self.world) * (ntx // self.world)  # tiles one rank holds of one band
        self.stride = self.chunk_tiles * self.world                   # tiles per rank (= its slab)
        words = self.stride * TILE_WORDS
        self.local = [torch.zeros(words, dtype=torch.int32, device=device) for _ in range(depth)]
        self.recv = [torch.zeros((self.world, self.chunk_tiles * TILE_WORDS), dtype=torch.int32, device=device) for _ in range(depth)]
        self.band = [torch.zeros((self.band_rows, width, 4), dtype=torch.uint8, device=device) for _ in range(depth)]

    def output(self, slot):
        return self.local[slot]

    def set_layout(self, layout):
        """Another deal of the same stripes (a registered stripe map: every rank keeps its number of stripes, so nothing is resized)."""
        assert layout_stripe_log2(layout) == layout_stripe_log2(self.skew)
        self.skew = int(layout)

    def exchange(self, slot, stitch, image):
        """Frame in `output(slot)` -> rows of `image` ((H, W, 4) uint8, used on `dst`).  Ordered on the caller's current stream
        (the collectives make it wait); returns when everything is queued (GPU) or done (CPU)."""
        dist = self.dist
        if self.world > 1:
            dist.all_to_all_single(self.recv[slot].view(-1), self.local[slot], group=self.group)
        else:
            self.recv[slot].view(-1).copy_(self.local[slot])
        stitch(self.recv[slot], self.band[slot], self.band_rows)      # [nranks][chunk_tiles] tile slots -> band_rows x W pixels
        b = self.band_rows
        if self.world == 1:
            image.copy_(self.band[slot])
        elif self.rank == self.dst:
            dist.gather(self.band[slot], gather_list=[image[r * b:(r + 1) * b] for r in range(self.world)], dst=self.dst, group=self.group)
        else:
            dist.gather(self.band[slot], dst=self.dst, group=self.group)
