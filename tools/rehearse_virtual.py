#!/usr/bin/env python3
"""Dress rehearsal of the N-rank C4 frame with N VIRTUAL ranks in ONE process on one GPU.

The driver's 8-GPU command is `bench.py --gpus 8`: one process per GPU.  `bench.py --gpus N --rehearse` runs the same N processes on
one GPU over gloo -- for N <= 6, the most processes the GPU pool lets one user put on a card (its process guard kills the run beyond
that), so the N = 8 path -- single-column stripes (stripe_log2 0), eight bands of eight tile rows, chunks of 64 tiles -- cannot be
rehearsed that way.  Here the N ranks are N handles of one process, every rank sharded exactly as bench.py shards it
(vdist.default_stripe_log2 / layout_code, vf_terrain_set_tile_shard), and the exchange is bench.py's BandStitchExchange with its two
collectives replaced by the device copies they amount to:

    all-to-all   chunk b of rank r's slab  ->  slot r of rank b's receive buffer
    stitch       every rank stitches its band with vf_stitch_tiles_device (the very call bench.py makes)
    gather       band b  ->  rows [b H/N, (b+1) H/N) of the image

The stitched frame must equal a single-rank render byte for byte.  torch.distributed itself at world size 8 is covered on the CPU
(tests/test_dist_gloo.py, gloo) and the RCCL calls by the one-rank communicator tests; what has no rehearsal is xGMI.
Prints one JSON line shaped like bench.py's N > 1 line (`ranks` has N entries); exit code 3 when the frames differ.

usage: rehearse_virtual.py [N=8] [--camera default|fill] [--frames 12]"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # (as bench.py)
ap = argparse.ArgumentParser()
ap.add_argument("n", nargs="?", type=int, default=8)
ap.add_argument("--camera", default="default")
ap.add_argument("--frames", type=int, default=12)
ap.add_argument("--size", type=int, default=4096)
args = ap.parse_args()

import numpy as np   # noqa: E402
import torch         # noqa: E402  (first: one HIP runtime per process)
import vulkan_forge_amd as vf   # noqa: E402
from vulkan_forge_amd import cabi, dist as vdist   # noqa: E402
import importlib.util   # noqa: E402
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)

N, W, H, G = args.n, args.size, args.size, args.size
dev = torch.device("cuda", 0)
lut = vf.colormap_rgba8("viridis")
heights = torch.from_numpy(np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)).to(dev)
u = b.camera_uniforms(args.camera, W, H)
stripe = vdist.default_stripe_log2(N, (W + 63) // 64)
assert vdist.band_exchange_applies(W, H, N, stripe), "the band exchange needs stripes and bands that divide evenly"
layout = vdist.layout_code(0, stripe)
ntx, nty = W // 64, H // 64
chunk_tiles = (nty // N) * (ntx // N)
stride = chunk_tiles * N
band_rows = H // N
words = vdist.TILE_WORDS

ranks = []
for r in range(N):
    # one context for the process, as in bench.py and the drop-in module: its stream and its two plan streams are shared by the handles
    # (round 5 made a context per handle here: 27 streams on 8 hardware queues, plan chains serialised behind other handles' tile
    #  kernels -- ranks at 0.19 ... 0.32 ms where tools/exp_ranks.py measures 0.19 ... 0.20)
    t = cabi.Terrain(W, H, G, lut, lut_is_srgb=True, device=0, share_ctx=ranks[0]["t"] if ranks else None)
    t.set_height_device(heights.data_ptr(), G, G)
    t.set_tile_shard(r, N, layout)
    assert t.local_tiles() == len(vdist.tile_layout(W, H, r, N, layout)) == stride
    slab = torch.zeros(stride * words, dtype=torch.int32, device=dev)
    t.set_output_device(slab.data_ptr())
    t.set_uniforms(u)
    ranks.append({"t": t, "slab": slab, "recv": torch.zeros((N, chunk_tiles * words), dtype=torch.int32, device=dev),
                  "band": torch.zeros((band_rows, W, 4), dtype=torch.uint8, device=dev)})
image = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
stream = torch.cuda.ExternalStream(ranks[0]["t"].stream_handle(), device=dev)      # every handle of the process shares the context's stream
torch.cuda.set_stream(stream)


ex_events = []


def frame(time_exchange=False):
    for R in ranks:
        R["t"].render(stream.cuda_stream)
    for r, R in enumerate(ranks):                                     # all-to-all: chunk b of rank r -> slot r of rank b
        chunks = R["slab"].view(N, chunk_tiles * words)
        for bnd in range(N):
            ranks[bnd]["recv"][r].copy_(chunks[bnd], non_blocking=True)
    for bnd, R in enumerate(ranks):                                   # every rank stitches its band, the root gathers them in place
        if time_exchange:
            x0, x1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            x0.record(stream)
        R["t"].stitch_tiles(R["recv"].data_ptr(), R["band"].data_ptr(), N, layout, chunk_tiles, stream.cuda_stream, height=band_rows)
        image[bnd * band_rows:(bnd + 1) * band_rows].copy_(R["band"], non_blocking=True)
        if time_exchange:
            x1.record(stream)
            ex_events.append((bnd, x0, x1))


for _ in range(args.frames):
    frame()
torch.cuda.synchronize()
# What one rank costs on its own: steady-state frame period, one rank after another (the other ranks idle), measured the way
# tools/exp_ranks.py does -- 30 settle frames (the plan is feedback driven: round 5 measured after 8, on plans still settling, and
# reported 2.76x where exp_ranks.py says 3.55x), then the better of two runs of 40 frames; and the handle's own HIP events over a run.
periods, per_rank = [], []
for r, R in enumerate(ranks):
    t = R["t"]
    # (sharded again: the scheduling feedback starts from nothing, as on a rank of `bench.py --gpus N` -- the interleaved frames above
    #  ran every rank's tile kernels beside the others' and left their plans in another fixed point: 0.25-0.32 ms where a rank that
    #  starts cold settles at 0.19-0.20)
    t.set_tile_shard(r, N, layout)
    t.set_output_device(R["slab"].data_ptr())
    for _ in range(30):
        t.render(stream.cuda_stream)
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40):
            t.render(stream.cuda_stream)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 40 * 1e3)
    periods.append(best)
    t.enable_timing(True, stats=False)
    for _ in range(20):
        t.render(stream.cuda_stream)
    torch.cuda.synchronize()
    tm = t.timings()
    _tile, period = t.frame_times()
    t.enable_timing(True, stats=True)
    t.render(stream.cuda_stream)
    torch.cuda.synchronize()
    pairs = int(t.timings()["blocks_rasterised"])
    t.enable_timing(False)
    per_rank.append({"tile_kernel_ms": float(tm["tile_ms"]), "frame_period_ms": float(np.median(period[1:])), "local_pairs": pairs})
for _ in range(3):
    frame(time_exchange=True)
torch.cuda.synchronize()
for r in range(N):                                                  # (here: the band stitch + the band's copy into the image; the all-to-all is device copies)
    xs = [a.elapsed_time(c) for bnd, a, c in ex_events if bnd == r]
    per_rank[r]["exchange_ms"] = float(np.median(xs))
frame()
torch.cuda.synchronize()
got = image.clone()
single = cabi.Terrain(W, H, G, lut, lut_is_srgb=True, device=0, share_ctx=ranks[0]["t"])
single.set_height_device(heights.data_ptr(), G, G); single.set_uniforms(u); single.set_output_device(image.data_ptr())
for _ in range(30):                                                 # (settled like the ranks above)
    single.render(stream.cuda_stream)
one_gpu = 1e9
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40):
        single.render(stream.cuda_stream)
    torch.cuda.synchronize(); one_gpu = min(one_gpu, (time.perf_counter() - t0) / 40 * 1e3)
equal = bool(torch.equal(got, image))
di = ranks[0]["t"].device_info()
out = {"rehearsal": f"{N} virtual ranks in one process on one GPU (the pool allows at most 6 GPU processes per card): sharding, slabs, chunked all-to-all, "
                    "per-rank band stitch and in-place band gather as bench.py --gpus N runs them, the two collectives as device copies; not a performance number",
       "n_gpus": N, "config": {"workload": f"C4: Scene {W}x{H}, grid={G}, {args.camera} camera", "parallelism":
                               f"64x64 screen tiles in column stripes of {1 << stripe} tile(s) over {N} ranks, all-to-all + one band stitched per rank + bands gathered in place"},
       "gathered_frame_equals_single_rank_frame": equal,
       "ranks": [{"rank": r, "hip_device": 0, "pci_bus_id": f"{di['pci_bus_id']:02x}:{di['pci_device_id']:02x}", "local_tiles": R["t"].local_tiles(),
                  "frame_period_alone_ms": periods[r], **per_rank[r]} for r, R in enumerate(ranks)],
       "one_gpu_frame_period_ms": one_gpu, "slowest_rank_ms": max(periods), "emulated_compute_scaling": one_gpu / max(periods),
       "slowest_rank": int(np.argmax(periods)), "imbalance": max(periods) / float(np.mean(periods)),
       "exchange_ms_max": max(p["exchange_ms"] for p in per_rank), "exchange_hidden": bool(max(p["exchange_ms"] for p in per_rank) <= max(periods)),
       "stripe_log2": stripe, "band_rows": band_rows, "chunk_tiles": chunk_tiles}
print(json.dumps(out), flush=True)
torch.cuda.synchronize()
torch.cuda.set_stream(torch.cuda.default_stream(dev))
single.close()
for R in reversed(ranks):                                           # (rank 0's handle owns the shared context: last)
    R["t"].close()
sys.exit(0 if equal else 3)
