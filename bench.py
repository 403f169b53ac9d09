#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config.

  metric   Mpix/s terrain shade: W*H*frames / wall time, RGBA8 complete in HBM (rank 0 holds the gathered frame)
  workload c4 (default): C4 of SURVEY.md 8(d): Scene 4096x4096, grid=4096, 4096x4096 R32F heightmap
           (np.random.default_rng(20250816).random(float32)*0.5-0.25), default camera eye (3,2,3), viridis.
           A "step" = one frame: k_block_boxes + k_block_setup + k_plan + k_plan_sort (on the handle's side stream: they only
           touch plan state and overlap the previous frame's tile kernel) + k_clear + k_tile (+ gather to rank 0 + stitch
           when N > 1).
           c5: BASELINE config 5 -- 64 camera poses on the default camera's orbit over one grid=2048 terrain, 1920x1080
           (SURVEY.md 8(d)); a "step" = every rank renders its next pose (rank r takes poses k = r mod N; replicas, no
           collective, as python/tools/determinism_harness.py:39-62 runs independent renders).
  N > 1    one process per GPU (torch.distributed, backend nccl = RCCL).  c4: screen-tile split -- 64x64 tile (tx, ty)
           belongs to rank (tx + skew*ty) % N; every rank renders only its tiles into a tile-major slab, the slabs go to
           rank 0 point-to-point over xGMI (vulkan_forge_amd/dist.py::TileExchange; --cabi-gather: the library's own
           vf_dist_gather_tiles on an RCCL communicator made through the C-ABI) and vf_stitch_tiles_device writes the
           frame.  The exchange is double-buffered: frame k travels while frame k+1 renders; every one of the K timed
           frames is rendered, gathered and stitched inside the timed region (--serial: no overlap).  Total work is fixed,
           so scaling is "strong".

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8 ...          (no launcher needed: spawns python -m torch.distributed.run itself, before any GPU call)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (fields: see the driver contract; plus "roofline", "roofline_fragment" and "cpu_baseline").
"""
from __future__ import annotations

import argparse
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Before the HIP runtime starts (torch import): hardware queues for this process's streams.  A handle draws on one stream and plans /
# sets up the next frame on two more; torch.distributed and the exchange add theirs; with the default of 4 queues two of them share
# one and the next frame's set-up stops hiding under this frame's tile kernel (include/vf_hip.h, vf_ctx_stream).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SETTLE = 24                     # untimed frames of a new camera before the warm-up (see timed()): the plan's feedback settles, the handle probes its two line loops (frames 4..19)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["c4", "c5"], default="c4")
    ap.add_argument("--size", type=int, default=4096, help="c4: frame width = height")
    ap.add_argument("--grid", type=int, default=None, help="default: 4096 (c4), 2048 (c5)")
    ap.add_argument("--camera", choices=["default", "fill"], default="default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary measurements (other camera, fragment stage, cold frame)")
    ap.add_argument("--check", action="store_true", help="after timing, compare the gathered frame with a single-rank render "
                                                         "(always done at N > 1: a scaling number is only printed for a frame that was proven)")
    ap.add_argument("--no-api-latency", action="store_true",
                    help="skip the end-to-end timing of the drop-in API (TerrainSpike / Scene render_png and render_rgba incl. read-back and PNG "
                         "encode at C2, C3 and C4; a few hundred ms)")
    ap.add_argument("--serial", action="store_true", help="N>1: finish each frame's exchange before rendering the next (no overlap)")
    ap.add_argument("--cabi-gather", action="store_true",
                    help="N>1: exchange through the library's own RCCL entry points (vf_dist_comm_init / vf_dist_gather_tiles) "
                         "instead of torch.distributed's point-to-point calls")
    ap.add_argument("--root-stitch", action="store_true",
                    help="N>1: every slab to rank 0, which stitches the whole frame (round 2's exchange) instead of all-to-all + one band "
                         "stitched per rank + in-place band gather")
    ap.add_argument("--stripe-log2", type=int, default=None,
                    help="N>1: log2 of the width, in tiles, of the column stripes dealt to the ranks (default: a period of eight tile columns -- "
                         "4 tiles for 2 ranks, 2 for 4, 1 from 8 on; vulkan_forge_amd/dist.py::default_stripe_log2)")
    ap.add_argument("--no-balance", action="store_true",
                    help="N>1: keep the round-robin deal of the column stripes (default: after the settle frames the stripes are dealt again by "
                         "their measured times -- vf_terrain_tile_times, summed over the ranks, vf_balance_stripes -- and the plan settles once more)")
    ap.add_argument("--rehearse", action="store_true",
                    help="N>1 dress rehearsal on ONE GPU: every rank uses device 0 and the exchange runs over gloo through host "
                         "memory (RCCL refuses two ranks on one device); exercises sharding, exchange and reporting, not xGMI")
    return ap.parse_args()


def spawn_ranks(n):
    """--gpus N > 1 without a launcher: start N fresh rank processes and relay what rank 0 prints.  Decided before torch.cuda,
    the HIP library or a process group exist in this process -- it never touches the GPU and only waits for its children."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def look_at_uniforms(W, H, eye, target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fovy=45.0):
    """Uniform block through the product's own host code (the drop-in module), not the oracle."""
    import numpy as np
    import vulkan_forge_amd as vf
    view = vf.camera_look_at(eye, target, up)
    proj = vf.camera_perspective(fovy, W / H, 0.1, 100.0, "wgpu")
    u = np.zeros(44, np.float32)
    u[:16] = view.T.reshape(-1)                                   # column-major
    u[16:32] = proj.T.reshape(-1)
    sun = np.array([0.5, 0.8, 0.6], np.float32)                  # Scene keeps Globals::default (src/terrain/mod.rs:190)
    u[32:35] = sun * (np.float32(1.0) / np.sqrt(np.float32((sun * sun).sum())))
    u[35] = 1.0
    u[36:39] = [1.0, 1.0, 1.0]
    return u


def camera_uniforms(name, W, H):
    if name == "fill":                                           # frame-filling top-down camera
        return look_at_uniforms(W, H, (0.0, 2.2, 0.0), up=(0.0, 0.0, -1.0), fovy=60.0)
    return look_at_uniforms(W, H, (3.0, 2.0, 3.0))


def orbit_uniforms(k, W, H, nposes=64):
    """C5 pose k: eye on the default camera's orbit, theta_k = 2 pi k / 64; k = 8 is the default eye (3,2,3)."""
    th = 2.0 * math.pi * k / nposes
    return look_at_uniforms(W, H, (3.0 * math.sqrt(2.0) * math.cos(th), 2.0, 3.0 * math.sqrt(2.0) * math.sin(th)))


def host_cpus():
    """(os.cpu_count(), CPUs this process may run on, cgroup CPU quota or None, model name)."""
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = os.cpu_count() or 1
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(p)
    except (OSError, ValueError):
        pass
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return os.cpu_count() or 1, aff, quota, model


def toolchain():
    """the compiler the library in use was built with and its code-generation switches, as __graft_entry__.build() recorded them next to
    the library (nothing is executed here: under rocprofv3 --pmc the GPU is live before this program starts, and a GPU process must not
    start another program in its place) -- a compiler bump shows up in the bench line"""
    try:
        info = json.load(open(os.path.join(ROOT, "vulkan_forge_amd", "build_info.json")))
        return {"hipcc": info.get("hipcc"), "tuning_flags": info.get("tuning_flags"), "tuning_flags_applied": info.get("tuning_flags_applied", True),
                "built_lib_sha256": info.get("lib_sha256"), **({"note": info["note"]} if info.get("note") else {})}
    except Exception as e:  # noqa: BLE001
        return {"hipcc": None, "error": repr(e)}


def lib_sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))                         # (nothing above imported torch or loaded the HIP library)

    import numpy as np
    import torch                                                  # first: the HIP runtime it bundles is the one shared below
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible", file=sys.stderr)
        sys.exit(1)
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import vulkan_forge_amd as vf
    from vulkan_forge_amd import cabi, dist as vdist

    c5 = args.workload == "c5"
    W, H = (1920, 1080) if c5 else (args.size, args.size)
    G = args.grid or (2048 if c5 else 4096)
    lut = vf.colormap_rgba8("viridis")                            # the product's registry (src/colormap/mod.rs), not a test fixture
    rng = np.random.default_rng(20250817 if c5 else 20250816)
    height_host = rng.random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
    d_height = torch.from_numpy(height_host).to(dev)              # inputs resident in HBM before the timed region

    main_handle = []

    def new_handle():
        # one context per process and device, like the drop-in module's (its stream and the side streams handles borrow are made once)
        h = cabi.Terrain(W, H, G, lut, lut_is_srgb=True, device=local_rank, share_ctx=main_handle[0] if main_handle else None)
        if not main_handle:
            main_handle.append(h)
        h.set_height_device(d_height.data_ptr(), G, G)
        return h

    t = new_handle()
    # An explicit stream, made torch's current one: the raw handle of torch's DEFAULT stream is 0, which the C-ABI reads as "use the
    # context's own stream" -- torch-side events and collectives would then be ordered against a stream nothing renders on.
    # The library's own stream, and as torch's current one -- not one stream more (VF_BENCH_OWN_STREAM=1: a torch stream instead).
    render_stream = torch.cuda.Stream(dev) if os.environ.get("VF_BENCH_OWN_STREAM") else torch.cuda.ExternalStream(t.stream_handle(), device=dev)
    torch.cuda.set_stream(render_stream)
    stream = render_stream.cuda_stream
    image = torch.empty((H, W, 4), dtype=torch.uint8, device=dev) if (rank == 0 or c5) else None
    depth = 1 if args.serial else 2
    share = 1.0
    ex = comm = None
    banded = cabi_bands = False
    if world == 1 or c5:
        t.set_output_device(image.data_ptr())                    # whole frames: render straight into the frame
    else:
        # rehearsal: same exchange code over gloo through host memory (the device slabs are copied out and back in)
        # Default: the stitch is sharded like the rendering (all-to-all, every rank stitches one band, bands gathered in place) when the
        # tile grid divides by the ranks; --root-stitch / --cabi-gather: every slab to rank 0, which stitches the whole frame.
        stripe = vdist.default_stripe_log2(world, (W + 63) // 64) if args.stripe_log2 is None else args.stripe_log2
        banded = vdist.band_exchange_applies(W, H, world, stripe) and not args.root_stitch
        cabi_bands = banded and args.cabi_gather and not args.rehearse     # the same exchange through the library's own RCCL calls
        xdev = "cpu" if args.rehearse else dev
        ex = (vdist.BandStitchExchange(W, H, xdev, depth=depth, stripe_log2=stripe) if banded else
              vdist.TileExchange(W, H, xdev, depth=depth, skew=vdist.layout_code(0, stripe)))
        t.set_tile_shard(rank, world, ex.skew)
        assert t.local_tiles() == len(vdist.tile_layout(W, H, rank, world, ex.skew))
        share = t.local_tiles() / float(((W + 63) // 64) * ((H + 63) // 64))
        if args.rehearse:
            dev_local = torch.zeros(ex.stride * vdist.TILE_WORDS, dtype=torch.int32, device=dev)
            dev_gathered = torch.zeros((world, (ex.chunk_tiles if banded else ex.stride) * vdist.TILE_WORDS), dtype=torch.int32, device=dev) if (rank == 0 or banded) else None
            dev_band = torch.zeros((ex.band_rows, W, 4), dtype=torch.uint8, device=dev) if banded else None
            host_image = torch.zeros((H, W, 4), dtype=torch.uint8) if (banded and rank == 0) else None
        if args.cabi_gather and not args.rehearse:
            # an RCCL communicator of the library's own (vf_dist_comm_init); the 128-byte id travels over the host-side group
            uid = [t.dist_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            comm = t.dist_comm_init(uid[0], rank, world)
    frame_no = [0]
    # N > 1: the exchange and rank 0's stitch run on a stream of their own, ordered by events -- frame k travels and is stitched
    # (an HBM-bound copy of the whole frame) while frame k + 1 is being drawn on the render stream; a slot's next render waits for
    # the event that says its previous frame has left it.  (Rehearsal: host copies between the steps, no overlap to speak of.)
    side = torch.cuda.Stream(dev) if ex is not None else None
    ev_drawn = [torch.cuda.Event() for _ in range(depth)] if side is not None else None
    ev_left = [torch.cuda.Event() for _ in range(depth)] if side is not None else None

    # N > 1: how long a frame's exchange (all-to-all, band stitch, band gather -- everything on the side stream) takes on this rank,
    # from a pair of events around it; collected in the timed region only
    collect_ex = [False]
    ex_events = []

    pose = [rank]

    def pose_batch(n):
        """this rank's next n poses as one (n, 44) block of uniforms (replicas: rank r takes the poses k = r mod N)"""
        u = np.stack([orbit_uniforms((pose[0] + k * world) % 64, W, H) for k in range(n)])
        pose[0] += n * world
        return u

    def step():
        if c5:                                                       # one pose through the batch entry point (the settle frames)
            t.render_batch(pose_batch(1), None, stream)
            return
        if world == 1:
            t.render(stream)
            return
        slot = frame_no[0] % depth
        frame_no[0] += 1
        if ex.pending_frame[slot]:
            render_stream.wait_event(ev_left[slot])                  # frame k - depth has left the slot (sent; on rank 0 stitched)
        out = dev_local if args.rehearse else ex.output(slot)
        t.set_output_device(out.data_ptr())
        t.render(stream)
        ev_drawn[slot].record(render_stream)
        with torch.cuda.stream(side):
            side.wait_event(ev_drawn[slot])
            if collect_ex[0]:
                x0, x1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                x0.record(side)
            if args.rehearse:                                        # the same steps with a hop through host memory (gloo)
                side.synchronize()
                ex.output(slot).copy_(out)
            if cabi_bands:                                           # vf_dist_exchange_bands: all-to-all + band stitch + in-place band gather, no torch
                t.dist_exchange_bands(comm, 0, image.data_ptr() if rank == 0 else 0, side.cuda_stream)
            elif banded:
                def stitch_band(recv, band, rows):                   # [ranks][chunk] tile slots -> the rows of this rank's band (the C-ABI's kernel)
                    src, dst = (dev_gathered.copy_(recv), dev_band) if args.rehearse else (recv, band)
                    t.stitch_tiles(src.data_ptr(), dst.data_ptr(), world, ex.skew, ex.chunk_tiles, side.cuda_stream, height=rows)
                    if args.rehearse:
                        side.synchronize()
                        band.copy_(dev_band)
                ex.exchange(slot, stitch_band, host_image if args.rehearse else image)
                if args.rehearse and rank == 0:
                    image.copy_(host_image)
            else:
                if args.rehearse:
                    ex.start(slot)
                    g = ex.finish(slot)
                    if rank == 0:
                        dev_gathered.copy_(g)
                        g = dev_gathered
                elif comm is not None:                               # RCCL through the C-ABI, queued on the side stream
                    t.dist_gather_tiles(comm, 0, ex.gathered[slot].data_ptr() if rank == 0 else 0, ex.stride, side.cuda_stream)
                    g = ex.gathered[slot]
                else:                                                # torch.distributed: the work waits for the side stream's state, and the side stream for the work
                    ex.start(slot)
                    g = ex.finish(slot)
                if rank == 0:
                    t.stitch_tiles(g.data_ptr(), image.data_ptr(), world, ex.skew, ex.stride, side.cuda_stream)
            if collect_ex[0]:
                x1.record(side)
                ex_events.append((x0, x1))
            ev_left[slot].record(side)
        ex.pending_frame[slot] = True
        if args.serial:
            side.synchronize()

    def flush():
        if ex is not None:
            side.synchronize()                                       # every frame queued so far is gathered and stitched

    if ex is not None:
        ex.pending_frame = [False] * depth

    exchanged = [False]
    balance_info = {}
    # SETTLE untimed frames are set-up, not steps: the frame plan is feedback-driven (a frame's per-tile times decide order and
    # strip splitting two frames later, because frames overlap) and needs a few frames of a new camera to converge; results
    # never depend on it.  The JSON reports `settle_frames` and what a frame without feedback costs (`cold_frame_ms`).

    def timed(camera, steps, warmup):
        if not c5:
            t.set_uniforms(camera_uniforms(camera, W, H))
            for _ in range(SETTLE):
                t.render(stream)
        else:
            for _ in range(SETTLE):                                  # the orbit itself: settle on the poses before the first timed one
                step()
            pose[0] = rank
        if ex is not None and banded and not args.no_balance and not c5:
            # set-up, not steps: the stripes dealt again by what they cost under THIS camera -- and the new deal KEPT only if the slowest
            # rank got faster.  Every rank knows the times of its own tiles; summed per stripe and all-reduced they are the same vector
            # on every rank, and the deterministic rule gives every rank the same table.  Every rank keeps its number of stripes: slabs,
            # chunks and bands keep their sizes.  (Per-tile times are not the whole story -- a rank's frame is also its schedule's tail:
            # emulated, the new deal gains 5-6 % at the top-down camera and 0-2 % at the default one, where it lost 2 % once.)
            def slowest_rank_period(n=12):
                torch.cuda.synchronize()
                c0 = time.perf_counter()
                for _ in range(n):
                    t.render(stream)
                torch.cuda.synchronize()
                x = torch.tensor([(time.perf_counter() - c0) / n * 1e3], dtype=torch.float64, device="cpu" if args.rehearse else dev)
                dist.all_reduce(x, op=dist.ReduceOp.MAX)
                return float(x.item())

            round_robin = vdist.layout_code(0, vdist.layout_stripe_log2(ex.skew))
            if ex.skew != round_robin:                               # (a deal made for another camera: start from the plain one)
                flush()
                ex.set_layout(round_robin); t.set_tile_shard(rank, world, round_robin)
                for _ in range(SETTLE):
                    t.render(stream)
            sl2 = vdist.layout_stripe_log2(ex.skew)
            nstripes = ((W + 63) // 64) >> sl2
            mine = vdist.stripe_times(t.tile_times(), vdist.tile_layout(W, H, rank, world, ex.skew), nstripes, sl2)
            x = torch.from_numpy(mine).to("cpu" if args.rehearse else dev)
            dist.all_reduce(x, op=dist.ReduceOp.SUM)
            stripe_ms = x.cpu().numpy()
            word = vdist.balanced_layout(stripe_ms, world, sl2)
            balance_info.clear()
            balance_info.update({"camera": camera, "stripe_ms": [round(float(v), 4) for v in stripe_ms], "owner": [int(o) for o in cabi.balance_stripes(stripe_ms, world)],
                                 "kept": False})
            if word != round_robin:
                before = slowest_rank_period()
                flush()
                ex.set_layout(word); t.set_tile_shard(rank, world, word)
                for _ in range(SETTLE):
                    t.render(stream)
                after = slowest_rank_period()
                balance_info.update({"slowest_rank_ms_round_robin": before, "slowest_rank_ms_dealt_by_times": after, "kept": after < 0.995 * before})
                if not balance_info["kept"]:                         # (every rank holds the same two numbers: the same decision everywhere)
                    ex.set_layout(round_robin); t.set_tile_shard(rank, world, round_robin)
                    for _ in range(SETTLE):
                        t.render(stream)
        if ex is not None and not exchanged[0]:
            # set-up, not steps: the first exchange creates the point-to-point channels (RCCL opens them lazily, about a second);
            # like communicator creation it must not land in the timed region when the caller asks for --warmup 0
            for _ in range(depth):
                step()
            flush()
            exchanged[0] = True
        # C5 is BASELINE's "batch of 64 camera look-ats": the K timed poses are ONE vf_terrain_render_batch call (a step = one pose of
        # it), the W warm-up poses another; the uniform blocks are made before the clock starts
        batch_w = pose_batch(warmup) if (c5 and warmup) else None
        batch_k = pose_batch(steps) if c5 else None
        if c5:
            if batch_w is not None:
                t.render_batch(batch_w, None, stream)
        else:
            for _ in range(warmup):
                step()
        flush()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        # the plan chain's times (side streams): from four frames of their own in front of the timed region -- event records between the plan's
        # kernels delay the chain, and the timed region carries none
        plan_tm = None
        if not c5:
            t.enable_timing(True, stats=False)
            for _ in range(4):
                step()
            flush()
            torch.cuda.synchronize()
            plan_tm = t.timings()
            t.enable_timing(False)
            if world > 1:
                dist.barrier()
        # HIP events around the kernels of every FOURTH timed frame (every frame when the run is short): the kernels themselves run as
        # untimed, but two event records per frame on the draw stream keep each kernel from being launched under the one before it and
        # cost a C4 frame 2 % (tools/exp_timing_cost.py: 20-frame bursts 0.733 untimed, 0.750 with events on every frame, 0.737 on every fourth)
        t.enable_timing(True, stats=False, sampled=steps >= 8)
        ex_events.clear()
        collect_ex[0] = ex is not None
        t0 = time.perf_counter()
        if c5:
            t.render_batch(batch_k, None, stream)
        else:
            for _ in range(steps):
                step()
        flush()                                                      # every timed frame is gathered and stitched in here
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        collect_ex[0] = False
        tm = t.timings()
        tm["exchange_ms"] = [a.elapsed_time(b) for a, b in ex_events]
        if plan_tm is not None and steps >= 8:
            tm["ranges_ms"], tm["plan_ms"] = plan_tm["ranges_ms"], plan_tm["plan_ms"]
        tm["frame_tile_ms"], tm["frame_period_ms"] = t.frame_times()
        tm["raster_groups"] = t.raster_groups()
        t.enable_timing(False)
        if world > 1:
            x = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(x, op=dist.ReduceOp.MAX)
            dt = float(x.item())
        return dt, tm

    def frame_stats(x):
        """median / p95 / ... of per-frame device times (HIP events), the report of python/tools/perf_sanity.py:45-69"""
        x = np.asarray(x, np.float64)
        if x.size == 0:
            return None
        return {"median": float(np.median(x)), "p95": float(np.percentile(x, 95)), "min": float(x.min()), "max": float(x.max()),
                "stdev": float(x.std(ddof=1)) if x.size > 1 else 0.0, "frames": int(x.size)}

    dt, tm = timed(args.camera, args.steps, args.warmup)
    ms_per_step = dt / args.steps * 1e3
    # per timed frame, this rank's HIP events: frame period (end of the previous frame's tile kernel -> end of this one's;
    # back-to-back frames overlap, so this is what a frame costs) and the kernels on the caller's stream
    frame_ms = {"period": frame_stats(tm["frame_period_ms"][1:]), "tile_kernel": frame_stats(tm["frame_tile_ms"]),
                "clock": "HIP events on the render stream around the kernels of every fourth timed frame (every frame when steps < 8); periods are per frame, "
                         "from events four frames apart (vf_terrain_enable_timing(t, 3), vf_terrain_frame_times)"}
    frames = args.steps * (world if c5 else 1)
    value = W * H * frames / dt / 1e6

    extra = None
    if not args.no_extra and not c5:
        other = "fill" if args.camera == "default" else "default"
        n2 = max(3, args.steps // 4)
        dt2, tm2 = timed(other, n2, 1)
        extra = {"camera": other, "value": W * H * n2 / dt2 / 1e6, "ms_per_step": dt2 / n2 * 1e3, "tile_kernel_ms": tm2["tile_ms"]}

    check = None
    if (args.check or world > 1) and not c5:                         # N > 1: always -- untimed, after the timed region
        t.set_uniforms(camera_uniforms(args.camera, W, H))
        step()
        flush()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if rank == 0:
            got = image.clone()
            single = new_handle()                                    # unsharded handle, same inputs
            single.set_uniforms(camera_uniforms(args.camera, W, H))
            single.set_output_device(image.data_ptr())
            single.render(stream)
            torch.cuda.synchronize()
            check = bool(torch.equal(got, image))
            if not check:                                            # say where: which tiles, how far apart
                d = (got.to(torch.int16) - image.to(torch.int16)).abs().amax(dim=2)
                ys, xs = torch.nonzero(d, as_tuple=True)
                tiles = sorted({(int(x) // 64, int(y) // 64) for x, y in zip(xs[:2000].tolist(), ys[:2000].tolist())})
                print(f"bench.py --check: {int((d > 0).sum())} pixels differ (largest difference {int(d.max())} LSB); tiles (tx, ty): {tiles[:24]}", file=sys.stderr)
            single.close()
        if world > 1:
            dist.barrier()

    # ---- secondary measurements on one GPU: the fragment stage on its own, a frame without scheduling feedback --------------
    frag = cold = None
    if rank == 0 and world == 1 and not args.no_extra:
        probe = new_handle()
        cams = [("pose8", orbit_uniforms(8, W, H))] if c5 else [(c, camera_uniforms(c, W, H)) for c in ("default", "fill")]
        frag = {}
        for name, u in cams:
            probe.set_uniforms(u)
            probe.render(stream)
            torch.cuda.synchronize()
            ft = probe.fragment_stage(repeats=10)
            # SURVEY.md 8(d) B_frag: visibility read + RGBA8 written for every pixel + every height texel at most once;
            # "covered": only what the visible primitives need (12 B per covered pixel + the LUT), the rest is clear colour
            b_frag = 4 * W * H + 4 * W * H + 4 * G * G
            b_cov = 12 * ft["covered_pixels"] + 1024
            s = ft["resolve_ms"] * 1e-3
            frag_pm = None
            try:
                frag_pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(f"{W}x{H}_g{G}_frag_{name}")
                if frag_pm and frag_pm.get("lib_sha256") != lib_sha256(cabi.DEFAULT_LIB):
                    frag_pm = {"stale": True}
            except Exception:  # noqa: BLE001
                frag_pm = None
            meas = frag_pm.get("hbm_bytes_per_launch_fetch_uncorrected") if frag_pm and not frag_pm.get("stale") else None
            meas2 = frag_pm.get("hbm_bytes_per_launch") if frag_pm and not frag_pm.get("stale") else None
            frag[name] = {"traffic": frag_pm,
                          # the same launch priced on the bytes the counters SAW instead of SURVEY's B_frag (which charges every height
                          # texel once whether or not the camera shows it): uncorrected FETCH_SIZE + WRITE_SIZE, and with the guide's x2 on fetch
                          "frac_on_measured_bytes": (meas / s / 1e9 / HBM_PEAK_GBPS) if meas else None,
                          "frac_on_measured_bytes_fetch_x2": (meas2 / s / 1e9 / HBM_PEAK_GBPS) if meas2 else None,
                          "ms": ft["resolve_ms"], "covered_pixels": ft["covered_pixels"], "bytes": b_frag,
                          "GB/s": b_frag / s / 1e9, "frac": b_frag / s / 1e9 / HBM_PEAK_GBPS,
                          "bytes_covered_only": b_cov, "frac_covered_only": b_cov / s / 1e9 / HBM_PEAK_GBPS,
                          "equals_tile_kernel_output": bool(ft["equal_to_frame"])}
        probe.close()
        # first frames of a fresh handle: no feedback yet (unsplit items); includes the height-cache build on the very first
        fresh = new_handle()
        fresh.set_uniforms(cams[0][1])
        fresh.set_output_device(image.data_ptr())
        torch.cuda.synchronize()
        cold = []
        for _ in range(3):
            c0 = time.perf_counter()
            fresh.render(stream)
            torch.cuda.synchronize()
            cold.append((time.perf_counter() - c0) * 1e3)
        fresh.close()

    # ---- the drop-in API end to end (src/terrain/mod.rs:410-491: render -> copy_texture_to_buffer -> map -> PNG): what a caller of the
    #      reference's classes waits for, per BASELINE configuration; never part of `value` -------------------------------------------
    api_latency = None
    if rank == 0 and world == 1 and not args.no_extra and not args.no_api_latency and not c5:
        import tempfile
        api_latency = {"clock": "host wall clock around the call, best of 8 after two warm-up calls", "png_threads": os.environ.get("VF_PNG_THREADS", "default (up to 16)")}
        with tempfile.TemporaryDirectory() as tmpd:
            def best(fn, n=8):
                fn(); fn()
                ts = []
                for _ in range(n):
                    c0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - c0) * 1e3)
                return min(ts)
            spike = vf.TerrainSpike(800, 600, grid=128, colormap="viridis")
            spike.set_camera_look_at((3.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 45.0, 0.1, 100.0)
            api_latency["C2 TerrainSpike 800x600 grid=128"] = {"render_png": best(lambda: spike.render_png(os.path.join(tmpd, "c2.png"))),
                                                               "render_rgba": best(lambda: spike.render_rgba())}
            del spike
            for label, (w, h, g, seed) in (("C3 Scene 1920x1080 grid=1024", (1920, 1080, 1024, 20250815)), ("C4 Scene 4096x4096 grid=4096", (W, H, G, 20250816))):
                sc = vf.Scene(w, h, grid=g, colormap="viridis")
                sc.set_height_from_r32f(height_host if (g, seed) == (G, 20250816) else
                                        np.random.default_rng(seed).random((g, g), dtype=np.float32) * np.float32(0.5) - np.float32(0.25))
                api_latency[label] = {"render_png": best(lambda: sc.render_png(os.path.join(tmpd, "s.png"))), "render_rgba": best(lambda: sc.render_rgba())}
                del sc
        # the reference's actual usage -- construct, set the heights, render ONCE, PNG (src/terrain/mod.rs:259-491) -- in a cold process of
        # its own (tools/one_shot.py): a child, started after everything above is measured; it shares the GPU with nothing (this process idles)
        try:
            torch.cuda.synchronize()
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "one_shot.py"), str(W), str(H), str(G), "20250816"],
                               capture_output=True, text=True, timeout=300)
            api_latency["one_shot"] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-400:]}
        except Exception as e:  # noqa: BLE001
            api_latency["one_shot"] = {"error": repr(e)}

    # ---- roofline of the dominant kernel (k_tile: set-up + raster + fragment, fused) --------------------------------------
    # algorithmic bytes per launch (whole frame): height texture read once + RGBA8 written once + LUT
    kernel_s = tm["tile_ms"] * 1e-3                               # `share` = this rank's share of the frame's pixels
    rank_bytes = int(4 * G * G + 4 * W * H * share + 1024)
    achieved = rank_bytes / kernel_s / 1e9 if kernel_s > 0 else 0.0
    # SQ / TCC counters cannot be collected inside this run (rocprofv3 --pmc passes are separate runs of this very command):
    # they come from profiles/pmc_traffic.json and count only when that file was collected on the library loaded here
    lib_hash = lib_sha256(cabi.DEFAULT_LIB)
    traffic = sq = None
    source = None
    stale = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    key = f"{W}x{H}_g{G}_{'orbit' if c5 else args.camera}_n{world}"
    if os.path.exists(pmc_path):
        try:
            pm = json.load(open(pmc_path)).get(key)
        except Exception:  # noqa: BLE001
            pm = None
        if pm:
            source = f"profiles/pmc_traffic.json@{pm.get('tag', '?')}"
            stale = pm.get("lib_sha256") != lib_hash
            if not stale:
                traffic = pm["hbm_bytes_per_launch"]
                sq = pm.get("sq")
    # bound / achieved / peak / frac are ONE consistent set: algorithmic HBM bytes against the HBM peak, as the contract asks.
    # What actually limits the kernel (vector-ALU issue) is reported beside it under "valu", in its own units.
    roofline = {"bound": "hbm",
                "kernel": "k_clear + k_tile (the kernels on the caller's stream: they produce the frame)",
                "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic, "traffic_stale": stale, "counters_source": source,
                "lib_sha256": lib_hash, "algorithmic_bytes_per_launch": rank_bytes, "kernel_ms": tm["tile_ms"],
                "plan_on_side_stream_elapsed_ms": {"k_block_boxes+k_block_setup": tm["ranges_ms"], "k_plan+k_plan_sort": tm["plan_ms"]},
                "frames_averaged": tm["frames"], "rank_share_of_frame": share,
                "limiter": "valu",
                "note": "the path has no contraction and does not reach the HBM roof: the tile kernel is limited by vector-ALU issue "
                        "(see \"valu\": share of the SIMDs' issue slots in use, lanes active per instruction); frac is the "
                        "algorithmic-bytes figure against the HBM peak"}
    if sq:
        roofline["valu"] = {"issue_frac": sq["valu_busy_frac"], "lane_efficiency": sq["active_lanes_per_valu_inst"] / 64.0,
                            "lane_throughput_frac": sq["valu_busy_frac"] * sq["active_lanes_per_valu_inst"] / 64.0,
                            "wave_insts_per_frame": sq.get("valu_wave_insts"), "unit": "fraction of peak vector issue"}
        if sq.get("setup_valu_wave_insts"):
            # the frame as vector issue time: the tile kernel's and the set-up pass's wave instructions (the pass of the NEXT frame runs in
            # this frame's idle issue slots), four cycles each on a 16-lane SIMD, over the chip's SIMDs at the clock the device reports
            di0 = t.device_info()
            simds, hz = 4 * int(di0["compute_units"]), 1e3 * float(di0["clock_khz"])
            insts = float(sq["valu_wave_insts"]) + float(sq["setup_valu_wave_insts"])
            roofline["valu"].update({"setup_pass_wave_insts_per_frame": sq["setup_valu_wave_insts"],
                                     "frame_vector_issue_ms": insts * 4.0 / (simds * hz) * 1e3, "simds": simds, "clock_ghz": hz / 1e9})

    # ---- CPU baseline: the oracle (a port, not the reference: it cannot be built here) on this box's host cores -----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        ncpu, aff, quota, model = host_cpus()
        threads = max(1, min(aff, int(quota) if quota and quota >= 1 else aff, oracle.max_threads()))
        u = orbit_uniforms(8, W, H) if c5 else camera_uniforms(args.camera, W, H)
        c0 = time.perf_counter()
        oracle.render_terrain(u, W, H, G, height_host, lut, nthreads=threads, want_vis=False)
        cdt = time.perf_counter() - c0
        # one thread: a bounded sample -- the rows of every 8th 64-row band (1/8 of the frame's pixels; every primitive is
        # still transformed and set up, only rows outside the sample are skipped)
        nb = 8 if H >= 8 * 64 else 1
        rows = sum(min(64, H - b * 64) for b in range((H + 63) // 64) if b % nb == 0)
        c0 = time.perf_counter()
        oracle.render_terrain(u, W, H, G, height_host, lut, rank=0, nranks=nb, band_h=64, nthreads=1, want_vis=False)
        sdt = time.perf_counter() - c0
        cpu = {"value": W * H / cdt / 1e6, "unit": "Mpix/s", "cores": threads, "kind": "port",
               "single_thread": {"value": W * rows / sdt / 1e6, "unit": "Mpix/s", "cores": 1,
                                 "sample": f"rows of every {nb}th 64-row band ({rows} of {H} rows; all primitives set up) in {sdt:.2f} s"},
               "host": {"nproc": ncpu, "affinity": aff, "cgroup_cpu_quota": quota, "cpu_model": model},
               "sample": f"1 full frame of the same workload ({W}x{H}, grid {G}, {'pose 8' if c5 else args.camera + ' camera'}) in {cdt:.2f} s on "
                         f"{threads} threads (every CPU this process may use), oracle/vf_oracle.c gcc -O2 OpenMP; the reference's wgpu "
                         f"software-adapter path cannot be built in this image"}

    # ---- who ran: one entry per rank, so that an N > 1 line proves its own rank count and devices --------------------------------
    # ... and what it spent: a first run on hardware that scales badly must say whether compute, imbalance or the exchange did it
    ranks_info = rccl = scaling_diag = None
    if world > 1:
        di = t.device_info()
        pairs = None
        if not c5:                                                   # one more frame, untimed, with the per-item statistics on: the (tile, block) pairs this rank draws
            flush()
            t.set_uniforms(camera_uniforms(args.camera, W, H))
            t.enable_timing(True, stats=True)
            t.render(stream)
            torch.cuda.synchronize()
            pairs = int(t.timings()["blocks_rasterised"])
            t.enable_timing(False)
        period = np.asarray(tm["frame_period_ms"][1:], np.float64)
        xs = np.asarray(tm.get("exchange_ms") or [], np.float64)
        mine = {"rank": rank, "hip_device": local_rank, "pci_bus_id": f"{di['pci_bus_id']:02x}:{di['pci_device_id']:02x}", "name": di["name"],
                "local_tiles": t.local_tiles() if not c5 else None, "pid": os.getpid(),
                # HIP events of this rank's handle over the timed frames: its kernels on the render stream, and the period between two frames' ends
                "tile_kernel_ms": float(tm["tile_ms"]), "frame_period_ms": float(np.median(period)) if period.size else None,
                # events around the frame's exchange on the side stream (all-to-all + band stitch + band gather, or gather + root stitch)
                "exchange_ms": float(np.median(xs)) if xs.size else None,
                "local_pairs": pairs}
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, mine)
        per = [r["frame_period_ms"] for r in ranks_info if r.get("frame_period_ms")]
        exs = [r["exchange_ms"] for r in ranks_info if r.get("exchange_ms") is not None]
        if per:
            slowest = max(ranks_info, key=lambda r: r.get("frame_period_ms") or 0.0)
            scaling_diag = {"slowest_rank": slowest["rank"], "slowest_rank_frame_period_ms": max(per), "mean_frame_period_ms": float(np.mean(per)),
                            "imbalance": max(per) / float(np.mean(per)),
                            # frame k's exchange runs on the side stream while frame k + 1 is drawn: hidden when it is shorter than the frame
                            # period of the slowest rank AND the step is no longer than that period (plus a tenth: host jitter)
                            "exchange_ms_max": max(exs) if exs else None,
                            "exchange_hidden": bool(exs and max(exs) <= max(per) and ms_per_step <= 1.1 * max(per)),
                            "step_minus_slowest_period_ms": ms_per_step - max(per)}
        rccl = {"backend": dist.get_backend(), "rccl_world_size": dist.get_world_size(),
                "version": ".".join(str(v) for v in torch.cuda.nccl.version()) if not args.rehearse else None,
                "distinct_devices": len({(r["hip_device"], r["pci_bus_id"]) for r in ranks_info})}
        if comm is not None:
            rccl["version_c_abi"] = t.dist_version()

    if rank == 0:
        if c5:
            metric = f"Mpix/s terrain shade (grid={G}, {W}x{H}, 64-pose orbit batch)"
            workload = (f"C5: 64 camera look-ats on the default camera's orbit over one grid={G} terrain, Scene {W}x{H}, R32F {G}x{G} "
                        f"heightmap rng(20250817)*0.5-0.25, viridis; rank r renders poses k = r mod N through ONE vf_terrain_render_batch call on one handle")
            par = "1 GPU, poses in order" if world == 1 else f"pose-parallel replicas over {world} GPUs, no collective"
        else:
            metric = "Mpix/s terrain shade (grid=4096, 4096x4096)" if (W, G) == (4096, 4096) else f"Mpix/s terrain shade (grid={G}, {W}x{H})"
            workload = f"C4: Scene {W}x{H}, grid={G}, R32F {G}x{G} heightmap rng(20250816)*0.5-0.25, {args.camera} camera, viridis"
            par = ("1 GPU, whole frame" if world == 1 else
                   f"64x64 screen tiles in column stripes of {1 << vdist.layout_stripe_log2(ex.skew)} tile(s) over {world} GPUs (" +
                   ("stripes dealt by their measured times, the same number to every rank: vf_balance_stripes" if ex.skew & (1 << 20) else
                    f"owner = ((tx >> {vdist.layout_stripe_log2(ex.skew)}) + {ex.skew & 0xFFFF}*ty) % {world}") + "), " +
                   ("all-to-all (RCCL through the C-ABI, vf_dist_exchange_bands) + one band stitched per rank + bands gathered in place on rank 0, " if cabi_bands else
                    "all-to-all (RCCL via torch.distributed) + one band stitched per rank + bands gathered in place on rank 0, " if banded else
                    f"p2p gather to rank 0 ({'RCCL through the C-ABI, vf_dist_gather_tiles' if comm is not None else 'RCCL via torch.distributed'}) + stitch, ") +
                   f"{'serial' if args.serial else 'double-buffered'}")
        out = {
            "metric": metric, "value": value, "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "replicas" if c5 else "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "width": W, "height": H, "grid": G, "camera": "orbit" if c5 else args.camera, "parallelism": par},
            "settle_frames": SETTLE,
            "frame_ms": frame_ms,
            "shade_precision": "fast (hardware rcp/rsq/sin/cos/log/exp, within 1 LSB of the exact path; visibility identical)",
            # which of the raster stage's two line loops drew the timed frames (the handle times both on the settle frames and keeps the
            # faster per view: vf_terrain_set_raster_groups) and what its probes measured: the frame period (end of one frame's work on the
            # draw stream to the end of the next, both drawn by the same variant), per variant
            "raster_line_loop": {"with_line_groups": bool(tm["raster_groups"][0]), "frame_period_ms_probed": {"plain": tm["raster_groups"][1][0], "groups": tm["raster_groups"][1][1]}},
            "build": toolchain(),
            "roofline": roofline,
            "roofline_fragment": frag,
            "cpu_baseline": cpu,
            "other_camera": extra,
        }
        if c5:
            out["ms_per_pose"] = dt / args.steps * 1e3               # per rank: one pose per step
        if cold:
            out["cold_frame_ms"] = cold[0]
            out["cold_frames_ms"] = {"first (planned from the static estimate: no feedback yet)": cold[0], "second (first frame's feedback)": cold[1], "third": cold[2]}
        if check is not None:
            out["gathered_frame_equals_single_rank_frame"] = check
        if balance_info:
            out["stripe_balance"] = balance_info
        if ranks_info is not None:
            out["ranks"] = ranks_info
            out["rccl"] = rccl
        if scaling_diag is not None:
            out.update(scaling_diag)
        if api_latency is not None:
            out["api_latency_ms"] = api_latency
        if args.rehearse:
            out["rehearsal"] = "gloo via host memory on one GPU; not a performance number"
        print(json.dumps(out), flush=True)
    # Teardown.  torch's current stream is the handle's own (ExternalStream above) and t.close() destroys it: give torch its default
    # stream back and drain first, and take the process group down BEFORE the handle -- ProcessGroupNCCL records events on the
    # current stream in barrier() / destroy_process_group().
    torch.cuda.synchronize()
    torch.cuda.set_stream(torch.cuda.default_stream(dev))
    failed = check is False
    if world > 1:
        flag = torch.tensor([1 if failed else 0], dtype=torch.int32, device="cpu" if args.rehearse else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)                  # every rank leaves with the same code
        failed = bool(flag.item())
        dist.barrier()
    if comm is not None:
        t.dist_comm_destroy(comm)
    if world > 1:
        dist.destroy_process_group()
    t.close()
    if failed:
        if rank == 0:
            print("bench.py: the gathered frame differs from the single-rank frame -- the line above is not a valid measurement", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
