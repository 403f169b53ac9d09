#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config.

  metric   Mpix/s terrain shade: W*H*frames / wall time, RGBA8 complete in HBM (rank 0 holds the gathered frame)
  workload C4 of SURVEY.md 8(d): Scene 4096x4096, grid=4096, 4096x4096 R32F heightmap
           (np.random.default_rng(20250816).random(float32)*0.5-0.25), default camera eye (3,2,3), viridis.
           A "step" = one frame: k_block_boxes + k_plan + k_plan_sort (on the handle's side stream: they only touch plan state and
           overlap the previous frame's tile kernel) + k_clear + k_tile (+ gather to rank 0 + stitch when N > 1).
  N > 1    one process per GPU (torch.distributed, backend nccl = RCCL): screen-tile split -- 64x64 tile (tx, ty)
           belongs to rank (tx + skew*ty) % N; every rank renders only its tiles into a tile-major slab, the slabs go to
           rank 0 point-to-point over xGMI (vulkan_forge_amd/dist.py::TileExchange) and vf_stitch_tiles_device writes the
           frame.  The exchange is double-buffered: frame k travels while frame k+1 renders; every one of the K timed
           frames is rendered, gathered and stitched inside the timed region (--serial: no overlap).  Total work is fixed,
           so scaling is "strong".

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (fields: see the driver contract; plus "roofline" and "cpu_baseline").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=4096, help="frame width = height")
    ap.add_argument("--grid", type=int, default=4096)
    ap.add_argument("--camera", choices=["default", "fill"], default="default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary (fill camera) measurement")
    ap.add_argument("--check", action="store_true", help="after timing, compare the gathered frame with a single-rank render")
    ap.add_argument("--serial", action="store_true", help="N>1: finish each frame's exchange before rendering the next (no overlap)")
    ap.add_argument("--rehearse", action="store_true",
                    help="N>1 dress rehearsal on ONE GPU: every rank uses device 0 and the exchange runs over gloo through host "
                         "memory (RCCL refuses two ranks on one device); exercises sharding, exchange and reporting, not xGMI")
    return ap.parse_args()


def camera_uniforms(name, W, H):
    """Uniform block through the product's own host code (the drop-in module), not the oracle."""
    import vulkan_forge_amd as vf
    import numpy as np
    view = vf.camera_look_at((3.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))
    proj = vf.camera_perspective(45.0, W / H, 0.1, 100.0, "wgpu")
    if name == "fill":                                           # frame-filling top-down camera
        view = vf.camera_look_at((0.0, 2.2, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, -1.0))
        proj = vf.camera_perspective(60.0, W / H, 0.1, 100.0, "wgpu")
    u = np.zeros(44, np.float32)
    u[:16] = view.T.reshape(-1)                                   # column-major
    u[16:32] = proj.T.reshape(-1)
    sun = np.array([0.5, 0.8, 0.6], np.float32)                  # Scene keeps Globals::default (src/terrain/mod.rs:190)
    u[32:35] = sun * (np.float32(1.0) / np.sqrt(np.float32((sun * sun).sum())))
    u[35] = 1.0
    u[36:39] = [1.0, 1.0, 1.0]
    return u


def main():
    args = parse()
    import numpy as np
    import torch                                                  # first: the HIP runtime it bundles is the one shared below
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
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

    from vulkan_forge_amd import cabi, dist as vdist

    W = H = args.size
    G = args.grid
    lut = np.load(os.path.join(ROOT, "tests", "golden", "colormaps_rgba8.npz"))["viridis"]
    rng = np.random.default_rng(20250816)
    height_host = rng.random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
    d_height = torch.from_numpy(height_host).to(dev)              # inputs resident in HBM before the timed region

    t = cabi.Terrain(W, H, G, lut, lut_is_srgb=True, device=local_rank)
    t.set_height_device(d_height.data_ptr(), G, G)
    stream = torch.cuda.current_stream().cuda_stream
    image = torch.empty((H, W, 4), dtype=torch.uint8, device=dev) if rank == 0 else None
    depth = 1 if args.serial else 2
    if world == 1:
        t.set_output_device(image.data_ptr())                    # N=1: render straight into the frame
        share = 1.0
        ex = None
    else:
        # rehearsal: same exchange code over gloo through host memory (the device slabs are copied out and back in)
        ex = vdist.TileExchange(W, H, "cpu" if args.rehearse else dev, depth=depth)
        t.set_tile_shard(rank, world, ex.skew)
        assert t.local_tiles() == len(vdist.tile_layout(W, H, rank, world, ex.skew))
        share = t.local_tiles() / float(((W + 63) // 64) * ((H + 63) // 64))
        if args.rehearse:
            dev_local = torch.zeros(ex.stride * vdist.TILE_WORDS, dtype=torch.int32, device=dev)
            dev_gathered = torch.zeros((world, ex.stride * vdist.TILE_WORDS), dtype=torch.int32, device=dev) if rank == 0 else None
    frame_no = [0]

    def finish(slot):
        """complete the exchange that last used `slot`; rank 0 writes that frame"""
        g = ex.finish(slot)
        if rank == 0 and g is not None and ex.pending_frame[slot]:
            if args.rehearse:
                dev_gathered.copy_(g)
                g = dev_gathered
            t.stitch_tiles(g.data_ptr(), image.data_ptr(), world, ex.skew, ex.stride, stream)
        if ex is not None:
            ex.pending_frame[slot] = False

    def step():
        if world == 1:
            t.render(stream)
            return
        slot = frame_no[0] % depth
        frame_no[0] += 1
        finish(slot)                                                 # frame k - depth: its slab buffers are about to be reused
        out = dev_local if args.rehearse else ex.output(slot)
        t.set_output_device(out.data_ptr())
        t.render(stream)
        if args.rehearse:
            torch.cuda.synchronize()
            ex.output(slot).copy_(out)
        ex.start(slot)
        ex.pending_frame[slot] = True
        if args.serial:
            finish(slot)

    def flush():
        if world > 1:
            for k in range(depth):
                finish((frame_no[0] + k) % depth)                    # oldest first

    if ex is not None:
        ex.pending_frame = [False] * depth

    exchanged = [False]
    SETTLE = 16  # set-up, not steps: the frame plan is feedback-driven (last frame's per-tile cost decides order and strip
                 # splitting; two frames old, because frames overlap) and needs a few frames of a new camera to converge; results never depend on it

    def timed(camera, steps, warmup):
        t.set_uniforms(camera_uniforms(camera, W, H))
        for _ in range(SETTLE):
            t.render(stream)
        if world > 1 and not exchanged[0]:
            # set-up, not steps: the first exchange creates the point-to-point channels (RCCL opens them lazily, about a second);
            # like communicator creation it must not land in the timed region when the caller asks for --warmup 0
            for _ in range(depth):
                step()
            flush()
            exchanged[0] = True
        for _ in range(warmup):
            step()
        flush()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t.enable_timing(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        flush()                                                      # every timed frame is gathered and stitched in here
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tm = t.timings()
        t.enable_timing(False)
        if world > 1:
            x = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(x, op=dist.ReduceOp.MAX)
            dt = float(x.item())
        return dt, tm

    dt, tm = timed(args.camera, args.steps, args.warmup)
    ms_per_step = dt / args.steps * 1e3
    value = W * H * args.steps / dt / 1e6

    extra = None
    if not args.no_extra:
        other = "fill" if args.camera == "default" else "default"
        dt2, tm2 = timed(other, max(3, args.steps // 4), 1)
        n2 = max(3, args.steps // 4)
        extra = {"camera": other, "value": W * H * n2 / dt2 / 1e6, "ms_per_step": dt2 / n2 * 1e3, "tile_kernel_ms": tm2["tile_ms"]}

    check = None
    if args.check:
        t.set_uniforms(camera_uniforms(args.camera, W, H))
        step()
        flush()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if rank == 0:
            got = image.clone()
            single = cabi.Terrain(W, H, G, lut, lut_is_srgb=True, device=local_rank)   # unsharded handle, same inputs
            single.set_height_device(d_height.data_ptr(), G, G)
            single.set_uniforms(camera_uniforms(args.camera, W, H))
            single.set_output_device(image.data_ptr())
            single.render(stream)
            torch.cuda.synchronize()
            check = bool(torch.equal(got, image))
            single.close()
        if world > 1:
            dist.barrier()

    # ---- roofline of the dominant kernel (k_tile: vertex + setup + raster + fragment, fused) -------------------
    # algorithmic bytes per launch (whole frame): height texture read once + RGBA8 written once + LUT
    algo_bytes = 4 * G * G + 4 * W * H + 1024
    kernel_s = tm["tile_ms"] * 1e-3                               # `share` = this rank's share of the frame's pixels
    rank_bytes = int(4 * G * G + 4 * W * H * share + 1024)
    achieved = rank_bytes / kernel_s / 1e9 if kernel_s > 0 else 0.0
    traffic, sq = None, None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            pm = json.load(open(pmc_path))
            key = f"{W}x{H}_g{G}_{args.camera}_n{world}"
            if key in pm:
                traffic = pm[key]["hbm_bytes_per_launch"]
                sq = pm[key].get("sq")
        except Exception:  # noqa: BLE001
            traffic = None
    roofline = {"bound": "hbm", "kernel": "k_clear + k_tile (the kernels on the caller's stream: they produce the frame)", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "algorithmic_bytes_per_launch": rank_bytes, "kernel_ms": tm["tile_ms"],
                "plan_on_side_stream_elapsed_ms": {"k_block_boxes": tm["ranges_ms"], "k_plan+k_plan_sort": tm["plan_ms"]}, "frames_averaged": tm["frames"], "rank_share_of_frame": share}
    if sq:   # the path has no contraction and is not HBM-bound: what it IS bound by, from the committed SQ counter passes
        roofline["valu_busy_frac"] = sq["valu_busy_frac"]
        roofline["active_lanes_per_valu_inst"] = sq["active_lanes_per_valu_inst"]

    # ---- CPU baseline: the oracle (a port, not the reference: it cannot be built here) on this box's host cores -----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        threads = max(1, min(os.cpu_count() or 1, 16))
        u = camera_uniforms(args.camera, W, H)
        c0 = time.perf_counter()
        oracle.render_terrain(u, W, H, G, height_host, lut, nthreads=threads, want_vis=False)
        cdt = time.perf_counter() - c0
        cpu = {"value": W * H / cdt / 1e6, "unit": "Mpix/s", "cores": threads, "kind": "port",
               "sample": f"1 full frame of the same workload ({W}x{H}, grid {G}, {args.camera} camera) in {cdt:.2f} s, "
                         f"oracle/vf_oracle.c gcc -O2 OpenMP; the reference's wgpu software-adapter path cannot be built in this image"}

    if rank == 0:
        out = {
            "metric": "Mpix/s terrain shade (grid=4096, 4096x4096)" if (W, G) == (4096, 4096) else f"Mpix/s terrain shade (grid={G}, {W}x{H})",
            "value": value, "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C4: Scene {W}x{H}, grid={G}, R32F {G}x{G} heightmap rng(20250816)*0.5-0.25, {args.camera} camera, viridis",
                       "width": W, "height": H, "grid": G, "camera": args.camera,
                       "parallelism": "1 GPU, whole frame" if world == 1 else
                                      f"64x64 screen tiles interleaved over {world} GPUs (owner = (tx + {ex.skew}*ty) % {world}), p2p gather to rank 0 (RCCL) "
                                      f"+ stitch, {'serial' if args.serial else 'double-buffered'}"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "other_camera": extra,
        }
        if check is not None:
            out["gathered_frame_equals_single_rank_frame"] = check
        if args.rehearse:
            out["rehearsal"] = "gloo via host memory on one GPU; not a performance number"
        print(json.dumps(out), flush=True)
    t.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
