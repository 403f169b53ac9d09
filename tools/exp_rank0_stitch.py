"""Rank 0 of an N-rank tile shard on this one GPU: its steady-state frame period when every frame is also stitched from a full gather
buffer (the other ranks' slabs are whatever the buffer holds: the copy costs the same) -- on the render stream, as round 2's bench did,
or on a side stream ordered by events, as bench.py does now -- next to the period of a rank that does not stitch.
usage: exp_rank0_stitch.py [N] [camera]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("VF_HWQ", "8"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vulkan_forge_amd as vf
from vulkan_forge_amd import cabi, dist as vdist
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cam = sys.argv[2] if len(sys.argv) > 2 else "default"
W = H = G = 4096
dev = torch.device("cuda", 0)
h = np.random.default_rng(20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
d_h = torch.from_numpy(h).to(dev)
t = cabi.Terrain(W, H, G, vf.colormap_rgba8("viridis")); t.set_height_device(d_h.data_ptr(), G, G); t.set_uniforms(b.camera_uniforms(cam, W, H))
t.set_tile_shard(0, N, 0)
stride = vdist.stride_tiles(W, H, N, 0)
depth = 2
gathered = [torch.zeros((N, stride * vdist.TILE_WORDS), dtype=torch.int32, device=dev) for _ in range(depth)]
image = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
render = torch.cuda.ExternalStream(t.stream_handle(), device=dev); torch.cuda.set_stream(render); side = torch.cuda.Stream(dev)     # (the library's own stream as torch's current one)
drawn = [torch.cuda.Event() for _ in range(depth)]; left = [torch.cuda.Event() for _ in range(depth)]

band_rows = H // N
chunk = (H // 64 // N) * (W // 64 // N)
band = torch.empty((band_rows, W, 4), dtype=torch.uint8, device=dev)

def run(mode, frames):
    for k in range(frames):
        s = k % depth
        if mode in ("side", "band") and k >= depth: render.wait_event(left[s])
        t.set_output_device(gathered[s][0].data_ptr())
        t.render(render.cuda_stream)
        if mode == "same":
            t.stitch_tiles(gathered[s].data_ptr(), image.data_ptr(), N, 0, stride, render.cuda_stream)
        elif mode in ("side", "band"):
            drawn[s].record(render)
            with torch.cuda.stream(side):
                side.wait_event(drawn[s])
                if mode == "side": t.stitch_tiles(gathered[s].data_ptr(), image.data_ptr(), N, 0, stride, side.cuda_stream)
                else: t.stitch_tiles(gathered[s].data_ptr(), band.data_ptr(), N, 0, chunk, side.cuda_stream, height=band_rows)   # what EVERY rank does in dist.BandStitchExchange
                left[s].record(side)
    torch.cuda.synchronize()

for mode in ("none", "same", "side", "band", "none", "same", "side", "band"):
    run(mode, 40)
    t0 = time.perf_counter(); run(mode, 200); dt = (time.perf_counter() - t0) / 200 * 1e3
    print(f"rank 0 of {N}, {cam}: stitch {mode:5s}: frame period {dt:.4f} ms", flush=True)
