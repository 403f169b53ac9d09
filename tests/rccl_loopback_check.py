"""Run by tests/test_gpu_rccl_loopback.py in a fresh interpreter (torch must load its HIP runtime before the library does)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    luts = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "colormaps_rgba8.npz"))
    import torch                                           # before the library: one HIP runtime per process
    import torch.distributed as dist
    from vulkan_forge_amd import cabi, dist as vdist
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    if not torch.cuda.is_available():
        print("NO GPU"); return 2
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if dist.is_initialized():
        return 3
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1, device_id=dev)
    try:
        x = torch.full((4,), 3.0, device=dev)
        dist.all_reduce(x); dist.barrier()
        assert float(x.sum()) == 12.0
        W, H, G = 1000, 700, 192                           # edges cut tiles
        rng = np.random.default_rng(11)
        h = (rng.random((G, G), dtype=np.float32) - np.float32(0.5)) * np.float32(0.5)
        stream = torch.cuda.current_stream().cuda_stream
        whole = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
        t = cabi.Terrain(W, H, G, luts["viridis"], device=0)
        t.set_height(h); t.set_uniforms(b.camera_uniforms("fill", W, H))
        t.set_output_device(whole.data_ptr()); t.render(stream)
        # the same frame as a "1-rank tile shard" travelling through RCCL to itself, twice (slot reuse)
        skew = 1
        stride = vdist.stride_tiles(W, H, 1, skew)
        words = stride * vdist.TILE_WORDS
        slab = torch.zeros(words, dtype=torch.int32, device=dev)
        gathered = torch.zeros((1, words), dtype=torch.int32, device=dev)
        image = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
        t.set_tile_shard(0, 1, skew)
        for _ in range(2):
            t.set_output_device(slab.data_ptr()); t.render(stream)
            works = dist.batch_isend_irecv([dist.P2POp(dist.isend, slab, 0), dist.P2POp(dist.irecv, gathered[0], 0)])
            for w in works:
                w.wait()
            t.stitch_tiles(gathered.data_ptr(), image.data_ptr(), 1, skew, stride, stream)
        torch.cuda.synchronize()
        assert torch.equal(image, whole)
        t.close()
    finally:
        dist.destroy_process_group()
    print("LOOPBACK OK")
    return 0


if __name__ == "__main__":
    sys.exit(main())
