"""Consecutive frames on alternating streams and output buffers (their tile kernels may overlap: frame f + 1's persistent workgroups take
the CUs frame f's tail leaves idle) against the same frames on one stream: steady-state frame period at C4, one GPU and one rank of N.
usage: exp_overlap.py [camera] [rank n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import ctypes as C
import numpy as np
import vulkan_forge_amd as vf
from vulkan_forge_amd import cabi
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
W = H = G = 4096
if os.environ.get("VF_C5"): W, H, G = 1920, 1080, 2048            # BASELINE config 5's frame; camera "orbit" = the 64 poses in turn (a moving camera)
cam = sys.argv[1] if len(sys.argv) > 1 else "default"
shard = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else None
h = np.random.default_rng(20250817 if os.environ.get("VF_C5") else 20250816).random((G, G), dtype=np.float32) * np.float32(0.5) - np.float32(0.25)
t = cabi.Terrain(W, H, G, vf.colormap_rgba8("viridis")); t.set_height(h); t.set_uniforms(b.orbit_uniforms(0, W, H) if cam == "orbit" else b.camera_uniforms(cam, W, H))
orbit = [b.orbit_uniforms(k, W, H) for k in range(64)] if cam == "orbit" else None
if shard: t.set_tile_shard(shard[0], shard[1], 0)
hip = C.CDLL("libamdhip64.so.7")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
def dmalloc(n):
    p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), n) == 0; return p.value
def stream():
    s = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0; return s.value
outs = [dmalloc(W * H * 4) for _ in range(3)]
streams = [t.stream_handle(), stream(), stream()]
def period(nstreams, nbufs, n=192):
    for f in range(40):
        if orbit: t.set_uniforms(orbit[f % 64])
        t.set_output_device(outs[f % nbufs]); t.render(streams[f % nstreams])
    for s in streams: hip.hipStreamSynchronize(s)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for f in range(n):
            if orbit: t.set_uniforms(orbit[f % 64])
            t.set_output_device(outs[f % nbufs]); t.render(streams[f % nstreams])
        for s in streams: hip.hipStreamSynchronize(s)
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best
label = f"{cam} " + (f"rank {shard[0]}/{shard[1]}" if shard else "one GPU")
for ns, nb in ((1, 1), (1, 2), (2, 2), (3, 3), (2, 2), (1, 1)):
    print(f"{label}: {ns} stream(s), {nb} output buffer(s): {period(ns, nb):.4f} ms per frame", flush=True)
