#!/usr/bin/env python3
"""Determinism harness (protocol of the reference's python/tools/determinism_harness.py:1-10,84-112): render N times,
hash the raw RGBA bytes, require one unique hash; optionally in several processes at once.  Workloads: `triangle` (the
reference's) and `terrain` (TerrainSpike frame -- the asynchronous tile kernel paints with atomic max, so repeat-run
identity is a property worth checking, not a given)."""
from __future__ import annotations

import argparse
import hashlib
import os
import time

from ._stats import write_json


def render_bytes(width, height, kind="triangle", grid=64):
    import vulkan_forge_amd as vf
    if kind == "triangle":
        return vf.Renderer(width, height).render_triangle_rgba().tobytes()
    return vf.TerrainSpike(width, height, grid).render_rgba().tobytes()


def _child(width, height, kind, grid, q):
    q.put(hashlib.sha256(render_bytes(width, height, kind, grid)).hexdigest())


def run(width, height, runs, processes, kind, grid):
    """[(sha256, ms)] per run; a multi-process run reports the children's common hash."""
    out = []
    if processes <= 0:
        for _ in range(runs):
            t0 = time.perf_counter()
            digest = hashlib.sha256(render_bytes(width, height, kind, grid)).hexdigest()
            out.append((digest, (time.perf_counter() - t0) * 1e3))
        return out
    import multiprocessing as mp
    ctx = mp.get_context("spawn")                  # children must initialise the GPU themselves: never fork a HIP process
    for _ in range(runs):
        q = ctx.Queue()
        t0 = time.perf_counter()
        procs = [ctx.Process(target=_child, args=(width, height, kind, grid, q)) for _ in range(processes)]
        for p in procs:
            p.start()
        digests = [q.get(timeout=300) for _ in procs]
        for p in procs:
            p.join()
        if len(set(digests)) != 1:
            raise AssertionError(f"Non-deterministic across processes: {digests}")
        out.append((digests[0], (time.perf_counter() - t0) * 1e3))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--width", type=int, default=128)
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--runs", type=int, default=5)
    ap.add_argument("--processes", type=int, default=0, help=">0 enables the multi-process check (at most 6 share one GPU)")
    ap.add_argument("--png", action="store_true", help="also write a PNG to --out-dir")
    ap.add_argument("--out-dir", default="determinism_artifacts")
    ap.add_argument("--workload", choices=["triangle", "terrain"], default="triangle")
    ap.add_argument("--grid", type=int, default=64)
    a = ap.parse_args(argv)
    os.makedirs(a.out_dir, exist_ok=True)
    results = run(a.width, a.height, a.runs, min(a.processes, 6), a.workload, a.grid)
    hashes = [h for h, _ in results]
    report = {"width": a.width, "height": a.height, "runs": a.runs, "processes": a.processes, "workload": a.workload,
              "hashes": hashes, "unique": sorted(set(hashes)), "all_equal": len(set(hashes)) == 1,
              "avg_ms": sum(ms for _, ms in results) / max(1, len(results))}
    if a.png:
        import vulkan_forge_amd as vf
        name = "triangle.png" if a.workload == "triangle" else "terrain.png"
        try:
            if a.workload == "triangle":
                vf.Renderer(a.width, a.height).render_triangle_png(os.path.join(a.out_dir, name))
            else:
                vf.TerrainSpike(a.width, a.height, a.grid).render_png(os.path.join(a.out_dir, name))
            report["png"] = name
        except Exception as e:  # noqa: BLE001 - the report records it, as the reference's does
            report["png_error"] = str(e)
    write_json(os.path.join(a.out_dir, "determinism_report.json"), report)
    if not report["all_equal"]:
        raise SystemExit("Determinism check FAILED: differing hashes")
    print("Determinism check OK")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
