#!/usr/bin/env python3
"""Performance sanity harness (protocol of the reference's python/tools/perf_sanity.py:1-20,45-69).

  init_ms    object construction + first (cold) render
  steady     per-iteration wall times after `--warmups` unrecorded iterations: mean, median, p95, stdev, min, max

Workloads: `triangle` (the reference's: Renderer.render_triangle_rgba), `terrain` (TerrainSpike.render_rgba) and
`scene` (Scene with a seeded random R32F heightmap -- BASELINE.json's configs at --width/--height/--grid).
Never fails unless VF_ENFORCE_PERF=1: then steady p95 must stay within --regress-pct of --baseline's p95, or, without a
baseline, under 40 ms x (W*H / 512^2) x --budget-mult.
"""
from __future__ import annotations

import argparse
import csv
import json
import os
import time

from ._stats import summary, write_json


def _workload(kind, width, height, grid):
    import numpy as np
    import vulkan_forge_amd as vf
    if kind == "triangle":
        r = vf.Renderer(width, height)
        return r.render_triangle_rgba
    if kind == "terrain":
        t = vf.TerrainSpike(width, height, grid)
        return t.render_rgba
    s = vf.Scene(width, height, grid)
    s.set_height_from_r32f(np.random.default_rng(20250816).random((grid, grid), dtype=np.float32) * np.float32(0.5) - np.float32(0.25))
    return s.render_rgba


def measure(width, height, runs, warmups, kind="triangle", grid=128):
    t0 = time.perf_counter()
    render = _workload(kind, width, height, grid)
    render()
    init_ms = (time.perf_counter() - t0) * 1e3
    for _ in range(max(0, warmups)):
        render()
    samples = []
    for _ in range(runs):
        t = time.perf_counter()
        render()
        samples.append((time.perf_counter() - t) * 1e3)
    rep = {"width": width, "height": height, "runs": runs, "warmups": warmups, "workload": kind, "init_ms": init_ms, "steady": summary(samples)}
    if kind != "triangle":
        rep["grid"] = grid
        med = rep["steady"]["median_ms"]
        rep["mpix_per_s_incl_readback"] = width * height / med / 1e3 if med > 0 else float("nan")
    return rep


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--width", type=int, default=128)
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--runs", type=int, default=30)
    ap.add_argument("--warmups", type=int, default=3)
    ap.add_argument("--json", default="perf_report.json")
    ap.add_argument("--csv", default="")
    ap.add_argument("--baseline", default="")
    ap.add_argument("--regress-pct", type=float, default=50.0)
    ap.add_argument("--budget-mult", type=float, default=3.0)
    ap.add_argument("--workload", choices=["triangle", "terrain", "scene"], default="triangle")
    ap.add_argument("--grid", type=int, default=128)
    a = ap.parse_args(argv)

    rep = measure(a.width, a.height, a.runs, a.warmups, a.workload, a.grid)
    if a.csv:
        os.makedirs(os.path.dirname(a.csv) or ".", exist_ok=True)
        with open(a.csv, "w", newline="", encoding="utf-8") as f:
            w = csv.writer(f)
            w.writerow(["iter", "ms"])
            w.writerows([i, f"{ms:.3f}"] for i, ms in enumerate(rep["steady"]["samples_ms"]))
    write_json(a.json, rep)

    if os.environ.get("VF_ENFORCE_PERF", "").strip() == "1":
        p95 = float(rep["steady"]["p95_ms"])
        limit, why = None, ""
        if a.baseline:
            try:
                with open(a.baseline, encoding="utf-8") as f:
                    base = float(json.load(f)["steady"]["p95_ms"])
                limit, why = base * (1.0 + a.regress_pct / 100.0), f"baseline p95 {base:.3f} ms + {a.regress_pct:.1f}%"
            except (OSError, KeyError, ValueError) as e:
                print(f"WARNING: failed to read baseline '{a.baseline}': {e}")
        else:
            budget = 40.0 * (a.width * a.height) / (512.0 * 512.0)
            limit, why = budget * a.budget_mult, f"scaled budget {budget:.3f} ms x {a.budget_mult:.2f}"
        if limit is not None and p95 > limit:
            print(f"FAIL: p95 {p95:.3f} ms > {limit:.3f} ms ({why})")
            return 2
    print("Performance sanity OK")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
