#!/usr/bin/env python3
"""TerrainSpike CLI (flags of the reference's python/tools/terrain_spike.py:6-12): render the analytic terrain to a PNG."""
from __future__ import annotations

import argparse


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--grid", type=int, default=160)
    ap.add_argument("--out", default="terrain_spike.png")
    ap.add_argument("--colormap", default="viridis")
    a = ap.parse_args(argv)
    import vulkan_forge_amd as vf
    vf.TerrainSpike(a.width, a.height, a.grid, a.colormap).render_png(a.out)
    print(f"Wrote {a.out}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
