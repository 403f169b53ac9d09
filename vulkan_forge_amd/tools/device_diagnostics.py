#!/usr/bin/env python3
"""Device diagnostics (the reference's python/tools/device_diagnostics.py enumerates wgpu adapters per backend; here the
only backend is HIP): enumerate_adapters() + device_probe() from the drop-in module into a JSON report."""
from __future__ import annotations

import argparse
import platform
import sys

from ._stats import write_json


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--json", default="device_diagnostics.json")
    ap.add_argument("--summary", action="store_true")
    ap.add_argument("--backends", nargs="*", default=None, help="accepted for compatibility; 'hip' is the only backend")
    a = ap.parse_args(argv)
    import vulkan_forge_amd as vf
    report = {"python": sys.version.split()[0], "platform": platform.platform(), "version": vf.__version__, "backends": {}, "errors": []}
    try:
        adapters = vf.enumerate_adapters()
        report["backends"]["hip"] = {"adapters": adapters}
        if adapters:
            report["backends"]["hip"]["probe"] = vf.device_probe("hip")
    except Exception as e:  # noqa: BLE001
        report["errors"].append(str(e))
    write_json(a.json, report, echo=a.summary)
    if not report["backends"].get("hip", {}).get("adapters"):
        print("No supported backends detected (not fatal).")
    if report["errors"]:
        print("Diagnostics found errors. See JSON.")
        return 1
    print("Diagnostics OK")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
