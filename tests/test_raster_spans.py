"""The span solver of the tile kernel's fast raster path (vulkan_forge_amd/csrc/vf_raster.h: FP32 first, exact FP64 only where FP32
cannot decide) compiled for the HOST and checked line by line against a brute-force int64 evaluation of the coverage rule
(pixel centres, top-left rule).  The hardware's reciprocal is a 1-ulp estimate: the harness runs with the
host reciprocal as is and pushed one ulp up / down.  The GPU parity tests then check the same header as device code against the
oracle."""
import os
import subprocess

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "cpp", "raster_fuzz.cpp")


@pytest.mark.parametrize("ulps", [0, 1, -1])
def test_span_solver_against_brute_force(tmp_path, ulps):
    exe = tmp_path / f"raster_fuzz_{ulps}"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", f"-DVF_RASTER_RCP_ULPS={ulps}", SRC, "-o", str(exe)], check=True)
    r = subprocess.run([str(exe), "250000", str(20250816 + ulps)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    summary = [l for l in r.stdout.splitlines() if l.startswith("triangles")][0]
    assert "failures 0" in summary, summary
    # the terrain's primitives (slivers, general and sub-pixel triangles: kinds 2-6, 8) must be decided in FP32 almost always
    for line in r.stdout.splitlines():
        f = line.split()
        if f[:1] == ["kind"] and int(f[1].rstrip(":")) in (2, 3, 4, 5, 6, 8):
            assert float(f[5]) < 0.2 and float(f[10]) < 0.1, line      # irregular %, fallback % of lines
