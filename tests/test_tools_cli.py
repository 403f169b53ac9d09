"""CLI tools: report protocol of the reference's python/tools/perf_sanity.py and
determinism_harness.py (keys, percentile rule, enforcement switches).  Host logic here; the GPU runs are in
tests/test_gpu_api.py."""
import json

import numpy as np
import pytest

from vulkan_forge_amd.tools import _stats, device_diagnostics, perf_sanity


def test_percentile_and_summary_follow_the_reference_protocol():
    # linear interpolation between order statistics (perf_sanity.py:31-38): p95 of 1..4 = 3.85
    assert _stats.percentile([1, 2, 3, 4], 95.0) == pytest.approx(3.85)
    assert _stats.percentile([7.0], 95.0) == 7.0 and np.isnan(_stats.percentile([], 50.0))
    vals = [float(v) for v in np.random.default_rng(0).random(101)]
    assert _stats.percentile(sorted(vals), 50.0) == pytest.approx(np.percentile(vals, 50.0))
    assert _stats.percentile(sorted(vals), 95.0) == pytest.approx(np.percentile(vals, 95.0))
    s = _stats.summary([3.0, 1.0, 2.0])
    assert set(s) == {"samples_ms", "mean_ms", "median_ms", "p95_ms", "stdev_ms", "min_ms", "max_ms"}      # :56-64
    assert s["samples_ms"] == [3.0, 1.0, 2.0] and s["mean_ms"] == 2.0 and s["median_ms"] == 2.0
    assert s["stdev_ms"] == pytest.approx(np.std([1, 2, 3])) and (s["min_ms"], s["max_ms"]) == (1.0, 3.0)
    assert _stats.summary([5.0])["stdev_ms"] == 0.0


def test_device_diagnostics_report_without_a_gpu_is_not_fatal(tmp_path, capsys):
    out = tmp_path / "sub" / "diag.json"
    rc = device_diagnostics.main(["--json", str(out), "--summary"])
    rep = json.loads(out.read_text())
    assert rc == 0 and "hip" in rep["backends"] and rep["errors"] == []
    assert "Diagnostics OK" in capsys.readouterr().out


def test_perf_enforcement_rules(tmp_path, monkeypatch, capsys):
    """VF_ENFORCE_PERF=1: baseline p95 x (1 + regress%) or the scaled 40 ms @ 512x512 budget (perf_sanity.py:110-129)."""
    fake = {"width": 512, "height": 512, "runs": 3, "warmups": 0, "workload": "triangle", "init_ms": 1.0,
            "steady": _stats.summary([10.0, 10.0, 200.0])}
    monkeypatch.setattr(perf_sanity, "measure", lambda *a, **k: fake)
    args = ["--width", "512", "--height", "512", "--json", str(tmp_path / "r.json")]
    assert perf_sanity.main(args) == 0                                   # never fails by default
    monkeypatch.setenv("VF_ENFORCE_PERF", "1")
    assert perf_sanity.main(args) == 2                                   # p95 181 ms > 40 * 3.0
    assert perf_sanity.main(args + ["--budget-mult", "10"]) == 0
    base = tmp_path / "base.json"
    base.write_text(json.dumps({"steady": {"p95_ms": 150.0}}))
    assert perf_sanity.main(args + ["--baseline", str(base)]) == 0       # 181 <= 150 * 1.5
    assert perf_sanity.main(args + ["--baseline", str(base), "--regress-pct", "10"]) == 2
    assert perf_sanity.main(args + ["--baseline", str(tmp_path / "missing.json")]) == 0   # unreadable baseline: warning only
    assert "WARNING" in capsys.readouterr().out
    csv = tmp_path / "t.csv"
    monkeypatch.delenv("VF_ENFORCE_PERF")
    assert perf_sanity.main(args + ["--csv", str(csv)]) == 0
    assert csv.read_text().splitlines()[0] == "iter,ms" and len(csv.read_text().splitlines()) == 4
