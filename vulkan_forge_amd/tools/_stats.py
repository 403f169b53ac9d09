"""Shared helpers for the CLI tools: order statistics and JSON output."""
from __future__ import annotations

import json
import math
import os
import statistics


def percentile(sorted_values, p):
    """Linear-interpolated percentile of an ascending list (p in 0..100); NaN for an empty list."""
    if not sorted_values:
        return float("nan")
    pos = (len(sorted_values) - 1) * p / 100.0
    lo, hi = math.floor(pos), math.ceil(pos)
    return sorted_values[lo] + (sorted_values[hi] - sorted_values[lo]) * (pos - lo)


def summary(samples_ms):
    """The reference report's `steady` block: samples_ms, mean, median, p95, stdev (population), min, max."""
    nan = float("nan")
    ordered = sorted(samples_ms)
    return {
        "samples_ms": list(samples_ms),
        "mean_ms": statistics.fmean(samples_ms) if samples_ms else nan,
        "median_ms": statistics.median(samples_ms) if samples_ms else nan,
        "p95_ms": percentile(ordered, 95.0),
        "stdev_ms": statistics.pstdev(samples_ms) if len(samples_ms) > 1 else 0.0,
        "min_ms": ordered[0] if ordered else nan,
        "max_ms": ordered[-1] if ordered else nan,
    }


def write_json(path, report, echo=True):
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w", encoding="utf-8") as f:
        json.dump(report, f, indent=2)
    if echo:
        print(json.dumps(report, indent=2))
