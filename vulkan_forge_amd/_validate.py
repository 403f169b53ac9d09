"""Argument guards of the Python shim.

Same bounds and exception classes as the reference's python/vulkan_forge/_validate.py
(size_wh :15-22, grid :24-30, png_path :32-38), re-expressed for this package.
"""
from __future__ import annotations

import os

MAX_DIM = 8192  # headless target guard rail (reference: _validate.py:6)
MAX_GRID = 4096


def _to_int(name, value) -> int:
    try:
        return int(value)
    except Exception as exc:  # noqa: BLE001 - like the reference: anything non-integral is a ValueError
        raise ValueError(f"{name} must be an integer, got {type(value).__name__}") from exc


def size_wh(width, height):
    w, h = _to_int("width", width), _to_int("height", height)
    if w <= 0 or h <= 0:
        raise ValueError("width and height must be > 0")
    if w > MAX_DIM or h > MAX_DIM:
        raise ValueError(f"width/height must be <= {MAX_DIM}")
    return w, h


def grid(n) -> int:
    g = _to_int("grid", n)
    if g < 2:
        raise ValueError("grid must be >= 2")
    if g > MAX_GRID:
        raise ValueError(f"grid must be <= {MAX_GRID}")
    return g


def png_path(p) -> str:
    s = str(p)
    if not s.lower().endswith(".png"):
        raise ValueError("path must end with .png")
    parent = os.path.dirname(os.path.realpath(s))
    if not os.path.isdir(parent):
        raise ValueError(f"directory does not exist: {parent}")
    return s
