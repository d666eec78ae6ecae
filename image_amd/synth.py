"""Deterministic synthetic frames G(seed) for parity tests and bench.py.

The reference ships no synthetic data; SURVEY.md section 8(d) asks for seeded
frames that exist identically on host (numpy, here) and on device
(``imgfd_synth_frame`` in csrc/synth.hip): a smooth triangle-wave background,
``n_rect`` axis-aligned filled rectangles (grey 30..230, later ones paint over
earlier ones) and per-pixel hash noise 0..15 so that the Harris response has no
exact ties.

Rectangle parameters come from the LCG ``s <- s*1664525 + 1013904223`` (uint32),
taking ``s >> 8``; the noise is a stateless 32-bit mix of (seed, pixel index) so
it can be evaluated per pixel on the GPU.
"""
from __future__ import annotations

import numpy as np

_M32 = 0xFFFFFFFF


def _lcg(state: int) -> int:
    return (state * 1664525 + 1013904223) & _M32


def rectangles(seed: int, width: int, height: int, n_rect: int) -> np.ndarray:
    """(n_rect, 5) int32 rows ``x0, y0, x1, y1, value`` (x1/y1 exclusive, clipped)."""
    s = (seed * 2654435761 + 12345) & _M32
    out = np.zeros((n_rect, 5), dtype=np.int32)
    for r in range(n_rect):
        s = _lcg(s); x0 = (s >> 8) % width
        s = _lcg(s); y0 = (s >> 8) % height
        s = _lcg(s); w = 8 + (s >> 8) % 120
        s = _lcg(s); h = 8 + (s >> 8) % 120
        s = _lcg(s); v = 30 + (s >> 8) % 201
        out[r] = (x0, y0, min(width, x0 + w), min(height, y0 + h), v)
    return out


def rectangles_batch(seed0: int, n_frames: int, width: int, height: int, n_rect: int) -> np.ndarray:
    """``rectangles(seed0 + f, ...)`` for f in range(n_frames), shape (n_frames, n_rect, 5) int32: the same LCG walked for
    all frames at once (10 000-frame streams need 30 M LCG steps; one Python loop iteration per step is too slow)."""
    s = ((np.arange(seed0, seed0 + n_frames, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(12345)) & np.uint64(_M32))
    out = np.zeros((n_frames, n_rect, 5), dtype=np.int64)

    def nxt(st):
        return (st * np.uint64(1664525) + np.uint64(1013904223)) & np.uint64(_M32)
    for r in range(n_rect):
        s = nxt(s); x0 = (s >> np.uint64(8)) % np.uint64(width)
        s = nxt(s); y0 = (s >> np.uint64(8)) % np.uint64(height)
        s = nxt(s); w = np.uint64(8) + (s >> np.uint64(8)) % np.uint64(120)
        s = nxt(s); h = np.uint64(8) + (s >> np.uint64(8)) % np.uint64(120)
        s = nxt(s); v = np.uint64(30) + (s >> np.uint64(8)) % np.uint64(201)
        out[:, r, 0] = x0; out[:, r, 1] = y0
        out[:, r, 2] = np.minimum(np.uint64(width), x0 + w); out[:, r, 3] = np.minimum(np.uint64(height), y0 + h)
        out[:, r, 4] = v
    return out.astype(np.int32)


def hash_noise(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """lowbias32 mix of (seed*0x9E3779B9 + index), top 4 bits -> 0..15 (uint8)."""
    x = (np.arange(offset, offset + n, dtype=np.uint64) + ((seed * 0x9E3779B9) & _M32)) & _M32
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x.astype(np.uint64) * 0x7FEB352D & _M32).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x.astype(np.uint64) * 0x846CA68B & _M32).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return (x >> np.uint32(28)).astype(np.uint8)


def default_rects(width: int, height: int) -> int:
    """600 rectangles on a 4K frame, scaled by area, at least 8."""
    return max(8, int(round(600 * (width * height) / (3840 * 2160))))


def frame(seed: int, width: int, height: int, n_rect: int | None = None) -> np.ndarray:
    """uint8 image, shape (height, width), row-major (index y*width + x)."""
    if n_rect is None:
        n_rect = default_rects(width, height)
    xs = np.arange(width, dtype=np.int32)[None, :]
    ys = np.arange(height, dtype=np.int32)[:, None]
    t = (xs + 2 * ys) & 255
    tri = np.where(t < 128, t, 255 - t)
    img = (60 + (tri >> 1)).astype(np.int32)
    for x0, y0, x1, y1, v in rectangles(seed, width, height, n_rect):
        img[y0:y1, x0:x1] = v
    img += hash_noise(seed, width * height).reshape(height, width)
    return np.clip(img, 0, 255).astype(np.uint8)


def frame_rgb(seed: int, width: int, height: int, n_rect: int | None = None) -> np.ndarray:
    """uint8 RGB image (height, width, 3): channel c is ``frame(3*seed + c)`` (config 4)."""
    return np.stack([frame(3 * seed + c, width, height, n_rect) for c in range(3)], axis=-1)
