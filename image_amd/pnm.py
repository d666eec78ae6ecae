"""PGM (P2 ASCII / P5 binary) reader and writer.

The reference examples feed the detectors from PGM files through R's ``pixmap``
package (image.CornerDetectionF9/R/image_detect_corners.R:10-27,
image.CannyEdges/R/canny_edges_detector.R:25-60); this is the ingest step in
front of the hot path.
"""
from __future__ import annotations

import numpy as np


def _tokens(buf: bytes, count: int):
    """First ``count`` whitespace-separated header tokens (comments skipped) and the offset after them."""
    toks, i, n = [], 0, len(buf)
    while len(toks) < count:
        while i < n and buf[i:i + 1].isspace():
            i += 1
        if i < n and buf[i:i + 1] == b"#":
            while i < n and buf[i:i + 1] != b"\n":
                i += 1
            continue
        j = i
        while j < n and not buf[j:j + 1].isspace():
            j += 1
        toks.append(buf[i:j])
        i = j
    return toks, i


def read_pgm(path: str) -> np.ndarray:
    """Return a (height, width) uint8/uint16 array."""
    with open(path, "rb") as f:
        buf = f.read()
    (magic, w, h, maxval), off = _tokens(buf, 4)
    w, h, maxval = int(w), int(h), int(maxval)
    if magic == b"P5":
        off += 1  # single whitespace after maxval
        dt = np.dtype(">u2") if maxval > 255 else np.dtype("u1")
        return np.frombuffer(buf, dtype=dt, count=w * h, offset=off).reshape(h, w).astype(
            np.uint16 if maxval > 255 else np.uint8)
    if magic == b"P2":
        body = buf[off:]
        vals = np.array(body.split(), dtype=np.int64)
        if vals.size < w * h:
            raise ValueError("truncated P2 file")
        return vals[: w * h].reshape(h, w).astype(np.uint16 if maxval > 255 else np.uint8)
    raise ValueError(f"not a PGM file: magic {magic!r}")


def write_pgm(path: str, img: np.ndarray) -> None:
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (w, h))
        f.write(img.tobytes())
