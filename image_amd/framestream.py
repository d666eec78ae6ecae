"""Host frame streams: u8 frames in host memory -> Harris / FAST-9 / Canny with the upload overlapped.

Python face of ``imgfd_stream_*`` (include/imgfd.h, image_amd/csrc/frame_stream.hip).  The reference has one image per
R call (H/R/pkg.R:76-90, F9/R/image_detect_corners.R:10-27, CE/R/canny_edges_detector.R:63); this is the loop a user
writes around those calls, with defaults equal to the three R functions' defaults.

    with FrameStream(3840, 2160, batch=32, corner_cap=4096) as fs:
        for res in fs.run(frames_iter):          # frames_iter yields (n, ny, nx) uint8 arrays
            res["harris_counts"], res["corners"][f], ...
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Iterator

import numpy as np

from . import _binding, _lib


class PinnedFrames:
    """(n, ny, nx) uint8 array in pinned host memory (``imgfd_host_alloc``): the DMA engine reads it directly."""

    def __init__(self, n: int, ny: int, nx: int, lib: C.CDLL | None = None):
        self.lib = lib if lib is not None else _lib.load()
        self.nbytes = n * ny * nx
        self.ptr = self.lib.imgfd_host_alloc(self.nbytes)
        if not self.ptr:
            raise MemoryError(f"imgfd_host_alloc({self.nbytes}) failed")
        buf = (C.c_uint8 * self.nbytes).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=np.uint8).reshape(n, ny, nx)

    def free(self) -> None:
        if self.ptr:
            self.array = None
            self.lib.imgfd_host_free(self.ptr)
            self.ptr = None

    def __del__(self):  # pragma: no cover
        try:
            self.free()
        except Exception:
            pass


class FrameStream:
    def __init__(self, nx: int, ny: int, batch: int = 32, *, ctx=None, harris: bool = True, fast9: bool = True,
                 canny: bool = True, corner_cap: int = 0, point_cap: int = 0, keep_edges: bool = False, **params):
        self.ctx = ctx if ctx is not None else _lib.Context()
        self.lib = self.ctx.lib
        self.nx, self.ny, self.batch = nx, ny, batch
        p = _binding.StreamParams()
        self.lib.imgfd_stream_default_params(C.byref(p))
        p.harris, p.fast9, p.canny = int(harris), int(fast9), int(canny)
        p.corner_cap, p.point_cap, p.keep_edges = corner_cap, point_cap, int(keep_edges)
        names = {f[0] for f in _binding.StreamParams._fields_}
        for k, v in params.items():
            if k not in names:
                raise TypeError(f"FrameStream: unknown parameter {k!r}")
            setattr(p, k, v)
        self.params = p
        self.handle = C.c_void_p()
        self.ctx.check(self.lib.imgfd_stream_open(self.ctx.handle, nx, ny, batch, C.byref(p), C.byref(self.handle)),
                       "imgfd_stream_open")
        self._keep = []  # submitted arrays stay referenced until their batch is collected

    # ---- the three C calls
    def submit(self, frames: np.ndarray) -> None:
        if frames.dtype != np.uint8 or frames.ndim != 3 or frames.shape[1:] != (self.ny, self.nx):
            raise ValueError(f"frames must be uint8 (n, {self.ny}, {self.nx})")
        if not frames[0].flags.c_contiguous:
            frames = np.ascontiguousarray(frames)
        stride = frames.strides[0] if frames.shape[0] > 1 else self.nx * self.ny
        self.ctx.check(self.lib.imgfd_stream_submit(self.handle, C.c_void_p(frames.ctypes.data), frames.shape[0], stride),
                       "imgfd_stream_submit")
        self._keep.append(frames)

    def collect(self, copy: bool = True) -> dict | None:
        r = _binding.StreamResult()
        self.ctx.check(self.lib.imgfd_stream_collect(self.handle, C.byref(r)), "imgfd_stream_collect")
        if r.n_frames == 0:
            return None
        if self._keep:
            self._keep.pop(0)
        n, p = r.n_frames, self.params
        out = {"n_frames": n, "first_frame": r.first_frame}

        def view(ptr, dtype, shape):
            count = int(np.prod(shape))
            if not count:
                return np.zeros(shape, dtype)
            a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), (count * np.dtype(dtype).itemsize,)).view(dtype)
            a = a.reshape(shape)
            return a.copy() if copy else a

        for key in ("harris_counts", "fast9_counts", "canny_counts"):
            ptr = getattr(r, key)
            out[key] = view(ptr, np.int64, (n,)) if ptr else None
        corner_t = np.dtype([("x", np.float32), ("y", np.float32), ("R", np.float32)])
        point_t = np.dtype([("x", np.int32), ("y", np.int32)])
        if r.corners:
            rec = view(r.corners, corner_t, (n, p.corner_cap))
            out["corners"] = [rec[f, :min(int(out["harris_counts"][f]), p.corner_cap)] for f in range(n)]
        if r.points:
            rec = view(r.points, point_t, (n, p.point_cap))
            out["points"] = [rec[f, :min(int(out["fast9_counts"][f]), p.point_cap)] for f in range(n)]
        if r.edges:
            out["edges"] = view(r.edges, np.uint8, (n, self.ny, self.nx))
        return out

    def close(self) -> None:
        if self.handle:
            self.lib.imgfd_stream_close(self.handle)
            self.handle = C.c_void_p()
            self._keep.clear()

    # ---- convenience: keep two batches in flight
    def run(self, batches: Iterable[np.ndarray]) -> Iterator[dict]:
        pending = 0
        for frames in batches:
            if pending == 2:
                yield self.collect()
                pending -= 1
            self.submit(frames)
            pending += 1
        while pending:
            yield self.collect()
            pending -= 1

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
