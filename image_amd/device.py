"""Device-resident batch path: torch CUDA tensors in, torch CUDA tensors out, no host copies.

torch is plumbing here (device memory, the current stream, torch.distributed in stream.py); every
computation is a kernel of libimgfd.so launched through the ``*_dev`` entry points of the C ABI on
torch's current stream.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _binding, _lib, synth


class DeviceDetector:
    """One imgfd context bound to ``torch.cuda.current_stream(device)``."""

    def __init__(self, device: int | None = None):
        if not torch.cuda.is_available():
            raise RuntimeError("image_amd.device needs a HIP device (no CPU fallback)")
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.stream = torch.cuda.current_stream(self.device)
        self.ctx = _lib.Context(self.device, stream=self.stream.cuda_stream)
        self.lib = self.ctx.lib

    # ------------------------------------------------------------------ helpers
    def _frames(self, t: torch.Tensor) -> _binding.Frames:
        if t.dim() == 2:
            t = t[None]
        assert t.is_cuda and t.dim() == 3 and t.stride(2) == 1, "frames must be [n, ny, nx] CUDA tensors, x contiguous"
        dtype = {torch.uint8: 0, torch.float32: 1}[t.dtype]
        esz = t.element_size()
        n, ny, nx = t.shape
        return _binding.Frames(t.data_ptr(), n, nx, ny, t.stride(0) * esz if n > 1 else nx * ny * esz,
                               t.stride(1) * esz, dtype)

    def synth_frames(self, n: int, nx: int, ny: int, seed0: int, n_rect: int | None = None) -> torch.Tensor:
        """n frames G(seed0+f) generated on the device (host twin: image_amd.synth.frame)."""
        if n_rect is None:
            n_rect = synth.default_rects(nx, ny)
        rects = synth.rectangles_batch(seed0, n, nx, ny, n_rect)
        d_rects = torch.from_numpy(rects).to(f"cuda:{self.device}")
        out = torch.empty((n, ny, nx), dtype=torch.uint8, device=f"cuda:{self.device}")
        self.ctx.check(self.lib.imgfd_synth_frames(self.ctx.handle, out.data_ptr(), n, nx, ny, nx * ny, seed0 & 0xFFFFFFFF,
                                                   d_rects.data_ptr(), n_rect), "imgfd_synth_frames")
        self._keep = d_rects  # keep alive until the stream has consumed it
        return out

    # ------------------------------------------------------------------ detectors
    def harris(self, frames: torch.Tensor, cap: int = 65536, k=0.06, sigma_d=1.0, sigma_i=2.5, threshold=130.0,
               gaussian=0, gradient=0, measure=0, out=None):
        fr = self._frames(frames)
        dev = frames.device
        if out is None:
            out = (torch.empty((fr.n_frames, cap, 3), dtype=torch.float32, device=dev),
                   torch.empty((fr.n_frames,), dtype=torch.int64, device=dev))
        corners, counts = out
        self.ctx.check(self.lib.imgfd_harris_dev(self.ctx.handle, C.byref(fr), k, sigma_d, sigma_i, threshold, gaussian,
                                                 gradient, measure, corners.data_ptr(), corners.shape[1],
                                                 counts.data_ptr()), "imgfd_harris_dev")
        return corners, counts

    def fast9(self, frames: torch.Tensor, threshold: int = 50, suppress_non_max: bool = False, cap: int = 262144,
              out=None):
        fr = self._frames(frames)
        dev = frames.device
        if out is None:
            out = (torch.empty((fr.n_frames, cap, 2), dtype=torch.int32, device=dev),
                   torch.empty((fr.n_frames,), dtype=torch.int64, device=dev))
        points, counts = out
        self.ctx.check(self.lib.imgfd_fast9_dev(self.ctx.handle, C.byref(fr), int(threshold) & 0xFF,
                                                int(bool(suppress_non_max)), points.data_ptr(), points.shape[1],
                                                counts.data_ptr()), "imgfd_fast9_dev")
        return points, counts

    def canny(self, frames: torch.Tensor, s=2.0, low_thr=3.0, high_thr=10.0, accGrad=True, out=None):
        fr = self._frames(frames)
        dev = frames.device
        if out is None:
            out = (torch.empty((fr.n_frames, fr.ny, fr.nx), dtype=torch.uint8, device=dev),
                   torch.empty((fr.n_frames,), dtype=torch.int64, device=dev))
        edges, counts = out
        self.ctx.check(self.lib.imgfd_canny_dev(self.ctx.handle, C.byref(fr), float(s), float(low_thr), float(high_thr),
                                                int(bool(accGrad)), edges.data_ptr(), counts.data_ptr()),
                       "imgfd_canny_dev")
        return edges, counts

    def detect_all(self, frames: torch.Tensor, corners: torch.Tensor, points: torch.Tensor, edges: torch.Tensor,
                   counts: torch.Tensor, **params):
        """Harris + FAST-9 + Canny on one batch, overlapped on two HIP streams (imgfd_detect_dev).  corners [n, cap, 3]
        f32, points [n, cap, 2] i32, edges [n, ny, nx] u8, counts [3, n] i64 (harris, fast9, canny); params are fields of
        imgfd_stream_params (defaults: the three R functions' defaults)."""
        fr = self._frames(frames)
        p = _binding.StreamParams()
        self.lib.imgfd_stream_default_params(C.byref(p))
        p.corner_cap, p.point_cap = corners.shape[1], points.shape[1]
        for k, v in params.items():
            setattr(p, k, v)
        assert counts.is_contiguous() and counts.numel() == 3 * fr.n_frames
        self.ctx.check(self.lib.imgfd_detect_dev(self.ctx.handle, C.byref(fr), C.byref(p), corners.data_ptr(), points.data_ptr(),
                                                 edges.data_ptr(), counts.data_ptr()), "imgfd_detect_dev")
        return counts

    def fhog(self, tiles: torch.Tensor, out: torch.Tensor, cell_size=8, pad_r=1, pad_c=1):
        """imgfd_fhog_dev: tiles [n, rows, cols, 3] u8 -> out [n, 31, hog_nc, hog_nr] f32 (the reference glue's layout)."""
        n, rows, cols, _ = tiles.shape
        assert tiles.is_contiguous() and out.is_contiguous()
        self.ctx.check(self.lib.imgfd_fhog_dev(self.ctx.handle, tiles.data_ptr(), n, rows, cols, rows * cols * 3, cell_size,
                                               pad_r, pad_c, out.data_ptr()), "imgfd_fhog_dev")
        return out

    def surf(self, tiles: torch.Tensor, feat: torch.Tensor, counts: torch.Tensor, max_points=1000, threshold=30.0, redo=True):
        """imgfd_surf_dev: tiles [n, rows, cols, 3] u8 -> feat [n, cap, 70] f64 (x, y, angle, scale, score, laplacian,
        surf[64]), counts [n] i64.  imgfd_surf_dev only queues work; a tile whose candidates overflowed its record buffer
        reports a NEGATIVE count.  redo=True (default) follows it with imgfd_surf_dev_redo -- one wait for the stream, the
        counts read back once, such tiles redone with a larger buffer -- so that the counts are final when this returns;
        redo=False leaves the call asynchronous: check ``(counts < 0).any()`` before slicing with them."""
        n, rows, cols, _ = tiles.shape
        assert tiles.is_contiguous() and feat.is_contiguous()
        args = (self.ctx.handle, tiles.data_ptr(), n, rows, cols, rows * cols * 3, int(max_points), float(threshold), feat.data_ptr(),
                feat.shape[1], counts.data_ptr())
        self.ctx.check(self.lib.imgfd_surf_dev(*args), "imgfd_surf_dev")
        if redo:
            self.ctx.check(self.lib.imgfd_surf_dev_redo(*args, None), "imgfd_surf_dev_redo")
        return feat, counts

    def gradients_of(self, frame: torch.Tensor, ix: torch.Tensor, iy: torch.Tensor, sigma_d=1.0):
        """Ix, Iy of one u8/f32 frame as image_harris() computes them (stage doorways K1 + K2), into ix / iy [ny, nx] f32."""
        ny, nx = frame.shape
        f = frame.to(torch.float32).contiguous()
        sm = torch.empty_like(f)
        self.ctx.check(self.lib.imgfd_k_gaussian(self.ctx.handle, f.data_ptr(), sm.data_ptr(), nx, ny, sigma_d, 0), "imgfd_k_gaussian")
        self.ctx.check(self.lib.imgfd_k_gradient(self.ctx.handle, sm.data_ptr(), ix.data_ptr(), iy.data_ptr(), nx, ny, 0), "imgfd_k_gradient")
        self.ctx.sync()  # f, sm die here

    def time_structure_tensor_batch(self, ix: torch.Tensor, iy: torch.Tensor, sigma=2.5, gauss=0, warmup=3, iters=20, probe_us=0, out=None):
        """Mean microseconds per launch of the 20 B/px structure-tensor kernel over a batch [n, ny, nx] (HIP events on
        the context's stream, back-to-back launches).  probe_us > 0: one shader-clock probe of that span is queued right before the
        launches (after the output planes exist: allocating them takes milliseconds), so that it samples the clock while they run."""
        n, ny, nx = ix.shape
        A, B, Cc = out if out is not None else tuple(torch.empty_like(ix) for _ in range(3))   # out: planes of an earlier call (already touched)
        torch.cuda.synchronize()
        if probe_us:
            self.clock_probe(probe_us)
        us = C.c_double(0)
        self.ctx.check(self.lib.imgfd_time_structure_tensor_batch(self.ctx.handle, ix.data_ptr(), iy.data_ptr(), A.data_ptr(),
                                                                  B.data_ptr(), Cc.data_ptr(), nx, ny, n, sigma, gauss, warmup,
                                                                  iters, C.byref(us)), "imgfd_time_structure_tensor_batch")
        return us.value

    def profile_k3_read(self):
        """(summed microseconds, launches) of the structure-tensor launches since imgfd_profile_k3(ctx, 1)."""
        us, n = C.c_double(0), C.c_int(0)
        self.ctx.check(self.lib.imgfd_profile_k3_read(self.ctx.handle, C.byref(us), C.byref(n)), "imgfd_profile_k3_read")
        return us.value, n.value

    def clock_probe(self, span_us=200):
        """queue one shader-clock sample beside whatever runs (imgfd_clock_probe: one wavefront on a stream of its own)"""
        self.ctx.check(self.lib.imgfd_clock_probe(self.ctx.handle, int(span_us)), "imgfd_clock_probe")

    def clock_probe_read(self):
        """{"mean_GHz", "min_GHz", "max_GHz", "samples"} over the probes queued since the last read (waits for them)"""
        a, b, c, n = C.c_double(0), C.c_double(0), C.c_double(0), C.c_int(0)
        self.ctx.check(self.lib.imgfd_clock_probe_read(self.ctx.handle, C.byref(a), C.byref(b), C.byref(c), C.byref(n)), "imgfd_clock_probe_read")
        return {"mean_GHz": round(a.value, 4), "min_GHz": round(b.value, 4), "max_GHz": round(c.value, 4), "samples": n.value}

    def tensor_kernel_name(self) -> str:
        """which kernel imgfd_harris_dev / imgfd_detect_dev launch for the structure-tensor pass on the default path"""
        name = self.lib.imgfd_tensor_kernel_name(self.ctx.handle)
        return (name or b"?").decode()

    def time_structure_tensor(self, ix: torch.Tensor, iy: torch.Tensor, sigma=2.5, gauss=0, warmup=5, iters=50):
        """Mean microseconds per launch of the structure-tensor kernel (HIP events on the ctx stream)."""
        ny, nx = ix.shape[-2:]
        A, B, Cc = (torch.empty_like(ix) for _ in range(3))
        us = C.c_double(0)
        self.ctx.check(self.lib.imgfd_time_structure_tensor(self.ctx.handle, ix.data_ptr(), iy.data_ptr(), A.data_ptr(),
                                                            B.data_ptr(), Cc.data_ptr(), nx, ny, sigma, gauss, warmup,
                                                            iters, C.byref(us)), "imgfd_time_structure_tensor")
        return us.value
