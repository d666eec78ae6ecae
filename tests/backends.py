"""Two ways to drive the C ABI in tests.

GpuBackend  -- the product: image_amd/libimgfd.so on a real device, buffers are torch CUDA tensors.
EmuBackend  -- TEST ONLY: the same kernel sources compiled for the host against tests/hipemu (a fiber
               based HIP emulator); "device" buffers are numpy arrays.  It checks kernel logic on the
               GPU-less build machine and is never reachable from image_amd.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from image_amd import _binding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Base:
    name = "?"

    def __init__(self, lib, stream=None):
        self.lib = lib
        self.ctx = C.c_void_p()
        if stream is None:
            st = lib.imgfd_ctx_create(0, C.byref(self.ctx))
        else:   # a context on the caller's HIP stream (imgfd_ctx_create_on_stream)
            st = lib.imgfd_ctx_create_on_stream(0, stream, C.byref(self.ctx))
        assert st == 0, f"imgfd_ctx_create -> {st}"

    def check(self, st, what=""):
        _binding.check(self.lib, self.ctx, st, what)

    def close(self):
        if self.ctx:
            self.lib.imgfd_ctx_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def set_fir_mode(self, mode):
        self.check(self.lib.imgfd_set_fir_mode(self.ctx, mode))

    def set_tuning(self, name, value):
        self.check(self.lib.imgfd_set_tuning(self.ctx, name.encode(), int(value)), "imgfd_set_tuning")

    def get_counter(self, name):
        import ctypes as C
        v = C.c_int64(0)
        self.check(self.lib.imgfd_get_counter(self.ctx, name.encode(), C.byref(v)), "imgfd_get_counter")
        return int(v.value)

    def k_fhog_lut(self, arith=False):
        out = self.empty((511, 512), np.uint32)
        fn = self.lib.imgfd_k_fhog_lut_arith if arith else self.lib.imgfd_k_fhog_lut
        self.check(fn(self.ctx, self.ptr(out)), "k_fhog_lut")
        self.sync()
        return self.to_host(out)

    def frames(self, dev, n, nx, ny, dtype):
        esz = 1 if dtype == 0 else 4
        return _binding.Frames(self.ptr(dev), n, nx, ny, nx * ny * esz, nx * esz, dtype)

    # frames that are NOT packed: rows `row_pad` bytes apart beyond their pixels, `gap_rows` unused rows between frames
    _pad = None

    def padded(self, row_pad, gap_rows):
        """context manager: the *_dev helpers of this backend upload u8 frames with padded rows / gaps between frames"""
        be = self

        class _P:
            def __enter__(self_):
                be._pad = (row_pad, gap_rows)

            def __exit__(self_, *a):
                be._pad = None
        return _P()

    def upload_frames_u8(self, frames):
        """-> (device buffer, imgfd_frames): packed, or laid out as self.padded() says (the padding holds 0xAB)"""
        n, ny, nx = frames.shape
        if not self._pad:
            d = self.to_dev(np.ascontiguousarray(frames, np.uint8))
            return d, self.frames(d, n, nx, ny, 0)
        row_pad, gap = self._pad
        pitch = nx + row_pad
        host = np.full((n, ny + gap, pitch), 0xAB, np.uint8)
        host[:, :ny, :nx] = frames
        d = self.to_dev(host)
        return d, _binding.Frames(self.ptr(d), n, nx, ny, (ny + gap) * pitch, pitch, 0)

    # ---- stage doorways on host arrays -----------------------------------------------------
    def k_gaussian(self, img, sigma, type=0):
        ny, nx = img.shape
        d_in = self.to_dev(np.ascontiguousarray(img, np.float32)); d_out = self.empty((ny, nx), np.float32)
        self.check(self.lib.imgfd_k_gaussian(self.ctx, self.ptr(d_in), self.ptr(d_out), nx, ny, sigma, type), "k_gaussian")
        return self.to_host(d_out)

    def k_gradient(self, img, type=0):
        ny, nx = img.shape
        d_in = self.to_dev(np.ascontiguousarray(img, np.float32))
        ix = self.empty((ny, nx), np.float32); iy = self.empty((ny, nx), np.float32)
        self.check(self.lib.imgfd_k_gradient(self.ctx, self.ptr(d_in), self.ptr(ix), self.ptr(iy), nx, ny, type), "k_gradient")
        return self.to_host(ix), self.to_host(iy)

    def k_gauss_grad_u8(self, img_u8, sigma_d=1.0, type=0):
        ny, nx = img_u8.shape
        d_in = self.to_dev(np.ascontiguousarray(img_u8, np.uint8))
        ix = self.empty((ny, nx), np.float32); iy = self.empty((ny, nx), np.float32)
        self.check(self.lib.imgfd_k_gauss_grad_u8(self.ctx, self.ptr(d_in), self.ptr(ix), self.ptr(iy), nx, ny, sigma_d, type), "k_gauss_grad_u8")
        return self.to_host(ix), self.to_host(iy)

    def k_structure_tensor(self, ix, iy, sigma, gauss=0):
        ny, nx = ix.shape
        dx = self.to_dev(np.ascontiguousarray(ix, np.float32)); dy = self.to_dev(np.ascontiguousarray(iy, np.float32))
        A, B, Cc = (self.empty((ny, nx), np.float32) for _ in range(3))
        self.check(self.lib.imgfd_k_structure_tensor(self.ctx, self.ptr(dx), self.ptr(dy), self.ptr(A), self.ptr(B),
                                                     self.ptr(Cc), nx, ny, sigma, gauss), "k_structure_tensor")
        return self.to_host(A), self.to_host(B), self.to_host(Cc)

    def k_tensor_response(self, ix, iy, sigma, k=0.06):
        ny, nx = ix.shape
        dx = self.to_dev(np.ascontiguousarray(ix, np.float32)); dy = self.to_dev(np.ascontiguousarray(iy, np.float32))
        R = self.empty((ny, nx), np.float32)
        self.check(self.lib.imgfd_k_tensor_response(self.ctx, self.ptr(dx), self.ptr(dy), self.ptr(R), nx, ny, sigma, k), "k_tensor_response")
        return self.to_host(R)

    def k_response(self, A, B, Cc, measure=0, k=0.06):
        ny, nx = A.shape
        d = [self.to_dev(np.ascontiguousarray(p, np.float32)) for p in (A, B, Cc)]
        R = self.empty((ny, nx), np.float32)
        self.check(self.lib.imgfd_k_response(self.ctx, self.ptr(d[0]), self.ptr(d[1]), self.ptr(d[2]), self.ptr(R),
                                             nx, ny, measure, k), "k_response")
        return self.to_host(R)

    def k_nms(self, R, Th, radius, quads=False):
        """quads=True: the batch path's kernels (threshold quads + sparse NMS) instead of the tiled one."""
        ny, nx = R.shape
        d = self.to_dev(np.ascontiguousarray(R, np.float32))
        cap = nx * ny // 4 + 16
        out = self.empty((cap, 3), np.float32); cnt = self.empty((1,), np.int64)
        fn = self.lib.imgfd_k_nms_quads if quads else self.lib.imgfd_k_nms
        self.check(fn(self.ctx, self.ptr(d), nx, ny, Th, radius, self.ptr(out), cap, self.ptr(cnt)), "k_nms")
        n = int(self.to_host(cnt)[0])
        return self.to_host(out)[:n]

    # ---- host-pointer API -------------------------------------------------------------------
    def harris(self, img, **kw):
        p = dict(k=0.06, sigma_d=1.0, sigma_i=2.5, threshold=130.0, gaussian=0, gradient=0, strategy=0,
                 Nselect=1, measure=0, Nscales=1, precision=0, cells=10, verbose=0)
        p.update(kw)
        img = np.ascontiguousarray(img, np.float32)
        ny, nx = img.shape
        out = _binding.Corners()
        self.check(self.lib.imgfd_harris(self.ctx, img.ctypes.data_as(C.c_void_p), nx, ny, p["k"], p["sigma_d"],
                                         p["sigma_i"], p["threshold"], p["gaussian"], p["gradient"], p["strategy"],
                                         p["Nselect"], p["measure"], p["Nscales"], p["precision"], p["cells"],
                                         p["verbose"], C.byref(out)), "imgfd_harris")
        if not out.n:
            return np.zeros((0, 3), np.float32)
        arr = np.ctypeslib.as_array(C.cast(out.corners, C.POINTER(C.c_float)), shape=(out.n, 3)).copy()
        self.lib.imgfd_free(out.corners)
        return arr

    def fast9(self, img, threshold, nonmax=False, width=None):
        img = np.ascontiguousarray(img, np.uint8)
        h, stride = img.shape
        w = stride if width is None else width
        out = _binding.Points()
        self.check(self.lib.imgfd_fast9(self.ctx, img.ctypes.data_as(C.c_void_p), w, h, stride, threshold & 0xFF,
                                        int(nonmax), C.byref(out)), "imgfd_fast9")
        if not out.n:
            return np.zeros((0, 2), np.int32)
        arr = np.ctypeslib.as_array(C.cast(out.points, C.POINTER(C.c_int)), shape=(out.n, 2)).copy()
        self.lib.imgfd_free(out.points)
        return arr

    def canny(self, img, s=2.0, low_thr=3.0, high_thr=10.0, accGrad=True):
        img = np.ascontiguousarray(img, np.uint8)
        ny, nx = img.shape
        edges = np.zeros((ny, nx), np.uint8)
        n = C.c_int64(0)
        self.check(self.lib.imgfd_canny(self.ctx, img.ctypes.data_as(C.c_void_p), nx, ny, s, low_thr, high_thr,
                                        int(accGrad), edges.ctypes.data_as(C.c_void_p), C.byref(n)), "imgfd_canny")
        return edges, int(n.value)

    def fhog(self, rgb, cell_size=8, pad_r=1, pad_c=1):
        """host-pointer imgfd_fhog; returns dlib's array2d order (hog_nr, hog_nc, 31)"""
        rgb = np.ascontiguousarray(rgb, np.uint8)
        rows, cols, _ = rgb.shape
        hog = C.POINTER(C.c_float)(); nr = C.c_int(0); nc = C.c_int(0)
        self.check(self.lib.imgfd_fhog(self.ctx, rgb.ctypes.data_as(C.c_void_p), rows, cols, cell_size, pad_r, pad_c,
                                       C.byref(hog), C.byref(nr), C.byref(nc)), "imgfd_fhog")
        if not nr.value * nc.value:
            return np.zeros((0, 0, 31), np.float32)
        flat = np.ctypeslib.as_array(hog, shape=(31, nc.value, nr.value)).copy()
        self.lib.imgfd_free(hog)
        return np.ascontiguousarray(flat.transpose(2, 1, 0))

    def surf_integral(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        rows, cols, _ = rgb.shape
        out = np.zeros((rows, cols), np.int32)
        self.check(self.lib.imgfd_k_surf_integral(self.ctx, rgb.ctypes.data_as(C.c_void_p), rows, cols,
                                                  out.ctypes.data_as(C.c_void_p)), "imgfd_k_surf_integral")
        return out

    def surf_interest_points(self, rgb, threshold=30.0, cap=400000):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        rows, cols, _ = rgb.shape
        out = np.zeros((cap, 5), np.float64); n = C.c_int64(0)
        self.check(self.lib.imgfd_surf_interest_points(self.ctx, rgb.ctypes.data_as(C.c_void_p), rows, cols, threshold,
                                                       out.ctypes.data_as(C.c_void_p), cap, C.byref(n)), "imgfd_surf_interest_points")
        assert n.value <= cap
        return out[:n.value].copy()

    def surf_points_dev(self, frames, threshold=30.0, cap=4096):
        """batch device path; returns per tile the (n,5) points sorted into the reference's emission order"""
        frames = np.ascontiguousarray(frames, np.uint8)
        n, rows, cols, _ = frames.shape
        d = self.to_dev(frames)
        rec = self.empty((n, cap, 6), np.float64); cnt = self.empty((n,), np.int64)
        self.check(self.lib.imgfd_surf_points_dev(self.ctx, self.ptr(d), n, rows, cols, rows * cols * 3, threshold,
                                                  self.ptr(rec), cap, self.ptr(cnt)), "imgfd_surf_points_dev")
        self.sync()
        rec = self.to_host(rec); cnt = self.to_host(cnt)
        out = []
        for f in range(n):
            r = rec[f, :min(int(cnt[f]), cap)]
            keys = np.ascontiguousarray(r[:, 0]).view(np.uint64)
            out.append(r[np.argsort(keys, kind="stable")][:, 1:])
        return out, cnt

    def knn(self, data, query, k=1, column_major=False):
        data = np.asarray(data, np.float64); query = np.asarray(query, np.float64)
        nd, dim = data.shape
        nq = query.shape[0]
        a = np.asfortranarray(data) if column_major else np.ascontiguousarray(data)
        b = np.asfortranarray(query) if column_major else np.ascontiguousarray(query)
        idx = np.full((nq, k), -7, np.int32); dist = np.zeros((nq, k), np.float64)
        self.check(self.lib.imgfd_knn(self.ctx, a.ctypes.data_as(C.c_void_p), nd, b.ctypes.data_as(C.c_void_p), nq, dim, k,
                                      int(column_major), idx.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p)), "imgfd_knn")
        return idx, dist

    def knn_dev(self, data, query, k, data_strides, query_strides, dim):
        """data/query: flat float64 arrays already laid out by the caller; strides = (row, col) in doubles"""
        nd = data[1]; nq = query[1]
        d = self.to_dev(data[0]); q = self.to_dev(query[0])
        idx = self.empty((nq, k), np.int32); dist = self.empty((nq, k), np.float64)
        self.check(self.lib.imgfd_knn_dev(self.ctx, self.ptr(d), nd, data_strides[0], data_strides[1], self.ptr(q), nq,
                                          query_strides[0], query_strides[1], dim, k, self.ptr(idx), self.ptr(dist)), "imgfd_knn_dev")
        self.sync()
        return self.to_host(idx), self.to_host(dist)

    def surf_dev_counts(self, frames, max_points=1000, threshold=30.0):
        """imgfd_surf_dev alone: the raw per-tile counts (negative: candidates that overflowed the record buffer)"""
        frames = np.ascontiguousarray(frames, np.uint8)
        n, rows, cols, _ = frames.shape
        d = self.to_dev(frames)
        feat = self.empty((n, max_points, 70), np.float64); cnt = self.empty((n,), np.int64)
        self.check(self.lib.imgfd_surf_dev(self.ctx, self.ptr(d), n, rows, cols, rows * cols * 3, max_points, threshold,
                                           self.ptr(feat), max_points, self.ptr(cnt)), "imgfd_surf_dev")
        self.sync()
        return self.to_host(cnt)

    def surf_dev(self, frames, max_points=1000, threshold=30.0, cap=None):
        """batch path with K19 on the device; per tile a dict like surf()"""
        frames = np.ascontiguousarray(frames, np.uint8)
        n, rows, cols, _ = frames.shape
        cap = cap or max_points
        d = self.to_dev(frames)
        feat = self.empty((n, cap, 70), np.float64); cnt = self.empty((n,), np.int64)
        self.check(self.lib.imgfd_surf_dev(self.ctx, self.ptr(d), n, rows, cols, rows * cols * 3, max_points, threshold,
                                           self.ptr(feat), cap, self.ptr(cnt)), "imgfd_surf_dev")
        redone = C.c_int(0)
        self.check(self.lib.imgfd_surf_dev_redo(self.ctx, self.ptr(d), n, rows, cols, rows * cols * 3, max_points, threshold,
                                                self.ptr(feat), cap, self.ptr(cnt), C.byref(redone)), "imgfd_surf_dev_redo")
        self.last_surf_redone = redone.value
        self.sync()
        feat = self.to_host(feat); cnt = self.to_host(cnt)
        out = []
        for f in range(n):
            r = feat[f, :int(cnt[f])]
            out.append(dict(x=r[:, 0], y=r[:, 1], angle=r[:, 2], pyramid_scale=r[:, 3], score=r[:, 4], laplacian=r[:, 5],
                            surf=r[:, 6:]))
        return out

    def surf(self, rgb, max_points=1000, threshold=30.0):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        rows, cols, _ = rgb.shape
        o = _binding.SurfOut()
        self.check(self.lib.imgfd_surf(self.ctx, rgb.ctypes.data_as(C.c_void_p), rows, cols, max_points, threshold,
                                       C.byref(o)), "imgfd_surf")
        n = int(o.n)
        vec = lambda p, m: np.ctypeslib.as_array(p, shape=(m,)).copy() if m else np.zeros((0,))
        res = dict(x=vec(o.x, n), y=vec(o.y, n), angle=vec(o.angle, n), pyramid_scale=vec(o.pyramid_scale, n),
                   score=vec(o.score, n), laplacian=vec(o.laplacian, n), surf=vec(o.surf, n * 64).reshape(n, 64))
        if n:
            self.lib.imgfd_free(o.data)
        return res

    def fhog_dev(self, frames, cell_size=8, pad_r=1, pad_c=1):
        frames = np.ascontiguousarray(frames, np.uint8)
        n, rows, cols, _ = frames.shape
        nr = C.c_int(0); nc = C.c_int(0)
        self.check(self.lib.imgfd_fhog_size(rows, cols, cell_size, pad_r, pad_c, C.byref(nr), C.byref(nc)), "fhog_size")
        d = self.to_dev(frames)
        out = self.empty((n, 31, nc.value, nr.value), np.float32)
        self.check(self.lib.imgfd_fhog_dev(self.ctx, self.ptr(d), n, rows, cols, rows * cols * 3, cell_size, pad_r, pad_c,
                                           self.ptr(out)), "imgfd_fhog_dev")
        self.sync()
        return np.ascontiguousarray(self.to_host(out).transpose(0, 3, 2, 1))

    # ---- device-resident batch API --------------------------------------------------------------
    def harris_dev(self, frames, cap=None, **kw):
        p = dict(k=0.06, sigma_d=1.0, sigma_i=2.5, threshold=130.0, gaussian=0, gradient=0, measure=0)
        p.update(kw)
        n, ny, nx = frames.shape
        dtype = 0 if frames.dtype == np.uint8 else 1
        if dtype == 0:
            d, fr = self.upload_frames_u8(frames)
        else:
            d = self.to_dev(np.ascontiguousarray(frames))
            fr = self.frames(d, n, nx, ny, dtype)
        cap = cap or nx * ny // 4 + 16
        out = self.empty((n, cap, 3), np.float32); cnt = self.empty((n,), np.int64)
        self.check(self.lib.imgfd_harris_dev(self.ctx, C.byref(fr), p["k"], p["sigma_d"], p["sigma_i"], p["threshold"],
                                             p["gaussian"], p["gradient"], p["measure"], self.ptr(out), cap,
                                             self.ptr(cnt)), "imgfd_harris_dev")
        self.sync()
        cnt = self.to_host(cnt); out = self.to_host(out)
        return [out[f, :min(int(cnt[f]), cap)] for f in range(n)], cnt

    def fast9_dev(self, frames, threshold, nonmax=False, cap=None):
        n, ny, nx = frames.shape
        d, fr = self.upload_frames_u8(frames)
        cap = cap or (nx * ny // 2 + 16)
        out = self.empty((n, cap, 2), np.int32); cnt = self.empty((n,), np.int64)
        self.check(self.lib.imgfd_fast9_dev(self.ctx, C.byref(fr), threshold & 0xFF, int(nonmax), self.ptr(out), cap,
                                            self.ptr(cnt)), "imgfd_fast9_dev")
        self.sync()
        cnt = self.to_host(cnt); out = self.to_host(out)
        return [out[f, :min(int(cnt[f]), cap)] for f in range(n)], cnt

    def canny_dev(self, frames, s=2.0, low_thr=3.0, high_thr=10.0, accGrad=True):
        n, ny, nx = frames.shape
        d, fr = self.upload_frames_u8(frames)
        edges = self.empty((n, ny, nx), np.uint8); cnt = self.empty((n,), np.int64)
        self.check(self.lib.imgfd_canny_dev(self.ctx, C.byref(fr), s, low_thr, high_thr, int(accGrad),
                                            self.ptr(edges), self.ptr(cnt)), "imgfd_canny_dev")
        self.sync()
        return self.to_host(edges), self.to_host(cnt)

    def sync(self):
        self.check(self.lib.imgfd_ctx_sync(self.ctx), "sync")


class EmuBackend(_Base):
    name = "emu"

    def __init__(self):
        path = os.environ.get("IMGFD_EMU_LIB")   # e.g. a sanitizer build of the same sources (scripts/asan_emu.sh)
        if not path:
            path = os.path.join(ROOT, "tests", "hipemu", "libimgfd_emu.so")
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "image_amd", "csrc"), "emu"])
        super().__init__(_binding.bind(C.CDLL(path), strict=False))

    def to_dev(self, a):
        return np.ascontiguousarray(a).copy()

    def empty(self, shape, dtype):
        return np.zeros(shape, dtype)

    def ptr(self, a):
        return a.ctypes.data_as(C.c_void_p)

    def to_host(self, a):
        return a


class GpuBackend(_Base):
    name = "gpu"

    def __init__(self, stream=None):
        import torch
        assert torch.cuda.is_available(), "GPU tests need a device"
        from image_amd import _lib
        self.torch = torch
        super().__init__(_lib.load(), stream)

    def to_dev(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")

    def empty(self, shape, dtype):
        return self.torch.zeros(shape, dtype=getattr(self.torch, np.dtype(dtype).name), device="cuda:0")

    def ptr(self, t):
        self.torch.cuda.synchronize()
        return C.c_void_p(t.data_ptr())

    def to_host(self, t):
        self.sync()
        return t.cpu().numpy()
