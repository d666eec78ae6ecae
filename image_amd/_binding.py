"""ctypes signatures for the C ABI declared in include/imgfd.h.

``bind(cdll)`` attaches argtypes/restypes to a loaded library.  The product loads
``image_amd/libimgfd.so`` through :mod:`image_amd._lib`; the CPU-only kernel-logic tests bind the
same signatures onto the host-emulator build of the same sources (tests/hipemu).
"""
from __future__ import annotations

import ctypes as C

c_void_pp = C.POINTER(C.c_void_p)


class Corner(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("R", C.c_float)]


class Corners(C.Structure):
    _fields_ = [("corners", C.POINTER(Corner)), ("n", C.c_int64), ("stage_seconds", C.c_double * 7)]


class Point(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int)]


class Points(C.Structure):
    _fields_ = [("points", C.POINTER(Point)), ("n", C.c_int64)]


class SurfOut(C.Structure):
    _fields_ = [("n", C.c_int64)] + [(k, C.POINTER(C.c_double)) for k in
                                     ("x", "y", "angle", "pyramid_scale", "score", "laplacian", "surf", "data")]


class Frames(C.Structure):
    _fields_ = [("d_frames", C.c_void_p), ("n_frames", C.c_int), ("nx", C.c_int), ("ny", C.c_int),
                ("frame_stride_bytes", C.c_size_t), ("row_stride_bytes", C.c_int), ("dtype", C.c_int)]


class StreamParams(C.Structure):
    _fields_ = [("harris", C.c_int), ("fast9", C.c_int), ("canny", C.c_int),
                ("k", C.c_float), ("sigma_d", C.c_float), ("sigma_i", C.c_float), ("threshold", C.c_float),
                ("gaussian", C.c_int), ("gradient", C.c_int), ("measure", C.c_int),
                ("fast9_threshold", C.c_int), ("suppress_non_max", C.c_int),
                ("s", C.c_double), ("low_thr", C.c_double), ("high_thr", C.c_double), ("accGrad", C.c_int),
                ("corner_cap", C.c_int64), ("point_cap", C.c_int64), ("keep_edges", C.c_int)]


class StreamResult(C.Structure):
    _fields_ = [("n_frames", C.c_int), ("first_frame", C.c_int64),
                ("harris_counts", C.POINTER(C.c_int64)), ("fast9_counts", C.POINTER(C.c_int64)),
                ("canny_counts", C.POINTER(C.c_int64)),
                ("corners", C.POINTER(Corner)), ("points", C.POINTER(Point)), ("edges", C.POINTER(C.c_uint8))]


# every symbol include/imgfd.h declares: (restype, argtypes)
c_float_pp = C.POINTER(C.POINTER(C.c_float))
c_int_p = C.POINTER(C.c_int)

SIGNATURES = {
    "imgfd_version": (C.c_int, []),
    "imgfd_harris_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float] + [C.c_int] * 9 + [C.c_void_p]),
    "imgfd_fast9_i32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_ubyte, C.c_int, C.c_void_p]),
    "imgfd_canny_i32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p, C.POINTER(C.c_int64)]),
    "imgfd_canny_f64out": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p, C.POINTER(C.c_int64)]),
    "imgfd_fhog_f64out": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, c_int_p, c_int_p]),
    "imgfd_fhog_i32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_float_pp, c_int_p, c_int_p]),
    "imgfd_surf_i32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_double, C.POINTER(SurfOut)]),
    "imgfd_surf": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_double, C.POINTER(SurfOut)]),
    "imgfd_surf_interest_points": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "imgfd_surf_points_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_double, C.c_void_p, C.c_int64, C.c_void_p]),
    "imgfd_surf_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_long, C.c_double, C.c_void_p, C.c_int64, C.c_void_p]),
    "imgfd_surf_dev_redo": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_long, C.c_double, C.c_void_p, C.c_int64, C.c_void_p, c_int_p]),
    "imgfd_knn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "imgfd_knn_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "imgfd_k_surf_integral": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "imgfd_fhog": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_float_pp, c_int_p, c_int_p]),
    "imgfd_fhog_size": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_int_p, c_int_p]),
    "imgfd_fhog_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "imgfd_ctx_create": (C.c_int, [C.c_int, c_void_pp]),
    "imgfd_ctx_create_on_stream": (C.c_int, [C.c_int, C.c_void_p, c_void_pp]),
    "imgfd_ctx_destroy": (None, [C.c_void_p]),
    "imgfd_last_error": (C.c_char_p, [C.c_void_p]),
    "imgfd_ctx_stream": (C.c_void_p, [C.c_void_p]),
    "imgfd_ctx_sync": (C.c_int, [C.c_void_p]),
    "imgfd_free": (None, [C.c_void_p]),
    "imgfd_device_count": (C.c_int, [c_int_p]),
    "imgfd_set_fir_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "imgfd_set_tuning": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "imgfd_get_counter": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]),
    "imgfd_k_fhog_lut": (C.c_int, [C.c_void_p, C.c_void_p]),
    "imgfd_k_fhog_lut_arith": (C.c_int, [C.c_void_p, C.c_void_p]),
    "imgfd_harris": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                               C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_int, C.c_int, C.POINTER(Corners)]),
    "imgfd_fast9": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint8, C.c_int,
                              C.POINTER(Points)]),
    "imgfd_canny": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                              C.c_int, C.c_void_p, C.POINTER(C.c_int64)]),
    "imgfd_harris_dev": (C.c_int, [C.c_void_p, C.POINTER(Frames), C.c_float, C.c_float, C.c_float, C.c_float,
                                   C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "imgfd_fast9_dev": (C.c_int, [C.c_void_p, C.POINTER(Frames), C.c_uint8, C.c_int, C.c_void_p, C.c_int64,
                                  C.c_void_p]),
    "imgfd_canny_dev": (C.c_int, [C.c_void_p, C.POINTER(Frames), C.c_double, C.c_double, C.c_double, C.c_int,
                                  C.c_void_p, C.c_void_p]),
    "imgfd_k_gaussian": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int]),
    "imgfd_k_gradient": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "imgfd_k_gauss_grad_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int]),
    "imgfd_k_structure_tensor": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int, C.c_int, C.c_float, C.c_int]),
    "imgfd_k_response": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                   C.c_int, C.c_float]),
    "imgfd_k_nms": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_int64,
                              C.c_void_p]),
    "imgfd_k_nms_quads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_int64,
                                    C.c_void_p]),
    "imgfd_time_structure_tensor": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                                              C.POINTER(C.c_double)]),
    "imgfd_time_structure_tensor_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                                                    C.POINTER(C.c_double)]),
    "imgfd_tensor_kernel_name": (C.c_char_p, [C.c_void_p]),
    "imgfd_k_tensor_response": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float]),
    "imgfd_profile_k3": (C.c_int, [C.c_void_p, C.c_int]),
    "imgfd_clock_probe": (C.c_int, [C.c_void_p, C.c_int]),
    "imgfd_clock_probe_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "imgfd_profile_k3_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "imgfd_synth_frames": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_uint32,
                                     C.c_void_p, C.c_int]),
    "imgfd_detect_dev": (C.c_int, [C.c_void_p, C.POINTER(Frames), C.POINTER(StreamParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "imgfd_stream_default_params": (None, [C.POINTER(StreamParams)]),
    "imgfd_stream_open": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(StreamParams), c_void_pp]),
    "imgfd_stream_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t]),
    "imgfd_stream_collect": (C.c_int, [C.c_void_p, C.POINTER(StreamResult)]),
    "imgfd_stream_close": (None, [C.c_void_p]),
    "imgfd_host_alloc": (C.c_void_p, [C.c_size_t]),
    "imgfd_host_free": (None, [C.c_void_p]),
}


def bind(lib: C.CDLL, strict: bool = True) -> C.CDLL:
    missing = []
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing and strict:
        raise ImportError(f"{lib._name} does not export: {', '.join(missing)}")
    return lib


class ImgfdError(RuntimeError):
    pass


def check(lib: C.CDLL, ctx, status: int, what: str) -> None:
    if status != 0:
        msg = lib.imgfd_last_error(ctx) if ctx else b""
        raise ImgfdError(f"{what} failed with imgfd_status {status}: {(msg or b'').decode(errors='replace')}")
