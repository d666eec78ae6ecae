"""Loader of the product library ``image_amd/libimgfd.so`` (hipcc, gfx950).

There is no CPU fallback: if the shared library is missing or exports less than include/imgfd.h
declares, importing the API raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``
(or ``make -C image_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

from . import _binding

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libimgfd.so")

_lib = None
_lock = threading.Lock()


def load() -> C.CDLL:
    """Load and bind libimgfd.so once.  Raises ImportError when it has not been built."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    f"{LIB_PATH} is missing: the HIP backend has not been built "
                    "(run `make -C image_amd/csrc` or __graft_entry__.build()); image_amd has no CPU fallback")
            try:  # share torch's HIP runtime when torch is in the process (same libamdhip64 soname)
                import torch  # noqa: F401
            except Exception:  # pragma: no cover - torch is optional for the host-pointer API
                pass
            _lib = _binding.bind(C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL))
        return _lib


class Context:
    """RAII wrapper of ``imgfd_ctx`` (one HIP stream + workspace)."""

    def __init__(self, device: int = 0, stream: int | None = None, lib: C.CDLL | None = None):
        self.lib = lib if lib is not None else load()
        self.handle = C.c_void_p()
        if stream is None:
            st = self.lib.imgfd_ctx_create(device, C.byref(self.handle))
        else:
            st = self.lib.imgfd_ctx_create_on_stream(device, C.c_void_p(stream), C.byref(self.handle))
        if st != 0:
            raise _binding.ImgfdError(
                f"imgfd_ctx_create(device={device}) failed with status {st} "
                "(2 = no usable HIP device: this backend needs an MI355X/gfx950 GPU)")
        self.device = device

    def check(self, status: int, what: str) -> None:
        _binding.check(self.lib, self.handle, status, what)

    def sync(self) -> None:
        self.check(self.lib.imgfd_ctx_sync(self.handle), "imgfd_ctx_sync")

    def set_fir_mode(self, mode: int) -> None:
        self.check(self.lib.imgfd_set_fir_mode(self.handle, int(mode)), "imgfd_set_fir_mode")

    def close(self) -> None:
        if self.handle:
            self.lib.imgfd_ctx_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


_default_ctx: dict = {}


def default_context(device: int = 0) -> Context:
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
