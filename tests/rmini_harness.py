"""TEST INFRASTRUCTURE: runs r/<pkg>/src/glue.c without R.  tests/rmini/rmini.c implements the handful of R C-API entry
points the glue uses; this module builds rmini.c + one package's glue.c + the imgfd library under test into one shared
object and drives it the way R's .Call does (registered name -> address + arity -> call with SEXP arguments)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INTSXP, LGLSXP, REALSXP, VECSXP, NILSXP = 13, 10, 14, 19, 0


class RError(RuntimeError):
    """what R would raise as a condition: the routine called Rf_error()"""


class RPackage:
    def __init__(self, pkg: str, imgfd_lib_path: str, tag: str):
        self.pkg = pkg
        out_dir = os.path.join(ROOT, "tests", "rmini", "build")
        os.makedirs(out_dir, exist_ok=True)
        so = os.path.join(out_dir, f"{pkg}_{tag}.so")
        srcs = [os.path.join(ROOT, "tests", "rmini", "rmini.c"), os.path.join(ROOT, "r", pkg, "src", "glue.c")]
        deps = srcs + [os.path.join(ROOT, "r", "imgfd_glue.h"), os.path.join(ROOT, "include", "imgfd.h"),
                       os.path.join(ROOT, "r", "stub", "Rinternals.h")]
        if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
            tmp = f"{so}.{os.getpid()}.tmp"   # (xdist workers may build the same object)
            subprocess.check_call(["gcc", "-O1", "-g", "-Wall", "-Wextra", "-Werror", "-Wno-cast-function-type", "-shared", "-fPIC",
                                   "-I", os.path.join(ROOT, "r", "stub"), "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "r"), *srcs,
                                   imgfd_lib_path, f"-Wl,-rpath,{os.path.dirname(imgfd_lib_path)}", "-lm", "-o", tmp])
            os.replace(tmp, so)
        # DEEPBIND: the glue's imgfd_* references bind to the library it was linked with, whatever else the process has loaded
        # globally by then (a session that touched the product library first would otherwise hand the emulator's glue the
        # product's imgfd_ctx_create)
        self.dll = d = C.CDLL(so, mode=os.RTLD_NOW | os.RTLD_LOCAL | os.RTLD_DEEPBIND)
        vp = C.c_void_p
        for name, res, args in [("rmini_int_vector", vp, [vp, C.c_long]), ("rmini_real_vector", vp, [vp, C.c_long]),
                                ("rmini_logical", vp, [C.c_int]), ("rmini_type", C.c_uint, [vp]), ("rmini_length", C.c_long, [vp]),
                                ("rmini_nrow", C.c_int, [vp]), ("rmini_ncol", C.c_int, [vp]), ("rmini_data", vp, [vp]),
                                ("rmini_elt", vp, [vp, C.c_long]), ("rmini_name", C.c_char_p, [vp, C.c_long]),
                                ("rmini_last_error", C.c_char_p, []), ("rmini_printed", C.c_char_p, []),
                                ("rmini_clear_printed", None, []), ("rmini_protect_depth", C.c_int, []),
                                ("rmini_protect_max", C.c_int, []), ("rmini_dynamic_symbols", C.c_int, []),
                                ("rmini_lookup", vp, [C.c_char_p, C.POINTER(C.c_int)]),
                                ("rmini_call", vp, [vp, C.c_int, C.POINTER(vp)])]:
            f = getattr(d, name); f.restype = res; f.argtypes = args
        init = getattr(d, "R_init_" + pkg.replace(".", "_"))   # what R calls when it loads the package's DLL
        init.restype = None; init.argtypes = [vp]
        init(None)

    def unload(self):
        f = getattr(self.dll, "R_unload_" + self.pkg.replace(".", "_"))
        f.restype = None; f.argtypes = [C.c_void_p]
        f(None)

    # ---- R values
    def integer(self, v):
        a = np.ascontiguousarray(np.atleast_1d(v), np.int32)
        return self.dll.rmini_int_vector(a.ctypes.data_as(C.c_void_p), a.size)

    def numeric(self, v):
        a = np.ascontiguousarray(np.atleast_1d(v), np.float64)
        return self.dll.rmini_real_vector(a.ctypes.data_as(C.c_void_p), a.size)

    def logical(self, v):
        return self.dll.rmini_logical(int(bool(v)))

    def to_py(self, s):
        d = self.dll
        t, n = d.rmini_type(s), d.rmini_length(s)
        if t == NILSXP:
            return None
        if t in (REALSXP, INTSXP, LGLSXP):
            ct = C.c_double if t == REALSXP else C.c_int
            a = np.ctypeslib.as_array(C.cast(d.rmini_data(s), C.POINTER(ct)), shape=(n,)).copy() if n else np.zeros((0,), ct)
            if d.rmini_nrow(s) or d.rmini_ncol(s):
                a = a.reshape((d.rmini_nrow(s), d.rmini_ncol(s)), order="F")   # R matrices are column-major
            return a.astype(bool) if t == LGLSXP else a
        if t == VECSXP:
            vals = [self.to_py(d.rmini_elt(s, i)) for i in range(n)]
            names = [d.rmini_name(s, i) for i in range(n)]
            return {k.decode(): v for k, v in zip(names, vals)} if n and names[0] is not None else vals
        raise TypeError(f"SEXP type {t}")

    # ---- .Call
    def call(self, name: str, *args):
        d = self.dll
        k = C.c_int(0)
        fn = d.rmini_lookup(name.encode(), C.byref(k))
        assert fn, f"{name} is not in the table R_init_{self.pkg.replace('.', '_')} registered"
        assert k.value == len(args), f".Call: {name} is registered with {k.value} arguments, got {len(args)}"
        argv = (C.c_void_p * len(args))(*args)
        res = d.rmini_call(fn, len(args), argv)
        if not res:
            raise RError(d.rmini_last_error().decode())
        assert d.rmini_protect_depth() == 0, f"{name}: PROTECT/UNPROTECT unbalanced by {d.rmini_protect_depth()}"
        return self.to_py(res)

    def printed(self):
        s = self.dll.rmini_printed().decode()
        self.dll.rmini_clear_printed()
        return s
