"""TEST INFRASTRUCTURE: drive qmri_pnp_recon_poc_amd/mex/qmri_mex.cpp -- the MATLAB gateway, unchanged -- without MATLAB.

The gateway is compiled together with tests/cpp/mex_mock.cpp (a small stand-in for the MATLAB runtime's C API, declared in tests/stubs/mex.h)
into one shared library linked against libqmri.so; `qmri_mex(cmd, *args, nargout=n)` then makes the call a MATLAB session would make --
numpy arrays go in and come out as mxArrays with MATLAB's memory layout (column-major, interleaved complex), a Python dict of scalars is a
1 x 1 struct, a raised mexErrMsgIdAndTxt becomes MexError(id, message)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "qmri_pnp_recon_poc_amd")
OUT = os.path.join(ROOT, "tests", "cpp", "_build", "libqmri_mex_mock.so")
SRCS = [os.path.join(PKG, "mex", "qmri_mex.cpp"), os.path.join(ROOT, "tests", "cpp", "mex_mock.cpp")]
_CLS = {6: np.float64, 7: np.float32, 12: np.int32}          # mxClassID of tests/stubs/mex.h
_ID = {np.dtype(np.float64): 6, np.dtype(np.complex128): 6, np.dtype(np.float32): 7, np.dtype(np.complex64): 7, np.dtype(np.int32): 12}


class MexError(RuntimeError):
    def __init__(self, ident, msg):
        super().__init__(f"{ident}: {msg}")
        self.id, self.msg = ident, msg


def build() -> str:
    from qmri_pnp_recon_poc_amd import _lib
    _lib.lib()                                                # libqmri.so exists (built if missing)
    deps = SRCS + [os.path.join(ROOT, "tests", "stubs", "mex.h"), os.path.join(ROOT, "include", "qmri.h")]
    if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-shared", "-fPIC", "-Wall", "-I", os.path.join(ROOT, "tests", "stubs"), "-I", os.path.join(ROOT, "include")]
                       + SRCS + ["-L", PKG, "-lqmri", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib", "-o", OUT], check=True)
    return OUT


_L = None


def lib():
    global _L
    if _L is None:
        L = C.CDLL(build())
        vp = C.c_void_p
        for name, res, args in (("mxCreateNumericArray", vp, [C.c_size_t, C.POINTER(C.c_size_t), C.c_int, C.c_int]), ("mxGetData", vp, [vp]),
                                ("mxGetDimensions", C.POINTER(C.c_size_t), [vp]), ("mxGetNumberOfDimensions", C.c_size_t, [vp]), ("mxIsComplex", C.c_bool, [vp]),
                                ("mxDestroyArray", None, [vp]), ("mock_string", vp, [C.c_char_p]), ("mock_struct", vp, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double)]),
                                ("mock_class", C.c_int, [vp]), ("mock_is_struct", C.c_int, [vp]), ("mock_nfields", C.c_int, [vp]), ("mock_field_name", C.c_char_p, [vp, C.c_int]),
                                ("mock_field_value", vp, [vp, C.c_int]), ("mock_call", C.c_int, [C.c_int, C.POINTER(vp), C.c_int, C.POINTER(vp)]),
                                ("mock_error_id", C.c_char_p, []), ("mock_error_msg", C.c_char_p, []), ("mock_exit", None, [])):
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _L = L
    return _L


def to_mx(a):
    """Python value -> mxArray*: str -> char row, dict of numbers -> 1 x 1 struct, scalar -> double scalar, ndarray -> numeric array (MATLAB layout)."""
    L = lib()
    if isinstance(a, str):
        return L.mock_string(a.encode())
    if isinstance(a, dict):
        names = (C.c_char_p * len(a))(*[k.encode() for k in a])
        vals = (C.c_double * len(a))(*[float(v) for v in a.values()])
        return L.mock_struct(len(a), names, vals)
    a = np.asarray(a)
    if a.dtype == np.int64 or a.dtype == np.bool_:
        a = a.astype(np.float64)                              # (MATLAB's default numeric class)
    if a.dtype not in _ID:
        raise TypeError(f"no mxArray class for dtype {a.dtype}")
    shape = a.shape if a.ndim >= 2 else ((1, 1) if a.ndim == 0 else (a.shape[0], 1) if a.size else (0, 0))
    dims = (C.c_size_t * len(shape))(*shape)
    m = L.mxCreateNumericArray(len(shape), dims, _ID[a.dtype], int(np.iscomplexobj(a)))
    if a.size:
        flat = np.ascontiguousarray(a.reshape(shape).ravel(order="F"))
        C.memmove(L.mxGetData(m), flat.ctypes.data, flat.nbytes)
    return m


def from_mx(m):
    L = lib()
    if not m:
        return None
    if L.mock_is_struct(m):
        return {L.mock_field_name(m, k).decode(): from_mx(L.mock_field_value(m, k)) for k in range(L.mock_nfields(m))}
    nd = L.mxGetNumberOfDimensions(m)
    d = L.mxGetDimensions(m)
    shape = tuple(int(d[i]) for i in range(nd))
    base = _CLS[L.mock_class(m)]
    cplx = bool(L.mxIsComplex(m))
    n = int(np.prod(shape))
    dt = np.dtype(base)
    if cplx:
        dt = np.dtype(np.complex128 if base is np.float64 else np.complex64)
    if n == 0:
        return np.zeros(shape, dt)
    buf = (C.c_char * (n * dt.itemsize)).from_address(L.mxGetData(m))
    return np.frombuffer(buf, dtype=dt, count=n).copy().reshape(shape, order="F")


def qmri_mex(cmd, *args, nargout=0):
    """outputs = qmri_mex(cmd, args...)  -- what `[o1, ..., on] = qmri_mex('cmd', ...)` does in MATLAB (n = nargout)."""
    L = lib()
    prhs = [to_mx(cmd)] + [to_mx(a) for a in args]
    pr = (C.c_void_p * len(prhs))(*prhs)
    pl = (C.c_void_p * max(nargout, 1))()
    rc = L.mock_call(nargout, pl, len(prhs), pr)
    try:
        if rc:
            raise MexError(L.mock_error_id().decode(), L.mock_error_msg().decode())
        outs = [from_mx(pl[i]) for i in range(max(nargout, 1))]
    finally:
        for p in prhs:
            L.mxDestroyArray(p)
        for i in range(max(nargout, 1)):
            if pl[i]:
                L.mxDestroyArray(pl[i])
    if nargout <= 1:
        return outs[0]
    return tuple(outs[:nargout])


def mex_exit():
    """MATLAB quitting: runs the gateway's mexAtExit handler (releases the context and the kept arrays)."""
    lib().mock_exit()
