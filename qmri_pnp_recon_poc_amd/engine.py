"""numpy-facing wrapper of one libqmri context (include/qmri.h).

Arrays follow MATLAB conventions: `X[h, w, c]` (any memory order; copied into column-major interleaved-complex
buffers at the boundary), measurement vectors frame-major.  Every failure raises `QmriError` carrying the
library's message -- the Python analogue of the MATLAB exceptions the reference's plugins throw.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import AdmmParams, LrtvInfo, LrtvParams, NetDesc, Profile

ARCH_UNETRES, ARCH_SEQ_CONV = 0, 1
SOLVER_LSQR, SOLVER_DIRECT = 0, 1


class QmriError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libqmri error {code}: {msg}")
        self.code = code


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _cbuf(a):
    """complex array -> 1-D complex128 in column-major (MATLAB) element order."""
    return np.ascontiguousarray(np.asarray(a, dtype=np.complex128).ravel(order="F"))


def real_dictionary_array(a, name: str, dtype):
    """`dict.D` / `dict.V` as the real array the kernels take.  MATLAB may hold them complex-typed (the reference takes
    real(dict.V), main_recon_tsmis_FFT.m:129, and real(dict.D(...)), main_synthesize_tsmis.m:89, explicitly, while
    mrf_dtm_cpu.m:91 multiplies by dict.D as stored): a complex-typed array whose imaginary part is zero is accepted, one
    with a non-zero imaginary part is refused -- it is never silently truncated."""
    a = np.asarray(a)
    if np.iscomplexobj(a):
        if np.any(a.imag != 0):
            raise ValueError(f"{name} is complex with a non-zero imaginary part: the dictionary match implements real atoms "
                             f"(dict.D real, as in the reference's real_fisp dictionaries); pass real({name}) if that is what is meant")
        a = a.real
    return np.asarray(a, dtype=dtype)


def build_spiral(N: int, S: int, T: int):
    """setup_subsampling_spiralgrided.m:7-34 -> (frame_ptr[T+1], kidx[m]) int32 (0-based column-major k)."""
    L = _lib.lib()
    fp = np.zeros(T + 1, np.int32)
    k = np.zeros(S * T, np.int32)
    m = C.c_int(0)
    st = L.qmri_build_spiral(None, N, S, T, fp.ctypes.data_as(C.POINTER(C.c_int32)), k.ctypes.data_as(C.POINTER(C.c_int32)), k.size, C.byref(m))
    if st != 0:
        raise QmriError(st, L.qmri_last_error(None).decode())
    return fp, k[: m.value].copy()


def build_epi(N: int, M: int, percentage: float, T: int):
    """setup_subsampling_epi.m:20-33 -> (frame_ptr[T+1], kidx[m])."""
    L = _lib.lib()
    step = int(np.floor(1.0 / percentage + 0.5))
    cap = max((N // step) * M * T, 1)
    fp = np.zeros(T + 1, np.int32)
    k = np.zeros(cap, np.int32)
    m = C.c_int(0)
    st = L.qmri_build_epi(None, N, M, float(percentage), T, fp.ctypes.data_as(C.POINTER(C.c_int32)), k.ctypes.data_as(C.POINTER(C.c_int32)), cap, C.byref(m))
    if st != 0:
        raise QmriError(st, L.qmri_last_error(None).decode())
    return fp, k[: m.value].copy()


def read_onnx_unetres(path):
    """Weights of the ONNX file main_recon_tsmis_FFT.m:138 imports, through the library's own reader
    (qmri_onnx_read_unetres) -> (flat fp32 weights, dict(in_nc, out_nc, nc, nb))."""
    L = _lib.lib()
    d, n = NetDesc(), C.c_size_t(0)
    st = L.qmri_onnx_read_unetres(str(path).encode(), C.byref(d), None, 0, C.byref(n))
    if st != 0:
        raise QmriError(st, L.qmri_last_error(None).decode())
    w = np.empty(n.value, np.float32)
    st = L.qmri_onnx_read_unetres(str(path).encode(), C.byref(d), w.ctypes.data_as(C.POINTER(C.c_float)), w.size, C.byref(n))
    if st != 0:
        raise QmriError(st, L.qmri_last_error(None).decode())
    return w, {"in_nc": int(d.in_nc), "out_nc": int(d.out_nc), "nc": tuple(int(v) for v in d.nc), "nb": int(d.nb)}


class Engine:
    """One device context: operator + denoiser + dictionary + workspaces."""

    def __init__(self, device: int = 0):
        self.L = _lib.lib()
        h = C.c_void_p()
        st = self.L.qmri_create(int(device), C.byref(h))
        if st != 0:
            raise QmriError(st, self.L.qmri_last_error(None).decode())
        self.h = h
        self.device = device
        self.N = self.M = self.s = self.T = self.m = 0
        self.net_desc = None
        self.dict_shape = None

    def close(self):
        if getattr(self, "h", None):
            self.L.qmri_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != 0:
            raise QmriError(st, self.L.qmri_last_error(self.h).decode())

    # -- stream / sync ---------------------------------------------------------------------------
    def set_stream(self, hip_stream_ptr: int | None):
        self._check(self.L.qmri_set_stream(self.h, C.c_void_p(hip_stream_ptr) if hip_stream_ptr else None))

    def synchronize(self):
        self._check(self.L.qmri_synchronize(self.h))

    # -- operator ----------------------------------------------------------------------------------
    def set_operator(self, N, M, V, frame_ptr, kidx, max_batch=1):
        V = real_dictionary_array(V, "V", np.float64)
        if V.ndim != 2:
            raise ValueError("V must be T x s")
        T, s = V.shape
        Vf = np.ascontiguousarray(V.ravel(order="F"))
        fp = np.ascontiguousarray(frame_ptr, dtype=np.int32)
        k = np.ascontiguousarray(kidx, dtype=np.int32)
        self._check(self.L.qmri_set_operator(self.h, int(N), int(M), int(s), int(T), Vf.ctypes.data_as(C.POINTER(C.c_double)),
                                             fp.ctypes.data_as(C.POINTER(C.c_int32)), k.ctypes.data_as(C.POINTER(C.c_int32)), int(max_batch)))
        self.N, self.M, self.s, self.T, self.m = int(N), int(M), int(s), int(T), int(fp[-1])

    def forward(self, x):
        """y = F.forward(x)  (main_recon_tsmis_FFT.m:228).  x: [N,M,s] real or complex; single-precision input (float32 /
        complex64, a MATLAB `single` array) goes through the _f32 entry point and comes back as complex64."""
        x = np.asarray(x)
        if x.shape != (self.N, self.M, self.s):
            raise ValueError(f"x must be {self.N}x{self.M}x{self.s}")
        if x.dtype in (np.float32, np.complex64):
            cx = np.iscomplexobj(x)
            xb = np.ascontiguousarray(x.ravel(order="F"))
            y = np.empty(self.m, np.complex64)
            fp = C.POINTER(C.c_float)
            self._check(self.L.qmri_forward_f32(self.h, xb.view(np.float32).ctypes.data_as(fp), int(cx), y.view(np.float32).ctypes.data_as(fp)))
            return y
        y = np.empty(self.m, np.complex128)
        if np.iscomplexobj(x):
            xb = _cbuf(x)
            self._check(self.L.qmri_forward(self.h, _vp(xb), 1, _vp(y)))
        else:
            xb = np.ascontiguousarray(np.asarray(x, dtype=np.float64).ravel(order="F"))
            self._check(self.L.qmri_forward(self.h, _vp(xb), 0, _vp(y)))
        return y

    def adjoint(self, y):
        """x = F.adjoint(y)  (main_recon_tsmis_FFT.m:229); complex64 in -> complex64 out (the _f32 entry point)."""
        if np.asarray(y).dtype == np.complex64:
            y32 = np.ascontiguousarray(np.asarray(y).ravel(order="F"))
            if y32.size != self.m:
                raise ValueError(f"y must have {self.m} elements")
            x = np.empty(self.N * self.M * self.s, np.complex64)
            fp = C.POINTER(C.c_float)
            self._check(self.L.qmri_adjoint_f32(self.h, y32.view(np.float32).ctypes.data_as(fp), x.view(np.float32).ctypes.data_as(fp)))
            return x.reshape((self.N, self.M, self.s), order="F")
        yb = _cbuf(y)
        if yb.size != self.m:
            raise ValueError(f"y must have {self.m} elements")
        x = np.empty(self.N * self.M * self.s, np.complex128)
        self._check(self.L.qmri_adjoint(self.h, _vp(yb), _vp(x)))
        return x.reshape((self.N, self.M, self.s), order="F")

    # -- multi-coil extension (BASELINE configs[4]; no reference counterpart: README.md:63, single coil) -------------------
    def set_coils(self, maps):
        """maps: [N, M, ncoil] complex coil sensitivities (None clears them)."""
        if maps is None:
            self._check(self.L.qmri_set_coils(self.h, 0, None))
            self.ncoil = 0
            return
        maps = np.asarray(maps)
        if maps.ndim != 3 or maps.shape[:2] != (self.N, self.M):
            raise ValueError(f"maps must be {self.N}x{self.M}xncoil")
        mb = _cbuf(maps)
        self._check(self.L.qmri_set_coils(self.h, int(maps.shape[2]), _vp(mb)))
        self.ncoil = int(maps.shape[2])

    def forward_mc(self, x):
        """y[:, j] = F.forward(maps[..., j] * x): [m, ncoil] complex."""
        x = np.asarray(x)
        if x.shape != (self.N, self.M, self.s):
            raise ValueError(f"x must be {self.N}x{self.M}x{self.s}")
        y = np.empty(self.m * max(getattr(self, "ncoil", 0), 1), np.complex128)
        xb = _cbuf(x)
        self._check(self.L.qmri_forward_mc(self.h, _vp(xb), 1, _vp(y)))
        return y.reshape((self.m, -1), order="F")

    def adjoint_mc(self, y):
        """x = sum_j conj(maps[..., j]) * F.adjoint(y[:, j])."""
        yb = _cbuf(y)
        nc = getattr(self, "ncoil", 0)
        if nc and yb.size != self.m * nc:                        # (no maps set: the library says so, QMRI_ERR_STATE)
            raise ValueError(f"y must be {self.m} x {nc}")
        x = np.empty(self.N * self.M * self.s, np.complex128)
        self._check(self.L.qmri_adjoint_mc(self.h, _vp(yb), _vp(x)))
        return x.reshape((self.N, self.M, self.s), order="F")

    def xupdate_mc(self, y_mc, z, r, tol=1e-4, maxit=100, x0=None):
        """Multi-coil extension (no reference counterpart): x = lsqr(afun_mc, [y_mc; sqrt(r) z], tol, maxit, [], [], x0).  y_mc [m, ncoil].  Returns (x, iters, flag)."""
        yb, zb = _cbuf(y_mc), _cbuf(z)
        nc = getattr(self, "ncoil", 0)
        if nc and yb.size != self.m * nc:
            raise ValueError(f"y_mc must be {self.m} x {nc}")
        x0b = _cbuf(x0) if x0 is not None else None
        x = np.empty(self.N * self.M * self.s, np.complex128)
        it, fl = C.c_int32(0), C.c_int32(0)
        self._check(self.L.qmri_xupdate_mc(self.h, _vp(yb), _vp(zb), float(r), float(tol), int(maxit), _vp(x0b), _vp(x), C.byref(it), C.byref(fl)))
        return x.reshape((self.N, self.M, self.s), order="F"), it.value, fl.value

    def pnp_admm_mc(self, y_mc, gamma=0.05, iters=100, cg_tol=1e-4, cg_maxit=100, multi_level=False, noise_std=0.01, x0=None):
        """Multi-coil extension: PnP_ADMM(y, param) with F replaced by the SENSE operator of set_coils.  Returns (x, lsqr_iters)."""
        p = AdmmParams(float(gamma), int(iters), float(cg_tol), int(cg_maxit), SOLVER_LSQR, int(bool(multi_level)), float(noise_std), 0)
        yb = _cbuf(y_mc)
        nc = getattr(self, "ncoil", 0)
        if nc and yb.size != self.m * nc:
            raise ValueError(f"y_mc must be {self.m} x {nc}")
        x0b = _cbuf(x0) if x0 is not None else None
        x = np.empty(self.N * self.M * self.s, np.complex128)
        li = np.zeros(max(iters, 1), np.int32)
        self._check(self.L.qmri_pnp_admm_mc(self.h, _vp(yb), C.byref(p), _vp(x0b), _vp(x), li.ctypes.data_as(C.POINTER(C.c_int32))))
        return x.reshape((self.N, self.M, self.s), order="F"), li[:iters]

    def xupdate(self, y, z, r, tol=1e-4, maxit=100, x0=None, solver="lsqr"):
        """The x-update of PnP_ADMM.m:102 alone.  Returns (x, iters, flag)."""
        yb, zb = _cbuf(y), _cbuf(z)
        x = _cbuf(x0 if x0 is not None else np.zeros((self.N, self.M, self.s))).copy()
        it, fl = C.c_int32(0), C.c_int32(0)
        self._check(self.L.qmri_xupdate(self.h, _vp(yb), _vp(zb), float(r), float(tol), int(maxit),
                                        SOLVER_LSQR if solver == "lsqr" else SOLVER_DIRECT, _vp(x), C.byref(it), C.byref(fl)))
        return x.reshape((self.N, self.M, self.s), order="F"), it.value, fl.value

    # -- denoiser ------------------------------------------------------------------------------------
    def set_denoiser(self, weights, H, W, in_nc=10, out_nc=10, nc=(64, 128, 256, 512), nb=4, arch=ARCH_UNETRES,
                     residual_noise=False, max_batch=1):
        d = NetDesc(int(arch), int(in_nc), int(out_nc), (C.c_int32 * 4)(*[int(v) for v in nc]), int(nb), int(bool(residual_noise)))
        w = np.ascontiguousarray(weights, dtype=np.float32)
        self._check(self.L.qmri_set_denoiser(self.h, C.byref(d), w.ctypes.data_as(C.POINTER(C.c_float)), w.nbytes, int(H), int(W), int(max_batch)))
        self.net_desc = d
        self.net_hw = (int(H), int(W))

    def lsqr_persist(self, on: bool):
        """Test / A-B hook (qmri_debug_lsqr_persist): all LSQR iterations of an x-update in one launch (default) or two launches per iteration."""
        self._check(self.L.qmri_debug_lsqr_persist(self.h, 2 if on == 2 else int(bool(on))))

    def dict_filter(self, on: bool = True, margin_scale: float = 1.0):
        """Test / A-B hook (qmri_debug_dict_filter): f16 filter in front of the exact dictionary products (default on; same results)."""
        self.L.qmri_debug_dict_filter.argtypes = [C.c_void_p, C.c_int, C.c_float]
        self._check(self.L.qmri_debug_dict_filter(self.h, int(bool(on)), float(margin_scale)))

    def conv_resident(self, on=True):
        """Test / A-B hook (qmri_debug_conv_resident): the full-resolution ResBlocks as one launch with LDS-resident tiles (default) or one
        launch per layer; on = 2: a tile withholds its hand-off (recovery path).  Returns the hand-off time-outs seen so far."""
        n = C.c_int(0)
        self.L.qmri_debug_conv_resident.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        self._check(self.L.qmri_debug_conv_resident(self.h, 2 if on == 2 else int(bool(on)), C.byref(n)))
        return n.value

    def denoiser_scheme(self):
        """(scheme, fallbacks): 2 = f16 x 3 products, 3 = bf16 x 6 products; how often a run-time guard switched 2 -> 3."""
        sc, fb = C.c_int(0), C.c_int(0)
        self._check(self.L.qmri_denoiser_scheme(self.h, C.byref(sc), C.byref(fb)))
        return sc.value, fb.value

    def health(self) -> dict:
        """qmri_get_health: which self-checking fast paths are armed, how often one gave up (the work was then repeated on the slower path, same
        results), and wall clock + stage times of the most recent qmri_pnp_admm_dev call (stages: profile level 1 or 3)."""
        from ._lib import Health
        h = Health()
        self._check(self.L.qmri_get_health(self.h, C.byref(h)))
        st = list(h.last_call_stage_ms)
        return {"denoiser_scheme": {0: None, 2: "f16x3", 3: "bf16x6"}.get(h.denoiser_scheme, str(h.denoiser_scheme)),
                "denoiser_fallbacks": h.denoiser_fallbacks, "resident_tile_launch_armed": bool(h.resident_armed),
                "resident_tile_timeouts": h.resident_timeouts, "lsqr_one_launch": {-1: "undecided", 0: "off", 1: "armed"}.get(h.lsqr_one_launch, "armed"),
                "lsqr_one_launch_timeouts": h.lsqr_timeouts, "repeated_calls": h.repeated_calls,
                "last_call_wall_ms": round(h.last_call_wall_ms, 3),
                "last_call_stage_ms": {"xupdate": round(st[0], 3), "denoiser": round(st[1], 3), "elementwise": round(st[2], 3), "diagnostics": round(st[3], 3)},
                "set_denoiser_ms": {"pack_and_upload": round(h.set_denoiser_ms[0], 2), "tensors_and_buffers": round(h.set_denoiser_ms[1], 2),
                                    "calibration_probe": round(h.set_denoiser_ms[2], 2)}}

    def denoise(self, x):
        """I = denoiseImage_PnP_ADMM(x, net, true, residual_noise): x [H,W,C] or [H,W,C,B] double."""
        x = np.asarray(x, dtype=np.float64)
        squeeze = x.ndim == 3
        if squeeze:
            x = x[..., None]
        if x.ndim != 4:
            raise ValueError("input must be H x W x C (x N)")       # images:denoiseImage:invalidImageFormat
        H, W, Cc, B = x.shape
        xb = np.ascontiguousarray(x.ravel(order="F"))
        out_nc = self.net_desc.out_nc if self.net_desc is not None else 1    # unset: the library reports QMRI_ERR_STATE
        out = np.empty(H * W * out_nc * B, np.float64)
        self._check(self.L.qmri_denoise(self.h, xb.ctypes.data_as(C.POINTER(C.c_double)), H, W, Cc, B, out.ctypes.data_as(C.POINTER(C.c_double))))
        out = out.reshape((H, W, out_nc, B), order="F")
        return out[..., 0] if squeeze else out

    # -- PnP-ADMM ------------------------------------------------------------------------------------
    def pnp_admm(self, y, gamma=0.05, iters=100, cg_tol=1e-4, cg_maxit=100, solver="lsqr", multi_level=False,
                 noise_std=0.01, x0=None, gt=None, want_diag=False):
        """x = PnP_ADMM(y, param)  (PnP_ADMM.m:1).  Returns (x [N,M,s] complex, diag [iters,2] or None, lsqr_iters)."""
        p = AdmmParams(float(gamma), int(iters), float(cg_tol), int(cg_maxit), SOLVER_LSQR if solver == "lsqr" else SOLVER_DIRECT,
                       int(bool(multi_level)), float(noise_std), int(bool(want_diag)))
        yb = _cbuf(y)
        if yb.size != self.m:
            raise ValueError(f"y must have {self.m} elements")
        x0b = _cbuf(x0) if x0 is not None else None
        gtb = _cbuf(gt) if gt is not None else None
        x = np.empty(self.N * self.M * self.s, np.complex128)
        diag = np.zeros(2 * max(iters, 1), np.float64) if want_diag else None
        li = np.zeros(max(iters, 1), np.int32)
        self._check(self.L.qmri_pnp_admm(self.h, _vp(yb), C.byref(p), _vp(x0b), _vp(gtb), _vp(x),
                                         diag.ctypes.data_as(C.POINTER(C.c_double)) if diag is not None else None,
                                         li.ctypes.data_as(C.POINTER(C.c_int32))))
        return (x.reshape((self.N, self.M, self.s), order="F"),
                diag[: 2 * iters].reshape(iters, 2) if diag is not None else None, li[:iters])

    def pnp_admm_batch(self, ys, slices_per_launch=15, gamma=0.05, iters=100, cg_tol=1e-4, cg_maxit=100, solver="lsqr", multi_level=False,
                       noise_std=0.01):
        """A slice stack ys [S, m] through this context, slices_per_launch at a time (qmri_pnp_admm_batch; what `PnP_ADMM_hip(Y, param)` calls
        for a measurement matrix).  Returns (X [S,N,M,s] complex, lsqr_iters [S, iters])."""
        p = AdmmParams(float(gamma), int(iters), float(cg_tol), int(cg_maxit), SOLVER_LSQR if solver == "lsqr" else SOLVER_DIRECT,
                       int(bool(multi_level)), float(noise_std), 0)
        yb = np.ascontiguousarray(np.asarray(ys, np.complex128))
        if yb.ndim != 2 or yb.shape[1] != self.m:
            raise ValueError(f"ys must be [slices, {self.m}]")
        S, n = yb.shape[0], self.N * self.M * self.s
        x = np.empty((S, n), np.complex128)
        li = np.zeros((S, max(iters, 1)), np.int32)
        self._check(self.L.qmri_pnp_admm_batch(self.h, S, int(slices_per_launch), _vp(yb), C.byref(p), None, None, _vp(x), None,
                                               li.ctypes.data_as(C.POINTER(C.c_int32))))
        return np.stack([x[i].reshape((self.N, self.M, self.s), order="F") for i in range(S)]), li[:, :iters]

    # -- dictionary ----------------------------------------------------------------------------------
    def set_dictionary(self, D, normD, lut):
        D = real_dictionary_array(D, "dict.D", np.float32)
        lut = np.asarray(lut, dtype=np.float32)
        K, s = D.shape
        Q = lut.shape[1]
        Df = np.ascontiguousarray(D.ravel(order="F"))
        lf = np.ascontiguousarray(lut.ravel(order="F"))
        nd = np.ascontiguousarray(normD, dtype=np.float32)
        f = C.POINTER(C.c_float)
        self._check(self.L.qmri_set_dictionary(self.h, K, s, Q, Df.ctypes.data_as(f), nd.ctypes.data_as(f), lf.ctypes.data_as(f)))
        self.dict_shape = (K, s, Q)

    def synthesize_tsmi(self, qmap, mode="real"):
        """TSMI of a quantitative map (main_synthesize_tsmis.m:82-103): qmap [..., 3] (T1, T2, PD) ->
        (X, idx [...] 1-based nearest dictionary entry).  mode 'real' (:91-98): X [..., s] float32, |PD| and the sign of channel 1
        folded in; mode 'complex' (:100-103): PD may be complex, X [..., 2s] = the real parts of the s channels, then the imaginary
        ones.  Needs set_dictionary."""
        if mode not in ("real", "complex"):
            raise ValueError("mode must be 'real' or 'complex'")
        qmap = np.asarray(qmap)
        if qmap.shape[-1] != 3:
            raise ValueError("qmap must have T1, T2, PD along its last dimension")
        if getattr(self, "dict_shape", None) is None:
            raise ValueError("dictionary not set")
        if mode == "real" and np.iscomplexobj(qmap):
            qmap = np.concatenate([qmap[..., :2].real, np.abs(qmap[..., 2:3])], axis=-1)     # abs(qm(:,3)), :92
        shp = qmap.shape[:-1]
        pd_im = np.ascontiguousarray(qmap[..., 2].imag.reshape(-1, order="F"), dtype=np.float64) if np.iscomplexobj(qmap) else None
        q = np.ascontiguousarray(np.asarray(qmap.real, dtype=np.float64).reshape(-1, 3, order="F").ravel(order="F"))
        npix, s = q.size // 3, self.dict_shape[1]
        nch = s if mode == "real" else 2 * s
        X = np.empty(npix * nch, np.float32)
        idx = np.empty(npix, np.int32)
        dp, fp, ip = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32)
        if mode == "real":
            self._check(self.L.qmri_synthesize_tsmi(self.h, q.ctypes.data_as(dp), npix, X.ctypes.data_as(fp), idx.ctypes.data_as(ip)))
        else:
            self._check(self.L.qmri_synthesize_tsmi_complex(self.h, q.ctypes.data_as(dp), pd_im.ctypes.data_as(dp) if pd_im is not None else None,
                                                            npix, X.ctypes.data_as(fp), idx.ctypes.data_as(ip)))
        return X.reshape(shp + (nch,), order="F"), idx.reshape(shp, order="F")

    # -- LRTV option ---------------------------------------------------------------------------------
    def lrtv(self, y, K=4e-5, iters=200, step=None, tol=1e-4, backtrack=True, prox_tol=None, prox_maxit=None):
        """x = FISTA_deep(data, param)  (FISTA_deep.m:1, parameters of main_recon_tsmis_FFT.m:274-281).
        Returns (x [N,M,s] complex, info dict)."""
        if not hasattr(self, "N"):
            raise ValueError("operator not set")
        N, M, s = self.N, self.M, self.s
        p = LrtvParams(float(K), int(iters), float(step) if step else 0.0, float(tol), int(bool(backtrack)),
                       float(prox_tol) if prox_tol else 0.0, int(prox_maxit) if prox_maxit else 0)
        yb = _cbuf(y)
        if yb.size != self.m:
            raise ValueError(f"y must have {self.m} elements")
        x = np.empty(N * M * s, np.complex128)
        info = LrtvInfo()
        self._check(self.L.qmri_lrtv(self.h, _vp(yb), C.byref(p), _vp(x), C.byref(info)))
        return x.reshape((N, M, s), order="F"), {k: getattr(info, k) for k, _ in LrtvInfo._fields_}

    def prox_tv(self, b, gamma, tol=10e-4, maxit=200):
        """[sol, info] = prox_tv(b, gamma)  (unlocbox/prox/prox_tv.m:1) on a real 2-D image.  Returns (sol, iters, obj)."""
        b = np.asfortranarray(b, dtype=np.float64)
        if b.ndim != 2:
            raise ValueError("b must be a 2-D image")
        sol = np.empty_like(b, order="F")
        it, obj = C.c_int32(0), C.c_double(0.0)
        dp = C.POINTER(C.c_double)
        self._check(self.L.qmri_prox_tv(self.h, b.ctypes.data_as(dp), b.shape[0], b.shape[1], float(gamma), float(tol), int(maxit),
                                        sol.ctypes.data_as(dp), C.byref(it), C.byref(obj)))
        return sol, int(it.value), float(obj.value)

    def norm_tv(self, I):
        """y = norm_tv(I)  (unlocbox/utils/norm_tv.m:1)."""
        I = np.asfortranarray(I, dtype=np.float64)
        if I.ndim != 2:
            raise ValueError("I must be a 2-D image")
        out = C.c_double(0.0)
        self._check(self.L.qmri_norm_tv(self.h, I.ctypes.data_as(C.POINTER(C.c_double)), I.shape[0], I.shape[1], C.byref(out)))
        return float(out.value)

    def dict_match(self, X, want_mt=True, want_dm=True, want_xfit=False):
        """out = mrf_dtm_cpu(dict, data, par)  (mrf_dtm_cpu.m:1).  X [..., s] complex -> dict of arrays; want_xfit adds Xfit [..., s]
        complex64 (par.f.Xout, :95,129-134)."""
        X = np.asarray(X, dtype=np.complex128)
        K, s, Q = self.dict_shape
        if X.shape[-1] != s:
            raise ValueError("last dimension of X must equal the dictionary's channel count")
        lead = X.shape[:-1]
        npix = int(np.prod(lead))
        xb = np.ascontiguousarray(X.reshape((npix, s), order="F").ravel(order="F"))
        qmap = np.empty(npix * Q, np.float32)
        pd = np.empty(2 * npix, np.float32)
        mt = np.empty(npix, np.float32) if want_mt else None
        dm = np.empty(npix, np.int32) if want_dm else None
        xfit = np.empty(2 * npix * s, np.float32) if want_xfit else None
        f = C.POINTER(C.c_float)
        self._check(self.L.qmri_dict_match_xfit(self.h, _vp(xb), npix, qmap.ctypes.data_as(f), pd.ctypes.data_as(f),
                                                mt.ctypes.data_as(f) if mt is not None else None,
                                                dm.ctypes.data_as(C.POINTER(C.c_int32)) if dm is not None else None,
                                                xfit.ctypes.data_as(f) if xfit is not None else None))
        out = {"qmap": qmap.reshape(lead + (Q,), order="F"), "pd": pd.view(np.complex64).reshape(lead, order="F")}
        if xfit is not None:
            out["Xfit"] = xfit.view(np.complex64).reshape(lead + (s,), order="F")
        if mt is not None:
            out["mt"] = mt.reshape(lead, order="F")
        if dm is not None:
            out["dm"] = dm.reshape(lead, order="F")
        return out

    def dict_match_dev(self, d_X: int, npix: int, d_qmap: int = 0, d_pd: int = 0, d_mt: int = 0, d_dm: int = 0, d_xfit: int = 0):
        """qmri_dict_match_xfit_dev: device pointers (X Npix x s complex double column-major; outputs as qmri.h lays them out), asynchronous on
        the engine's stream."""
        self._check(self.L.qmri_dict_match_xfit_dev(self.h, C.c_void_p(d_X), int(npix), C.c_void_p(d_qmap or None), C.c_void_p(d_pd or None),
                                                    C.c_void_p(d_mt or None), C.c_void_p(d_dm or None), C.c_void_p(d_xfit or None)))

    # -- profiling -----------------------------------------------------------------------------------
    def profile_enable(self, level: int):
        self._check(self.L.qmri_profile_enable(self.h, int(level)))

    def profile_get(self, reset=True) -> dict:
        p = Profile()
        self._check(self.L.qmri_profile_get(self.h, C.byref(p), int(reset)))
        return {k: getattr(p, k) for k, _ in Profile._fields_}
