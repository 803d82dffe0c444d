"""ctypes bindings for the CPU oracle (oracle/qmri_oracle.h).

TEST INFRASTRUCTURE ONLY: importable from tests/, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
bench.py.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")
_SRCS = ["orc_masks.c", "orc_fft.c", "orc_operator.c", "orc_lsqr.c", "orc_net.c", "orc_admm.c",
         "orc_dictmatch.c", "orc_util.c", "orc_lrtv.c", "qmri_oracle.h", "orc_internal.h", "Makefile"]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc if the .so is missing or older than its sources."""
    stale = force or not os.path.exists(_LIB)
    if not stale:
        t = os.path.getmtime(_LIB)
        stale = any(os.path.getmtime(os.path.join(_HERE, s)) > t for s in _SRCS)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB


class AdmmParams(C.Structure):
    _fields_ = [("gamma", C.c_double), ("iters", C.c_int), ("cg_tol", C.c_double), ("cg_maxit", C.c_int),
                ("solver", C.c_int), ("multi_level", C.c_int), ("noise_std", C.c_double),
                ("residual_noise", C.c_int), ("want_diag", C.c_int)]


_lib = None


def usable_cpus() -> int:
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota (a GPU box exposes every
    host core but grants a share; running one OpenMP thread per visible core oversubscribes it badly)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    env = os.environ.get("OMP_NUM_THREADS")
    if env and env.isdigit():
        n = min(n, int(env)) if int(env) > 0 else n
    return max(1, min(n, 64))


def lib():
    global _lib
    if _lib is None:
        # QMRI_ORACLE_LIB: load another build of the same sources instead (tests run the suite's oracle calls once against
        # `make asan`'s liboracle_asan.so, in a child process started with libasan preloaded)
        L = C.CDLL(os.environ.get("QMRI_ORACLE_LIB") or build())
        vp, ip, dp, fp = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_float)
        L.orc_spiral_mask.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip, C.c_int]
        L.orc_epi_mask.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, ip, ip, C.c_int]
        L.orc_op_create.restype = vp
        L.orc_op_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, dp, ip, ip]
        L.orc_op_destroy.argtypes = [vp]
        L.orc_op_m.argtypes = [vp]
        L.orc_forward.argtypes = [vp, dp, dp]
        L.orc_adjoint.argtypes = [vp, dp, dp]
        L.orc_fft2.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, dp, dp]
        L.orc_lsqr_xupdate.argtypes = [vp, dp, dp, C.c_double, C.c_double, C.c_int, dp,
                                       C.POINTER(C.c_int), C.POINTER(C.c_int), dp]
        L.orc_direct_xupdate.argtypes = [vp, dp, dp, C.c_double, dp]
        L.orc_net_create.restype = vp
        L.orc_net_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, fp, C.c_size_t]
        L.orc_net_destroy.argtypes = [vp]
        L.orc_net_nparams.restype = C.c_size_t
        L.orc_net_nparams.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
        L.orc_net_forward.argtypes = [vp, fp, C.c_int, C.c_int, C.c_int, fp]
        L.orc_denoise.argtypes = [vp, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp]
        L.orc_pnp_admm.argtypes = [vp, vp, dp, C.POINTER(AdmmParams), dp, dp, dp, dp, ip]
        L.orc_admm_stage_seconds.argtypes = [dp]
        L.orc_dict_match.argtypes = [dp, C.c_int, C.c_int, fp, fp, fp, C.c_int, C.c_int, C.c_double,
                                     fp, fp, fp, ip, fp]
        L.orc_norm_tv.restype = C.c_double
        L.orc_norm_tv.argtypes = [dp, C.c_int, C.c_int]
        L.orc_prox_tv.argtypes = [dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, dp, dp]
        L.orc_synthesize_tsmi.argtypes = [dp, C.c_int, fp, fp, fp, C.c_int, C.c_int, fp, ip]
        L.orc_synthesize_tsmi_complex.argtypes = [dp, dp, C.c_int, fp, fp, fp, C.c_int, C.c_int, fp, ip]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_set_num_threads(usable_cpus())
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def _cplx_in(a):
    """complex128 array -> contiguous Fortran-order memory viewed as doubles."""
    a = np.asarray(a, dtype=np.complex128)
    return np.ascontiguousarray(a.ravel(order="F")).view(np.float64)


def spiral_mask(N: int, S: int, T: int):
    cap = S * T
    fp_ = np.zeros(T + 1, np.int32)
    k = np.zeros(cap, np.int32)
    m = lib().orc_spiral_mask(N, S, T, _ip(fp_), _ip(k), cap)
    assert m >= 0
    return fp_, k[:m].copy()


def epi_mask(N: int, M: int, pct: float, T: int):
    step = int(np.floor(1.0 / pct + 0.5))
    cap = (N // step) * M * T
    fp_ = np.zeros(T + 1, np.int32)
    k = np.zeros(max(cap, 1), np.int32)
    m = lib().orc_epi_mask(N, M, float(pct), T, _ip(fp_), _ip(k), cap)
    assert m >= 0
    return fp_, k[:m].copy()


class Operator:
    """struct F of main_recon_tsmis_FFT.m:228-229 (forward / adjoint), CPU oracle."""

    def __init__(self, N, M, V, frame_ptr, kidx):
        V = np.asarray(V, dtype=np.float64)
        self.N, self.M = int(N), int(M)
        self.T, self.s = V.shape
        self._V = np.ascontiguousarray(V.ravel(order="F"))
        self._fp = np.ascontiguousarray(frame_ptr, dtype=np.int32)
        self._k = np.ascontiguousarray(kidx, dtype=np.int32)
        self.h = lib().orc_op_create(self.N, self.M, self.s, self.T, _dp(self._V), _ip(self._fp), _ip(self._k))
        self.m = lib().orc_op_m(self.h)

    def __del__(self):
        try:
            lib().orc_op_destroy(self.h)
        except Exception:
            pass

    def forward(self, x):
        xin = _cplx_in(x)
        y = np.empty(2 * self.m, np.float64)
        lib().orc_forward(self.h, _dp(xin), _dp(y))
        return y.view(np.complex128)

    def adjoint(self, y):
        yin = _cplx_in(y)
        x = np.empty(2 * self.N * self.M * self.s, np.float64)
        lib().orc_adjoint(self.h, _dp(yin), _dp(x))
        return x.view(np.complex128).reshape((self.N, self.M, self.s), order="F")

    # Multi-coil extension (BASELINE.json configs[4]): the reference is single-coil (README.md:63), so these two follow NO reference line -- they
    # restate the textbook SENSE model on top of the single-coil operator above (coil maps times image before the transform; conjugate maps and
    # a sum over the coils behind the adjoint).  Parity unpinned: checked by adjointness and closed forms (tests/test_gpu_operator.py).
    def forward_mc(self, x, maps):
        maps = np.asarray(maps, np.complex128)
        return np.stack([self.forward(maps[..., j, None] * np.asarray(x)) for j in range(maps.shape[2])], axis=1)

    def adjoint_mc(self, y, maps):
        maps, y = np.asarray(maps, np.complex128), np.asarray(y)
        x = np.zeros((self.N, self.M, self.s), np.complex128)
        for j in range(maps.shape[2]):                           # (j ascending: the order the product adds in)
            x += np.conj(maps[..., j, None]) * self.adjoint(y[:, j])
        return x

    def lsqr_mc(self, y_mc, maps, z, r, tol=1e-4, maxit=100, x0=None):
        """Multi-coil x-update (extension, no reference line): orc_lsqr.c's recurrences and stop rules, statement by statement, with the operator
        replaced by forward_mc / adjoint_mc -- x = lsqr(@afun, [y_mc; sqrt(r) z], tol, maxit, [], [], x0), afun: [A_mc; sqrt(r) I].
        Returns (x, iters, flag)."""
        eps = np.finfo(np.float64).eps
        y = np.asarray(y_mc, np.complex128)
        z = np.asarray(z, np.complex128)
        x = np.array(x0 if x0 is not None else np.zeros((self.N, self.M, self.s)), np.complex128)
        sr = np.sqrt(r)
        nsq = lambda a: float(np.sum(a.real ** 2) + np.sum(a.imag ** 2))
        n2b = np.sqrt(nsq(y) + r * nsq(z))
        tolb = tol * n2b
        ut = y - self.forward_mc(x, maps)
        ub = z * sr - x * sr
        beta = np.sqrt(nsq(ut) + nsq(ub))
        normr = beta
        if beta != 0.0:
            ut, ub = ut / beta, ub / beta
        c, s_, phibar = 1.0, 0.0, beta
        v = self.adjoint_mc(ut, maps) + ub * sr
        alpha = np.sqrt(nsq(v))
        if alpha != 0.0:
            v = v / alpha
        normar = alpha * beta
        if normar == 0.0 or n2b == 0.0:
            return x, 0, 0
        d = np.zeros_like(x)
        norma, stag, it, flag = 0.0, 0, maxit, 1
        for ii in range(1, maxit + 1):
            ut = self.forward_mc(v, maps) - alpha * ut
            ub = v * sr - alpha * ub
            beta = np.sqrt(nsq(ut) + nsq(ub))
            ut, ub = ut / beta, ub / beta
            norma = np.sqrt(norma * norma + alpha * alpha + beta * beta)
            thet, rhot = -s_ * alpha, c * alpha
            rho = np.sqrt(rhot * rhot + beta * beta)
            c, s_ = rhot / rho, -beta / rho
            phi = c * phibar
            if phi == 0.0:
                stag = 1
            phibar = s_ * phibar
            d = (v - thet * d) / rho
            stag = stag + 1 if abs(phi) * np.sqrt(nsq(d)) < eps * np.sqrt(nsq(x)) else 0
            if normar / (norma * normr) <= tol or normr <= tolb:
                it, flag = ii - 1, 0
                break
            if stag >= 3:
                it, flag = ii - 1, 3
                break
            x = x + phi * d
            normr = abs(s_) * normr
            v = (self.adjoint_mc(ut, maps) + ub * sr) - beta * v
            alpha = np.sqrt(nsq(v))
            v = v / alpha
            normar = alpha * abs(s_ * phi)
        return x, it, flag

    def lsqr(self, y, z, r, tol=1e-4, maxit=100, x0=None):
        yin, zin = _cplx_in(y), _cplx_in(z)
        x = _cplx_in(x0 if x0 is not None else np.zeros((self.N, self.M, self.s))).copy()
        it, fl, rr = C.c_int(0), C.c_int(0), C.c_double(0)
        lib().orc_lsqr_xupdate(self.h, _dp(yin), _dp(zin), float(r), float(tol), int(maxit), _dp(x),
                               C.byref(it), C.byref(fl), C.byref(rr))
        return x.view(np.complex128).reshape((self.N, self.M, self.s), order="F"), it.value, fl.value, rr.value

    def direct(self, y, z, r):
        yin, zin = _cplx_in(y), _cplx_in(z)
        x = np.empty(2 * self.N * self.M * self.s, np.float64)
        lib().orc_direct_xupdate(self.h, _dp(yin), _dp(zin), float(r), _dp(x))
        return x.view(np.complex128).reshape((self.N, self.M, self.s), order="F")


def fft2(x, sign=-1):
    """Per-channel 2-D DFT of an [N,M,s] complex array (unnormalised forward, 1/(NM) inverse)."""
    x = np.asarray(x, dtype=np.complex128)
    N, M, s = x.shape
    xin = _cplx_in(x)
    out = np.empty_like(xin)
    lib().orc_fft2(N, M, s, sign, _dp(xin), _dp(out))
    return out.view(np.complex128).reshape((N, M, s), order="F")


class Net:
    """param.net of main_recon_tsmis_FFT.m:164, CPU oracle.  arch 0 = UNetRes, 1 = sequential conv stack."""

    def __init__(self, weights, in_nc=10, out_nc=10, nc=(64, 128, 256, 512), nb=4, arch=0):
        self.in_nc, self.out_nc, self.nc, self.nb, self.arch = in_nc, out_nc, tuple(nc), nb, arch
        w = np.ascontiguousarray(weights, dtype=np.float32)
        nc4 = (C.c_int * 4)(*self.nc)
        self.h = lib().orc_net_create(arch, in_nc, out_nc, nc4, nb, _fp(w), w.size)
        if not self.h:
            raise ValueError("weight blob size does not match the architecture")

    def __del__(self):
        try:
            lib().orc_net_destroy(self.h)
        except Exception:
            pass

    def forward_f32(self, x):
        """x: [H,W,C] or [H,W,C,B] float32 (MATLAB dims) -> [H,W,out_nc(,B)] float32."""
        x = np.asarray(x, dtype=np.float32)
        squeeze = x.ndim == 3
        if squeeze:
            x = x[..., None]
        H, W, Cc, B = x.shape
        xin = np.ascontiguousarray(x.ravel(order="F"))
        out = np.empty(H * W * self.out_nc * B, np.float32)
        lib().orc_net_forward(self.h, _fp(xin), H, W, B, _fp(out))
        out = out.reshape((H, W, self.out_nc, B), order="F")
        return out[..., 0] if squeeze else out

    def denoise(self, x, residual_noise=False):
        """denoiseImage_PnP_ADMM(x, net, true, residual_noise): double in, double out."""
        x = np.asarray(x, dtype=np.float64)
        squeeze = x.ndim == 3
        if squeeze:
            x = x[..., None]
        H, W, Cc, B = x.shape
        xin = np.ascontiguousarray(x.ravel(order="F"))
        out = np.empty(H * W * self.out_nc * B, np.float64)
        lib().orc_denoise(self.h, _dp(xin), H, W, Cc, B, int(residual_noise), _dp(out))
        out = out.reshape((H, W, self.out_nc, B), order="F")
        return out[..., 0] if squeeze else out


def pnp_admm(op: Operator, net: Net, y, gamma=0.05, iters=100, cg_tol=1e-4, cg_maxit=100, solver="lsqr",
             multi_level=False, noise_std=0.01, residual_noise=False, x0=None, gt=None, want_diag=False):
    """x = PnP_ADMM(y, param)  (PnP_ADMM.m:1).  Returns (x, diag[iters,2] or None, lsqr_iters[iters])."""
    p = AdmmParams(float(gamma), int(iters), float(cg_tol), int(cg_maxit), 0 if solver == "lsqr" else 1,
                   int(multi_level), float(noise_std), int(residual_noise), int(want_diag))
    yin = _cplx_in(y)
    n = op.N * op.M * op.s
    x = np.empty(2 * n, np.float64)
    x0in = _cplx_in(x0) if x0 is not None else None
    gtin = _cplx_in(gt) if gt is not None else None
    diag = np.zeros(2 * iters, np.float64) if want_diag else None
    li = np.zeros(iters, np.int32)
    lib().orc_pnp_admm(op.h, net.h, _dp(yin), C.byref(p), _dp(x0in), _dp(gtin), _dp(x), _dp(diag), _ip(li))
    xo = x.view(np.complex128).reshape((op.N, op.M, op.s), order="F")
    return xo, (diag.reshape(iters, 2) if diag is not None else None), li


def pnp_admm_mc(op: Operator, net: Net, y_mc, maps, gamma=0.05, iters=100, cg_tol=1e-4, cg_maxit=100, multi_level=False, noise_std=0.01):
    """Multi-coil extension (no reference counterpart, parity unpinned): the loop of PnP_ADMM.m:76-146 with F replaced by the SENSE operator
    (Operator.forward_mc / adjoint_mc) -- x = F.adjoint(Y); v = x; uold = 0; repeat: x-update (lsqr_mc), min-max normalise real(x + uold) over the
    whole stack, denoise in single precision, undo, dual update.  Returns (x, lsqr_iters)."""
    x = op.adjoint_mc(y_mc, maps)
    v = x.copy()
    u = np.zeros_like(x)
    li = np.zeros(iters, np.int32)
    for it in range(iters):
        x, li[it], _ = op.lsqr_mc(y_mc, maps, v - u, gamma, cg_tol, cg_maxit, x0=x)
        w = np.real(x + u)
        lo, hi = w.min(), w.max()
        w = (w - lo) / (hi - lo)
        if multi_level:
            w = np.concatenate([w, np.full(w.shape[:2] + (1,), noise_std)], axis=2)
        vv = net.denoise(w) * (hi - lo) + lo
        u = u + x - vv
        v = vv
    return x, li


def admm_stage_seconds() -> dict:
    """Wall-clock split of the most recent pnp_admm call (bench.py cpu_baseline)."""
    t = np.zeros(4, np.float64)
    lib().orc_admm_stage_seconds(_dp(t))
    return {"xupdate": float(t[0]), "diagnostics": float(t[1]), "denoiser": float(t[2]), "elementwise": float(t[3])}


def norm_tv(I):
    """norm_tv(I) of a real 2-D image (unlocbox/utils/norm_tv.m:45-55)."""
    I = np.asfortranarray(I, dtype=np.float64)
    return float(lib().orc_norm_tv(_dp(I), I.shape[0], I.shape[1]))


def prox_tv(b, gamma, tol=10e-4, maxit=200):
    """[sol, info] = prox_tv(b, gamma) with the defaults FISTA_deep.m uses (prox_tv.m:99-104).  Returns (sol, iters, obj)."""
    b = np.asfortranarray(b, dtype=np.float64)
    sol = np.empty_like(b, order="F")
    obj = C.c_double(0.0)
    it = lib().orc_prox_tv(_dp(b), b.shape[0], b.shape[1], float(gamma), float(tol), int(maxit), _dp(sol), C.byref(obj))
    return sol, int(it), obj.value


def stack_ri(x):
    """[reshape(real(x),N,[]); reshape(imag(x),N,[])]  (FISTA_deep.m:66,75): N x M x L complex -> 2N x (M L) real."""
    N = x.shape[0]
    return np.concatenate([np.real(x).reshape(N, -1, order="F"), np.imag(x).reshape(N, -1, order="F")], axis=0)


def unstack_ri(b, shape):
    N = shape[0]
    return (b[:N, :] + 1j * b[N:, :]).reshape(shape, order="F")


def fista_lrtv(op: Operator, y, K=4e-5, iters=200, step=None, tol=1e-4, backtrack=True):
    """x = FISTA_deep(data, param)  (FISTA_deep.m:31-104) with the parameters of main_recon_tsmis_FFT.m:274-281.
    Returns (x, info) with info = dict(iters, obj[], prox_iters[], step, halvings)."""
    y = np.asarray(y, dtype=np.complex128).ravel()
    shape = (op.N, op.M, op.s)
    if step is None:
        step = (op.N * op.M * op.s) / y.size                               # param.step = numel(X0)/numel(Y), :277
    x = np.zeros(shape, np.complex128)
    x2_prev = x.copy()
    t, obj_prev = 1, 0.0
    objs, pits, halv = [], [], 0
    it = 0
    for it in range(1, iters + 1):
        err = op.forward(x).ravel() - y
        grad1 = op.adjoint(err)
        cvxobj = 0.5 * float(np.vdot(err, err).real)
        val = norm_tv(stack_ri(x))
        while True:
            x2 = x - grad1 * step
            if K > 0:
                b, n, _ = prox_tv(stack_ri(x2), step * K)
                pits.append(n)
                x2 = unstack_ri(b, shape)
            if not backtrack:
                break
            r2 = op.forward(x2).ravel() - y
            tmp = 0.5 * float(np.vdot(r2, r2).real)
            d = (x2 - x).ravel(order="F")
            if tmp > cvxobj + float(np.real(np.vdot(grad1.ravel(order="F"), d))) + 1.0 / (2 * step) * float(np.vdot(d, d).real):
                step = step / 2
                halv += 1
            else:
                break
        x = x2 + (t - 1) / (t + 2) * (x2 - x2_prev)
        x2_prev = x2
        t += 1
        obj = cvxobj + K * val
        objs.append(obj)
        if abs(obj - obj_prev) / obj < tol:
            break
        obj_prev = obj
    return x, {"iters": it, "obj": np.array(objs), "prox_iters": np.array(pits, np.int32), "step": step, "halvings": halv}


def synthesize_tsmi(qmap, D, normD, lut, mode="real"):
    """main_synthesize_tsmis.m:82-103: qmap [..., 3] (T1, T2, PD) -> (X float32, idx 1-based int32).  mode 'real': X [..., s];
    mode 'complex' (PD may be complex): X [..., 2s] = real parts of the s channels, then the imaginary parts."""
    qmap = np.asarray(qmap)
    if mode == "real" and np.iscomplexobj(qmap):
        qmap = np.concatenate([qmap[..., :2].real, np.abs(qmap[..., 2:3])], axis=-1)
    shp = qmap.shape[:-1]
    pim = np.ascontiguousarray(qmap[..., 2].imag.reshape(-1, order="F"), dtype=np.float64) if np.iscomplexobj(qmap) else None
    q = np.asfortranarray(np.asarray(qmap.real, dtype=np.float64).reshape(-1, 3, order="F"))
    D = np.asfortranarray(D, dtype=np.float32)
    lut = np.asfortranarray(lut, dtype=np.float32)
    nd = np.ascontiguousarray(normD, dtype=np.float32).ravel()
    K, s = D.shape
    nch = s if mode == "real" else 2 * s
    X = np.empty((q.shape[0], nch), np.float32, order="F")
    idx = np.empty(q.shape[0], np.int32)
    if mode == "real":
        lib().orc_synthesize_tsmi(_dp(q), q.shape[0], _fp(D), _fp(nd), _fp(lut), K, s, _fp(X), _ip(idx))
    else:
        lib().orc_synthesize_tsmi_complex(_dp(q), _dp(pim), q.shape[0], _fp(D), _fp(nd), _fp(lut), K, s, _fp(X), _ip(idx))
    return X.reshape(shp + (nch,), order="F"), idx.reshape(shp, order="F")


def dict_match(X, D, normD, lut, block_size=1e9, want_mt=True, want_dm=True, want_xfit=False):
    """out = mrf_dtm_cpu(dict, data, par)  (mrf_dtm_cpu.m:1).  X: [...,s] complex; returns dict of arrays."""
    X = np.asarray(X, dtype=np.complex128)
    s = X.shape[-1]
    lead = X.shape[:-1]
    npix = int(np.prod(lead))
    xin = np.ascontiguousarray(X.reshape((npix, s), order="F").ravel(order="F")).view(np.float64)
    D = np.asarray(D, dtype=np.float32)
    K = D.shape[0]
    lut = np.asarray(lut, dtype=np.float32)
    Q = lut.shape[1]
    Df = np.ascontiguousarray(D.ravel(order="F"))
    lf = np.ascontiguousarray(lut.ravel(order="F"))
    nd = np.ascontiguousarray(normD, dtype=np.float32)
    qmap = np.empty(npix * Q, np.float32)
    pd = np.empty(2 * npix, np.float32)
    mt = np.empty(npix, np.float32) if want_mt else None
    dm = np.empty(npix, np.int32) if want_dm else None
    xf = np.empty(2 * npix * s, np.float32) if want_xfit else None
    lib().orc_dict_match(_dp(xin), npix, s, _fp(Df), _fp(nd), _fp(lf), K, Q, float(block_size),
                         _fp(qmap), _fp(pd), _fp(mt), _ip(dm), _fp(xf))
    out = {"qmap": qmap.reshape(lead + (Q,), order="F"),
           "pd": pd.view(np.complex64).reshape(lead, order="F")}
    if mt is not None:
        out["mt"] = mt.reshape(lead, order="F")
    if dm is not None:
        out["dm"] = dm.reshape(lead, order="F")
    if xf is not None:
        out["Xfit"] = xf.view(np.complex64).reshape(lead + (s,), order="F")
    return out


def num_threads() -> int:
    return lib().orc_num_threads()


def set_num_threads(n: int) -> None:
    lib().orc_set_num_threads(int(n))
