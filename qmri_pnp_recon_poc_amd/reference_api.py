"""Host-side mirror of the reference's plugin surface for the hot path (same names, argument meaning and error
behaviour as the MATLAB functions), implemented on top of the C ABI.  MATLAB itself is not available in this
pipeline, so this Python mirror is the executable counterpart of the `.m` wrappers in `matlab/`.

    P   = setup_subsampling_spiralgrided(N, M, S, V)        # setup_subsampling_spiralgrided.m:1
    P   = setup_subsampling_epi(N, M, percentage, V)        # setup_subsampling_epi.m:1
    F   = make_F(P)                                         # F.forward / F.adjoint, main_recon_tsmis_FFT.m:228-229
    net = make_net(weights, denoiser_type, residual_noise)  # param.net, main_recon_tsmis_FFT.m:164
    x   = PnP_ADMM(y, param)                                # PnP_ADMM.m:1, param = dict with the reference's field names
    out = mrf_dtm_cpu(dict, data, par)                      # mrf_dtm_cpu.m:1 (name kept; it runs on the GPU)
    x   = FISTA_deep(data, param)                           # LRTV option, FISTA_deep.m:1 (+ TV_operator, prox_tv, norm_tv)
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np

from . import engine as E

_engines = {}


def _engine(device=0) -> E.Engine:
    if device not in _engines:
        _engines[device] = E.Engine(device)
    return _engines[device]


def release():
    for e in _engines.values():
        e.close()
    _engines.clear()


def setup_subsampling_spiralgrided(N, M, S, V):
    """Gridded spiral masks (setup_subsampling_spiralgrided.m:7-34).  The reference returns closures over the sparse
    matrix P; here P is its defining data (masks + V) and the products happen inside F."""
    V = np.real(np.asarray(V, dtype=np.complex128)).astype(np.float64)        # V = real(dict.V), main_recon_tsmis_FFT.m:129
    fp, k = E.build_spiral(int(N), int(S), V.shape[0])
    return SimpleNamespace(N=int(N), M=int(M), V=V, frame_ptr=fp, kidx=k, pattern="Spiral")


def setup_subsampling_epi(N, M, percentage, V):
    """Multi-shot EPI comb masks (setup_subsampling_epi.m:20-33)."""
    V = np.real(np.asarray(V, dtype=np.complex128)).astype(np.float64)
    fp, k = E.build_epi(int(N), int(M), float(percentage), V.shape[0])
    return SimpleNamespace(N=int(N), M=int(M), V=V, frame_ptr=fp, kidx=k, pattern="EPI")


def make_F(P, device=0):
    """F.forward = @(x) P.for(reshape(fft2(x),[],1))/sqrt(N*M);  F.adjoint = @(x) ifft2(reshape(P.adj(x),N,M,[]))*sqrt(N*M)."""
    eng = _engine(device)
    eng.set_operator(P.N, P.M, P.V, P.frame_ptr, P.kidx)
    return SimpleNamespace(forward=eng.forward, adjoint=eng.adjoint, _engine=eng, _P=P)


def denoiseImage_PnP_ADMM(A, net, onnx_dagnetwork=True, residual_noise=False):
    """I = denoiseImage_PnP_ADMM(A, net, onnx_dagnetwork, residual_noise)  (denoiseImage_PnP_ADMM.m:1).
    `net` is the handle returned by make_net; input validation follows validateInputImage (:119-127)."""
    A = np.asarray(A)
    if np.iscomplexobj(A):
        raise TypeError("Expected A to be real.")                            # validateattributes 'real'
    if A.size == 0:
        raise ValueError("Expected A to be nonempty.")
    if not np.all(np.isfinite(A)):
        raise ValueError("Expected A to be finite.")                         # 'nonnan','finite'
    if A.ndim > 4:
        raise ValueError("images:denoiseImage:invalidImageFormat")
    if bool(residual_noise) != bool(net._residual_noise):
        raise ValueError("residual_noise differs from the value the network handle was created with")
    return net._engine.denoise(A)


def make_net(weights, denoiser_type="single_level", residual_noise=False, H=224, W=224, nc=(64, 128, 256, 512), nb=4,
             out_nc=10, device=0):
    """param.net = @(x) denoiseImage_PnP_ADMM(x, Net, true, residual_noise)  (main_recon_tsmis_FFT.m:138-164)."""
    if denoiser_type not in ("single_level", "multi_level"):
        raise ValueError(f"unknown denoiser_type {denoiser_type}")
    eng = _engine(device)
    in_nc = out_nc + (1 if denoiser_type == "multi_level" else 0)
    eng.set_denoiser(weights, H, W, in_nc=in_nc, out_nc=out_nc, nc=nc, nb=nb, residual_noise=residual_noise)

    def net(x):
        return denoiseImage_PnP_ADMM(x, net, True, residual_noise)

    net._engine, net._residual_noise, net._denoiser_type = eng, bool(residual_noise), denoiser_type
    return net


def build_noise_map(noise_std, rows, cols):
    """noise_map = repmat(noise_std, rows, cols)  (build_noise_map.m:19)."""
    return np.full((rows, cols), float(noise_std))


def PnP_ADMM(y, param):
    """x = PnP_ADMM(y, param)  (PnP_ADMM.m:1).  param: dict with iter, gamma, F, cg_tol, gt_tsmi, net, denoiser_type,
    noise_map (multi_level), X0 (PnP_ADMM.m:62-76).  F and net must be the handles made by make_F / make_net on the same
    device: the whole loop then runs on the GPU with one boundary crossing."""
    F, net = param["F"], param["net"]
    if not hasattr(F, "_engine") or not hasattr(net, "_engine") or F._engine is not net._engine:
        raise TypeError("param.F and param.net must come from make_F / make_net of this package (same device)")
    multi = param.get("denoiser_type", net._denoiser_type) == "multi_level"
    noise_std = float(np.asarray(param["noise_map"]).ravel()[0]) if multi else 0.01
    x, diag, li = F._engine.pnp_admm(y, gamma=param["gamma"], iters=int(param["iter"]), cg_tol=param["cg_tol"], cg_maxit=100,
                                     solver=param.get("solver", "lsqr"), multi_level=multi, noise_std=noise_std,
                                     x0=param.get("X0"), gt=param.get("gt_tsmi"), want_diag=param.get("gt_tsmi") is not None)
    PnP_ADMM.last_diagnostics, PnP_ADMM.last_lsqr_iters = diag, li
    return x


def FISTA_deep(data, param):
    """[x] = FISTA_deep(data, param)  (FISTA_deep.m:1): data = dict(N, M, L, y, F, D), param = dict(K, iter, step, tol,
    backtrack, usegpu) as main_recon_tsmis_FFT.m:274-281 builds them.  F must come from make_F; the whole loop runs on the
    GPU (param.usegpu is ignored)."""
    F = data["F"]
    if not hasattr(F, "_engine"):
        raise TypeError("data.F must come from make_F of this package")
    eng = F._engine
    if (data["N"], data["M"], data["L"]) != (eng.N, eng.M, eng.s) and (data["M"], data["M"], data["L"]) != (eng.N, eng.M, eng.s):
        raise ValueError("data.N / M / L do not match the operator")       # (the script passes data.N = M, :281)
    x, info = eng.lrtv(data["y"], K=param["K"], iters=int(param["iter"]), step=param.get("step"), tol=param["tol"],
                       backtrack=param.get("backtrack", 1))
    FISTA_deep.last_info = info
    return x


def TV_operator(mode="2D", usegpu=0, device=0):
    """J = TV_operator('2D', usegpu)  (TV_operator.m:1): J.norm / J.prox applied slice by slice along the third dimension."""
    if mode != "2D":
        raise NotImplementedError("only the 2-D operator is on the path (main_recon_tsmis_FFT.m:280)")
    eng = _engine(device)

    def norm(x2):
        x2 = np.asarray(x2, dtype=np.float64)
        x2 = x2[:, :, None] if x2.ndim == 2 else x2
        return float(sum(eng.norm_tv(x2[:, :, i]) for i in range(x2.shape[2])))

    def prox(x2, gamma):
        x2 = np.asarray(x2, dtype=np.float64)
        squeeze = x2.ndim == 2
        x2 = x2[:, :, None] if squeeze else x2
        out = np.stack([eng.prox_tv(x2[:, :, i], gamma)[0] for i in range(x2.shape[2])], axis=2)
        return out[:, :, 0] if squeeze else out

    return SimpleNamespace(norm=norm, prox=prox)


def mrf_dtm_cpu(dict_, data, par, device=0):
    """out = mrf_dtm_cpu(dict, data, par)  (mrf_dtm_cpu.m:1): dict.{D,normD,lut}, data.X, par.f.{qout,pdout,mtout,dmout,Xout}."""
    eng = _engine(device)
    eng.set_dictionary(dict_["D"], dict_["normD"], dict_["lut"])
    f = par.get("f", {})
    r = eng.dict_match(data["X"], want_mt=True, want_dm=True, want_xfit=bool(f.get("Xout", 0)))
    out = {}
    if f.get("qout", 1):
        out["qmap"], out["mask"] = r["qmap"], np.ones(np.asarray(data["X"]).shape[:-1], bool)
    if f.get("pdout", 1):
        out["pd"] = r["pd"]
    if f.get("mtout", 0):
        out["mt"] = r["mt"]
    if f.get("dmout", 0):
        out["dm"] = r["dm"].astype(np.float32)
    if f.get("Xout", 0):                                      # mrf_dtm_cpu.m:129-134: the scaled matched atoms and the input
        out["Xfit"], out["X"] = r["Xfit"], data["X"]
    return out
