"""The harness around the hot path: what `main_recon_tsmis_FFT.m` does before and after `PnP_ADMM` / `mrf_dtm_cpu`
 -- load the `.mat` inputs, crop, subsample + noise, reconstruct, match, and the metrics
the script prints -- so that someone holding the reference's data files gets the script's numbers from this engine.
Host logic in numpy/scipy; the reconstruction and the match run on the GPU through `reference_api`.

  load_mat(path)                       `load(...)` of a MATLAB file: v5/v7 (scipy.io) or -v7.3 (HDF5; mat73.py, no HDF5 library needed)
  load_dictionary(path)                `load(dict_dir); V = real(dict.V)`                     main_recon_tsmis_FFT.m:121-130
  load_tsmi(path) / crop_tsmi(X)       `load(tsmi_dir); X0 = X((4:227),(4:227),:)`            :199-212
  load_qmaps(path, slice)              qmap(slice,:,:,:) -> N x M x 3, cropped the same way   :177-189
  getmask_fromPD(PD, thresh)           foreground mask                                         getmask_fromPD.m:9-15
  awgn_measured(y, snr_db, seed)       `awgn(Y, snr, 'measured')` with an explicit seed       :243
  psnr(A, ref) / ssim(A, ref)          MATLAB `psnr` / `ssim` defaults for double images [MathWorks]   :352-372
  metrics(qmap, qmap0, mask, X, X0)    the block :327-374 as a dict
  synthesize_tsmis(qmap, dictionary)   main_synthesize_tsmis.m:76-100: quantitative maps -> TSMIs (GPU)
  training_volume / save_training_pickle   the `.mat` -> training-pickle layout of main_save_python_tsmis.py:132-190
  recon_tsmis(...)                     the script's main flow for recon_method 'SVD_MRF' | 'LRTV' | 'PnP_ADMM'  :263-319

[MathWorks] functions are restated from their documented defaults (no MATLAB here: parity unpinned for them, see
DESIGN.md section 10): psnr peak value 1 for class double; ssim with an isotropic Gaussian of sigma 1.5 truncated at
radius ceil(3 sigma) = 5 (11 x 11), border replication, exponents 1, C1 = (0.01 L)^2, C2 = (0.03 L)^2, L = 1 for double,
mean over all pixels; imfill(I, 8, 'holes') followed by `> 0` = every zero pixel that is not 8-connected to the border
through zeros becomes foreground.
"""
from __future__ import annotations

import numpy as np

__all__ = ["load_mat", "load_dictionary", "load_tsmi", "crop_tsmi", "load_qmaps", "getmask_fromPD", "awgn_measured",
           "psnr", "ssim", "metrics", "recon_tsmis", "synthesize_tsmis", "training_volume", "save_training_pickle"]

CROP = slice(3, 227)          # MATLAB (4:227): 230 -> 224                                       main_recon_tsmis_FFT.m:189,212


# ------------------------------------------------------------------------------------------------------------
# files
# ------------------------------------------------------------------------------------------------------------
def load_mat(path):
    """dict of the variables in a MATLAB file; structs become objects with attribute access.  v5 / v7 files go through scipy.io,
    -v7.3 files (HDF5 containers) through the dependency-free reader in mat73.py -- same shapes, same squeezing."""
    from . import mat73
    if mat73.is_mat73(path):
        return mat73.load_mat73(path, squeeze_me=True)
    import scipy.io
    return {k: v for k, v in scipy.io.loadmat(path, squeeze_me=True, struct_as_record=False).items() if not k.startswith("__")}


def load_dictionary(path):
    """`load(dict_dir)` -> the fields the path uses: V (T x s, real part taken as in :129), D (K x s), normD (K), lut (K x Q)."""
    d = load_mat(path)
    if "dict" not in d:
        raise KeyError(f"{path} holds no variable 'dict'")
    s = d["dict"]
    from .engine import real_dictionary_array
    # V: real(dict.V) as :129 takes it.  D: used as stored by mrf_dtm_cpu.m:91 -- a complex-typed D must have a zero imaginary part
    out = {"V": np.real(np.asarray(s.V)).astype(np.float64), "D": real_dictionary_array(s.D, "dict.D", np.float32),
           "lut": np.asarray(s.lut, dtype=np.float32)}
    out["normD"] = np.asarray(s.normD, dtype=np.float32).ravel() if hasattr(s, "normD") else np.linalg.norm(out["D"], axis=1).astype(np.float32)
    return out


def crop_tsmi(X):
    """X0((4:227),(4:227),:)"""
    X = np.asarray(X)
    if X.shape[0] < 227 or X.shape[1] < 227:
        raise ValueError(f"cannot crop (4:227, 4:227) out of {X.shape}")
    return X[CROP, CROP, ...]


def load_tsmi(path, crop=True):
    d = load_mat(path)
    if "X" not in d:
        raise KeyError(f"{path} holds no variable 'X'")
    X = np.asarray(d["X"])
    return crop_tsmi(X) if crop else X


def load_qmaps(path, slice_index, crop=True):
    """qmap (slices x 3 x W x H in the file, :180 'Batch x C x W x H') -> N x M x 3 of one slice (1-based index as in the script).
    The script's permute/reshape chain (:182-186) amounts to moving the channel axis last."""
    d = load_mat(path)
    if "qmap" not in d:
        raise KeyError(f"{path} holds no variable 'qmap'")
    q = np.asarray(d["qmap"])
    if q.ndim != 4:
        raise ValueError(f"qmap has {q.ndim} dimensions, expected slices x 3 x N x M")
    q = np.transpose(q[slice_index - 1], (1, 2, 0))
    return q[CROP, CROP, :] if crop else q


# ------------------------------------------------------------------------------------------------------------
# small image-processing pieces
# ------------------------------------------------------------------------------------------------------------
def getmask_fromPD(PD, thresh):
    """getmask_fromPD.m:9-15: |PD| scaled to unit maximum, values below thresh zeroed, holes filled (8-connected
    background), then binarised."""
    from scipy import ndimage
    pd = np.abs(np.asarray(PD)).astype(np.float64)
    mx = pd.max()
    pd = pd / mx if mx > 0 else pd
    fg = pd >= thresh                                          # pd(pd < thresh) = 0 ; later mask(mask > 0) = 1
    fg &= pd > 0
    # zero pixels reachable from the border through zeros (8-connectivity) stay background, every other pixel is filled
    lab, n = ndimage.label(~fg, structure=np.ones((3, 3), bool))
    border = np.unique(np.concatenate([lab[0, :], lab[-1, :], lab[:, 0], lab[:, -1]]))
    outside = np.isin(lab, border[border > 0])
    return (~outside).astype(np.float64)


def awgn_measured(y, snr_db, seed=0):
    """`awgn(Y, snr, 'measured')` for complex Y: noise power = mean(|y|^2) / 10^(snr/10), split equally between the real
    and imaginary parts.  MATLAB draws from its global stream; here the stream is numpy's PCG64 with an explicit seed."""
    y = np.asarray(y, dtype=np.complex128)
    p = np.mean(np.abs(y) ** 2) / (10.0 ** (snr_db / 10.0))
    rng = np.random.default_rng(seed)
    n = rng.standard_normal(y.shape) + 1j * rng.standard_normal(y.shape)
    return y + np.sqrt(p / 2.0) * n


def psnr(A, ref, peakval=1.0):
    """MATLAB psnr(A, ref) for double inputs: 10 log10(peakval^2 / mean((A - ref)^2)), peakval = 1."""
    A, ref = np.asarray(A, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    if A.shape != ref.shape:
        raise ValueError("A and ref must have the same size")     # images:validate:unequalSizeMatrices
    mse = np.mean((A - ref) ** 2)
    return float("inf") if mse == 0 else float(10.0 * np.log10(peakval * peakval / mse))


def _gauss_kernel(sigma=1.5):
    r = int(np.ceil(3 * sigma))
    x = np.arange(-r, r + 1, dtype=np.float64)
    g = np.exp(-(x * x) / (2 * sigma * sigma))
    return g / g.sum()


def _gfilt(img, g):
    """separable Gaussian with border replication (imfilter(..., 'replicate'))"""
    from scipy import ndimage
    return ndimage.correlate1d(ndimage.correlate1d(img, g, axis=0, mode="nearest"), g, axis=1, mode="nearest")


def ssim(A, ref, dynamic_range=1.0, sigma=1.5, K=(0.01, 0.03)):
    """MATLAB ssim(A, ref) defaults for 2-D double images; returns the global value (mean of the local map)."""
    A, ref = np.asarray(A, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    if A.shape != ref.shape or A.ndim != 2:
        raise ValueError("A and ref must be 2-D images of the same size")
    g = _gauss_kernel(sigma)
    C1, C2 = (K[0] * dynamic_range) ** 2, (K[1] * dynamic_range) ** 2
    mux, muy = _gfilt(A, g), _gfilt(ref, g)
    sxx = np.maximum(_gfilt(A * A, g) - mux * mux, 0.0)
    syy = np.maximum(_gfilt(ref * ref, g) - muy * muy, 0.0)
    sxy = _gfilt(A * ref, g) - mux * muy
    num = (2 * mux * muy + C1) * (2 * sxy + C2)
    den = (mux * mux + muy * muy + C1) * (sxx + syy + C2)
    return float(np.mean(num / den))


def metrics(qmap, qmap0, foreground_mask, X=None, X0=None):
    """main_recon_tsmis_FFT.m:327-374.  qmap / qmap0: N x M x 3 (T1, T2, PD; PD may be complex), mask N x M.
    MAE inside the mask, PSNR / SSIM over the whole masked images, TSMI PSNR / SSIM as channel means of |X|."""
    qmap, qmap0 = np.asarray(qmap), np.asarray(qmap0)
    m = np.asarray(foreground_mask, dtype=np.float64)
    ind = m > 0
    out = {}
    maps = {}
    for i, name in enumerate(("t1", "t2")):
        maps[name] = (np.real(qmap[:, :, i]).astype(np.float64) * m, np.real(qmap0[:, :, i]).astype(np.float64) * m)
    pd, pd_ref = np.abs(qmap[:, :, 2] * m), np.abs(qmap0[:, :, 2] * m)
    maps["pd"] = (pd / pd.max() if pd.max() > 0 else pd, pd_ref / pd_ref.max() if pd_ref.max() > 0 else pd_ref)   # :339-342
    for name, (a, r) in maps.items():
        out[f"{name}_mae"] = float(np.mean(np.abs(a[ind] - r[ind]))) if ind.any() else float("nan")
        out[f"{name}_psnr"] = psnr(a, r)
        out[f"{name}_ssim"] = ssim(a, r)
    if X is not None and X0 is not None:
        X, X0 = np.asarray(X), np.asarray(X0)
        out["tsmi_mean_psnr"] = float(np.mean([psnr(np.abs(X[:, :, c]), np.abs(X0[:, :, c])) for c in range(X0.shape[2])]))
        out["tsmi_mean_ssim"] = float(np.mean([ssim(np.abs(X[:, :, c]), np.abs(X0[:, :, c])) for c in range(X0.shape[2])]))
    return out


def synthesize_tsmis(qmap, dictionary, device=0, mode="real"):
    """main_synthesize_tsmis.m:76-103 for one volume: qmap slices x 3 x N x M (the file layout, :180) -> X slices x N x M x C single.
    mode 'real' (:27,91-98): C = s, |PD| folded in, first SVD channel non-negative; mode 'complex' (:100-103): PD may be complex,
    C = 2s (real parts of the s channels, then the imaginary parts).  Runs on the GPU (exhaustive nearest-entry search)."""
    from . import reference_api as R
    q = np.asarray(qmap)
    q = q.astype(np.complex128 if np.iscomplexobj(q) else np.float64)
    if q.ndim != 4 or q.shape[1] != 3:
        raise ValueError("qmap must be slices x 3 x N x M")
    eng = R._engine(device)
    eng.set_dictionary(dictionary["D"], dictionary["normD"], dictionary["lut"])
    return np.stack([eng.synthesize_tsmi(np.transpose(q[i], (1, 2, 0)), mode=mode)[0] for i in range(q.shape[0])])


def training_volume(X_slices, channels_to_save=None):
    """PyTorch_Denoiser/main_save_python_tsmis.py:132-166: the TSMIs of one volume (slices x N x M x C, as synthesize_tsmis
    returns them or as loaded from the per-slice `.mat` files) -> the float64 array slices x C x N x M the training kit
    pickles (`data_slice` transposed (2,1,0) then (0,2,1), i.e. channel first); optionally only the first channels."""
    v = np.moveaxis(np.asarray(X_slices, dtype=np.float64), 3, 1)
    return np.ascontiguousarray(v if channels_to_save is None else v[:, :channels_to_save])


def save_training_pickle(path, X_slices, channels_to_save=None):
    """pickle.dump(vol_data, f) of main_save_python_tsmis.py:184-190 (file naming is the caller's)."""
    import pickle
    with open(path, "wb") as f:
        pickle.dump(training_volume(X_slices, channels_to_save), f)


# ------------------------------------------------------------------------------------------------------------
# the script's main flow
# ------------------------------------------------------------------------------------------------------------
def recon_tsmis(dictionary, X0, qmap0, weights=None, recon_method="PnP_ADMM", subsampling_pattern="Spiral",
                spiral_sampling_curve=771, epi_sampling_rate=1 / 65, measurements_type="noisy", measurements_noise=30,
                denoiser_type="single_level", noise_map_std=0.01, residual_noise=False, iters=100, seed=0, Y=None, device=0,
                net_arch=None, lrtv_iters=None):
    """main_recon_tsmis_FFT.m:216-374 on already loaded (and cropped) arrays.

    dictionary  dict(V, D, normD, lut) (load_dictionary);  X0  N x M x s ground-truth TSMI;  qmap0  N x M x 3
    weights     flat fp32 UNetRes weights, or the path of a `.pt` / `.onnx` file (weights.load_denoiser_weights); needed for PnP_ADMM
    Y           precomputed measurements (the script's save / load option, :248-262) instead of subsample + noise
    net_arch    dict(nc=..., nb=...) when `weights` is a flat blob of a non-default UNetRes (files carry their architecture)
    Returns dict(X, qmap (N x M x 3: T1, T2, PD), Y, metrics, foreground_mask).
    """
    from . import reference_api as R
    net_arch = dict(net_arch or {})
    X0 = np.asarray(X0)
    N, M, s = X0.shape
    V = np.asarray(dictionary["V"], dtype=np.float64)
    if subsampling_pattern == "Spiral":
        P = R.setup_subsampling_spiralgrided(N, M, spiral_sampling_curve, V)
    elif subsampling_pattern == "EPI":
        P = R.setup_subsampling_epi(N, M, epi_sampling_rate, V)
    else:
        raise ValueError(f"unknown subsampling pattern {subsampling_pattern}")
    F = R.make_F(P, device=device)
    if Y is None:
        Y = F.forward(X0.astype(np.complex128))                                      # :237
        if measurements_type == "noisy":
            Y = awgn_measured(Y, measurements_noise, seed=seed)                      # :243
        elif measurements_type != "clean":
            raise ValueError(f"unknown measurements type {measurements_type}")
    if recon_method == "SVD_MRF":                                                    # :270-271
        X = F.adjoint(np.asarray(Y, dtype=np.complex128))
    elif recon_method == "PnP_ADMM":                                                 # :284-293
        if weights is None:
            raise ValueError("PnP_ADMM needs the denoiser weights")
        arch = {}
        if isinstance(weights, (str, bytes)) or hasattr(weights, "__fspath__"):
            from .weights import load_denoiser_weights
            weights, arch = load_denoiser_weights(weights)
            want = 10 if denoiser_type == "single_level" else 11
            if arch["in_nc"] != want:
                raise ValueError(f"the weight file takes {arch['in_nc']} input channels, denoiser type {denoiser_type} needs {want}")
        net = R.make_net(weights, denoiser_type, residual_noise, H=N, W=M, out_nc=s, device=device, **{**net_arch, **({"nc": arch["nc"], "nb": arch["nb"]} if arch else {})})
        param = {"eta": 20, "sigma_squared": 1, "gamma": 1 / 20, "iter": iters, "cg_tol": 1e-4, "F": F, "gt_tsmi": X0,
                 "X0": F.adjoint(Y), "net": net, "denoiser_type": denoiser_type,
                 "noise_map": R.build_noise_map(noise_map_std, N, M)}                # :166-171
        X = R.PnP_ADMM(np.asarray(Y, dtype=np.complex128), param)
    elif recon_method == "LRTV":                                                     # :273-282
        param = {"K": 4e-5, "iter": 200 if lrtv_iters is None else int(lrtv_iters), "step": X0.size / np.asarray(Y).size, "tol": 1e-4,
                 "backtrack": 1, "usegpu": 0}
        X = R.FISTA_deep({"N": M, "M": M, "L": s, "y": np.asarray(Y, dtype=np.complex128), "F": F, "D": []}, param)
    else:
        raise ValueError(f"unknown reconstruction method {recon_method}")
    par = {"f": {"qout": 1, "pdout": 1, "mtout": 0, "Xout": 0, "dmout": 0, "Yout": 0, "verbose": 0}, "fp": {"blockSize": 1e9}}   # :302-309
    out = R.mrf_dtm_cpu(dictionary, {"X": X}, par, device=device)
    qmap = np.concatenate([np.asarray(out["qmap"], dtype=np.complex128), np.asarray(out["pd"]).reshape(N, M, 1)], axis=2)       # :316
    mask = getmask_fromPD(np.asarray(qmap0)[:, :, 2], 0.15)                          # :192
    return {"X": X, "qmap": qmap, "Y": Y, "foreground_mask": mask, "metrics": metrics(qmap, qmap0, mask, X, X0)}
