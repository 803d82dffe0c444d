"""Seeded, dependency-free synthetic inputs for the PnP-ADMM MRF hot path.

The reference ships no data, dictionary or trained weights (`.MISSING_LARGE_BLOBS:1-5`,
`datasets/README.md:17-19`, `onnx_models/README.md:11-29`), so every test and benchmark input is generated
here, identically for the HIP path and for the CPU oracle:

* phantom quantitative maps (T1, T2, PD)          -- stands in for `datasets/gt_qmaps/qmap_gt_vol*.mat`
* a synthetic SVD-compressed dictionary            -- stands in for `dictionaries/.../SVD_dict_FISP_cut*.mat`
  (fields D, normD, lut, V as documented in `mrf_dtm_cpu.m:8-12`)
* TSMI synthesis exactly as `main_synthesize_tsmis.m:82-98` (nearest atom x normD x PD, sign-aligned to ch 1)
* complex AWGN at a *measured* SNR, as `awgn(Y,30,'measured')` (`main_recon_tsmis_FFT.m:243`) but seeded
* UNetRes weights in state-dict order (`network_unet.py:68-117`): procedural-random or structured-synthetic

Random numbers come from a counter-based SplitMix64 so that any consumer (numpy here, C++ elsewhere) can
reproduce them bit for bit.
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """n 64-bit outputs of SplitMix64 started at `seed`, counter positions offset .. offset+n-1."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """Doubles in [0,1) with 53 random bits."""
    return (splitmix64(seed, n, offset) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(seed: int, n: int) -> np.ndarray:
    """Standard normals by Box-Muller from two uniform streams."""
    u1 = uniform01(seed, n, 0)
    u2 = uniform01(seed ^ 0x5DEECE66D, n, 1 << 32)
    u1 = np.maximum(u1, 2.0 ** -53)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


# ----------------------------------------------------------------------------------------------------
# dictionary
# ----------------------------------------------------------------------------------------------------
def flip_angle_train(T: int, seed: int = 0) -> np.ndarray:
    """Smooth pseudo-random flip-angle train in radians (FISP-MRF style lobes)."""
    t = np.arange(T, dtype=np.float64)
    lobes = np.abs(np.sin(np.pi * t / 125.0)) * (10.0 + 50.0 * (0.5 + 0.5 * np.sin(2 * np.pi * t / 410.0 + 0.7)))
    jitter = 2.0 * (uniform01(seed, T) - 0.5)
    return np.deg2rad(5.0 + lobes + jitter)


def make_dictionary(T: int = 200, n_t1: int = 128, n_t2: int = 64, s: int = 10, seed: int = 0, uncompressed: bool = False) -> dict:
    """Synthetic MRF dictionary with K = n_t1*n_t2 atoms.

    uncompressed=True: D holds the unit-norm fingerprints themselves, [K, T] (the s = T case of mrf_dtm_cpu.m:41-50, V = identity);
    `s` is ignored and V is None.

    Returns dict(D [K,s] f32 unit-norm, normD [K] f32, lut [K,2] f32 = (T1,T2) seconds, V [T,s] f64 with
    orthonormal columns) -- the fields `main_recon_tsmis_FFT.m:127-129` loads and `mrf_dtm_cpu.m:91-96` uses.
    Signal model: inversion, then per frame an RF rotation by alpha_t, readout with exp(-TE/T2), T1 recovery
    and T2 decay over TR with a fixed spoiling factor.  Not a Bloch/EPG simulation; only the (T1,T2)
    manifold structure and the low rank matter here.
    """
    t1 = np.exp(np.linspace(np.log(0.1), np.log(4.0), n_t1))
    t2 = np.exp(np.linspace(np.log(0.01), np.log(0.6), n_t2))
    T1, T2 = np.meshgrid(t1, t2, indexing="ij")
    T1 = T1.ravel()
    T2 = T2.ravel()
    K = T1.size
    alpha = flip_angle_train(T, seed)
    TR, TE, spoil = 0.012, 0.002, 0.6
    e1 = np.exp(-TR / T1)
    e2 = np.exp(-TR / T2) * spoil
    ete = np.exp(-TE / T2)
    mx = np.zeros(K)
    mz = -np.ones(K)
    F = np.empty((K, T), dtype=np.float64)
    for t in range(T):
        ca, sa = np.cos(alpha[t]), np.sin(alpha[t])
        mx, mz = ca * mx + sa * mz, -sa * mx + ca * mz
        F[:, t] = mx * ete
        mx = mx * e2
        mz = 1.0 + (mz - 1.0) * e1
    if uncompressed:
        normD = np.linalg.norm(F, axis=1)
        return {"D": np.ascontiguousarray((F / normD[:, None]).astype(np.float32)), "normD": normD.astype(np.float32),
                "lut": np.ascontiguousarray(np.stack([T1, T2], axis=1).astype(np.float32)), "V": None, "t1_grid": t1, "t2_grid": t2, "K": K}
    # temporal subspace: top-s eigenvectors of F'F (== right singular vectors of F)
    G = F.T @ F
    evals, evecs = np.linalg.eigh(G)
    V = evecs[:, ::-1][:, :s].copy()
    # deterministic sign: largest-magnitude entry of each column positive
    for c in range(s):
        j = int(np.argmax(np.abs(V[:, c])))
        if V[j, c] < 0:
            V[:, c] = -V[:, c]
    Dc = F @ V
    normD = np.linalg.norm(Dc, axis=1)
    D = Dc / normD[:, None]
    lut = np.stack([T1, T2], axis=1)
    return {
        "D": np.ascontiguousarray(D.astype(np.float32)),
        "normD": normD.astype(np.float32),
        "lut": np.ascontiguousarray(lut.astype(np.float32)),
        "V": np.ascontiguousarray(V),
        "t1_grid": t1,
        "t2_grid": t2,
        "K": K,
    }


# ----------------------------------------------------------------------------------------------------
# phantom + TSMI
# ----------------------------------------------------------------------------------------------------
def make_phantom_qmaps(N: int = 224, seed: int = 0) -> np.ndarray:
    """[N,N,3] float64 (T1 s, T2 s, PD) brain-like phantom: nested ellipses + smooth in-region variation.

    Value ranges follow the colour bars of `main_recon_tsmis_FFT.m:391-393` (T1 0..3, T2 0..0.3, PD 0..1);
    air is 0.  `seed` jitters the geometry so that a batch of slices differs slice to slice."""
    j = 2.0 * (uniform01(0xC0FFEE + seed, 64) - 0.5)
    yy, xx = np.meshgrid(np.linspace(-1, 1, N), np.linspace(-1, 1, N), indexing="ij")

    def ell(cx, cy, a, b, ang):
        ca, sa = np.cos(ang), np.sin(ang)
        u = (xx - cx) * ca + (yy - cy) * sa
        v = -(xx - cx) * sa + (yy - cy) * ca
        return (u / a) ** 2 + (v / b) ** 2 <= 1.0

    q = np.zeros((N, N, 3))
    # (T1, T2, PD): scalp/fat, CSF rim, grey matter, white matter, ventricles (CSF), two lesions
    regions = [
        (ell(0.0, 0.0, 0.80 + 0.02 * j[0], 0.92 + 0.02 * j[1], 0.0), (0.35, 0.07, 0.95)),
        (ell(0.0, 0.0, 0.74 + 0.02 * j[0], 0.86 + 0.02 * j[1], 0.0), (2.80, 0.28, 1.00)),
        (ell(0.0, 0.01 * j[2], 0.70 + 0.02 * j[3], 0.82 + 0.02 * j[4], 0.0), (1.40, 0.095, 0.85)),
        (ell(0.02 * j[5], 0.0, 0.56 + 0.03 * j[6], 0.66 + 0.03 * j[7], 0.05 * j[8]), (0.85, 0.065, 0.70)),
        (ell(-0.16 + 0.02 * j[9], -0.05, 0.09 + 0.01 * j[10], 0.24 + 0.02 * j[11], 0.25 + 0.1 * j[12]), (2.80, 0.28, 1.00)),
        (ell(0.16 + 0.02 * j[13], -0.05, 0.09 + 0.01 * j[14], 0.24 + 0.02 * j[15], -0.25 + 0.1 * j[16]), (2.80, 0.28, 1.00)),
        (ell(0.30 + 0.05 * j[17], 0.35 + 0.05 * j[18], 0.07, 0.05, 0.4), (1.90, 0.16, 0.90)),
        (ell(-0.28 + 0.05 * j[19], -0.40 + 0.05 * j[20], 0.05, 0.08, -0.3), (1.10, 0.045, 0.60)),
    ]
    for mask, (t1, t2, pd) in regions:
        q[mask, 0], q[mask, 1], q[mask, 2] = t1, t2, pd
    smooth = 1.0 + 0.06 * np.sin(3.1 * xx + 0.5 * j[21]) * np.cos(2.3 * yy + 0.5 * j[22])
    fg = q[:, :, 2] > 0
    q[:, :, 0] = np.where(fg, np.clip(q[:, :, 0] * smooth, 0.3, 3.0), 0.0)
    q[:, :, 1] = np.where(fg, np.clip(q[:, :, 1] * (2.0 - smooth), 0.03, 0.3), 0.0)
    q[:, :, 2] = np.where(fg, np.clip(q[:, :, 2] * (0.97 + 0.03 * smooth), 0.0, 1.0), 0.0)
    return q


def synthesize_tsmi(qmaps: np.ndarray, dic: dict) -> np.ndarray:
    """[N,M,s] float64 real TSMI, as `main_synthesize_tsmis.m:82-98`:
    nearest dictionary atom in (T1,T2) (`knnsearch` on `dict.lut`, :88) -> D(I,:) * normD(I) (:89-90)
    -> * |PD| (:93) -> multiply every channel by sign(channel 1) (:97-98)."""
    N, M, _ = qmaps.shape
    t1g, t2g = dic["t1_grid"], dic["t2_grid"]

    def nearest(grid, vals):
        idx = np.clip(np.searchsorted(grid, vals), 1, grid.size - 1)
        left = grid[idx - 1]
        right = grid[idx]
        return np.where(vals - left <= right - vals, idx - 1, idx)

    i1 = nearest(t1g, qmaps[:, :, 0].ravel(order="F"))
    i2 = nearest(t2g, qmaps[:, :, 1].ravel(order="F"))
    I = i1 * t2g.size + i2
    X = dic["D"][I, :].astype(np.float64) * dic["normD"][I].astype(np.float64)[:, None]
    X = X * np.abs(qmaps[:, :, 2].ravel(order="F"))[:, None]
    X = X.reshape((N, M, -1), order="F")
    sgn = np.sign(X[:, :, 0])
    return X * sgn[:, :, None]


def awgn_measured(y: np.ndarray, snr_db: float, seed: int) -> np.ndarray:
    """y + complex white Gaussian noise with total power mean(|y|^2)/10^(snr/10), split evenly re/im
    (`awgn(Y,30,'measured')`, `main_recon_tsmis_FFT.m:243`; MathWorks, seeded here)."""
    y = np.asarray(y, dtype=np.complex128)
    p = float(np.mean(np.abs(y) ** 2)) / (10.0 ** (snr_db / 10.0))
    g = normal(0xA5A5 + 7919 * seed, 2 * y.size)
    return y + np.sqrt(p / 2.0) * (g[: y.size] + 1j * g[y.size:]).reshape(y.shape)


# ----------------------------------------------------------------------------------------------------
# denoiser weights (state-dict order of UNetRes, network_unet.py:68-117)
# ----------------------------------------------------------------------------------------------------
def unetres_weight_shapes(in_nc: int = 10, out_nc: int = 10, nc=(64, 128, 256, 512), nb: int = 4):
    """[(name, shape)] in `state_dict()` order: Conv2d OIHW, ConvTranspose2d IOHW."""
    shapes = [("m_head.weight", (nc[0], in_nc, 3, 3))]
    for lvl in range(3):
        c = nc[lvl]
        for b in range(nb):
            shapes.append((f"m_down{lvl + 1}.{b}.res.0.weight", (c, c, 3, 3)))
            shapes.append((f"m_down{lvl + 1}.{b}.res.2.weight", (c, c, 3, 3)))
        shapes.append((f"m_down{lvl + 1}.{nb}.weight", (nc[lvl + 1], c, 2, 2)))
    for b in range(nb):
        shapes.append((f"m_body.{b}.res.0.weight", (nc[3], nc[3], 3, 3)))
        shapes.append((f"m_body.{b}.res.2.weight", (nc[3], nc[3], 3, 3)))
    for lvl in (3, 2, 1):
        cin, cout = nc[lvl], nc[lvl - 1]
        shapes.append((f"m_up{lvl}.0.weight", (cin, cout, 2, 2)))
        for b in range(1, nb + 1):
            shapes.append((f"m_up{lvl}.{b}.res.0.weight", (cout, cout, 3, 3)))
            shapes.append((f"m_up{lvl}.{b}.res.2.weight", (cout, cout, 3, 3)))
    shapes.append(("m_tail.weight", (out_nc, nc[0], 3, 3)))
    return shapes


def unetres_weight_slice(name: str, in_nc=10, out_nc=10, nc=(64, 128, 256, 512), nb=4) -> slice:
    """Position of one state-dict tensor inside the flat weight blob."""
    off = 0
    for nm, shp in unetres_weight_shapes(in_nc, out_nc, nc, nb):
        n = int(np.prod(shp))
        if nm == name:
            return slice(off, off + n)
        off += n
    raise KeyError(name)


def unetres_nparams(in_nc=10, out_nc=10, nc=(64, 128, 256, 512), nb=4) -> int:
    return int(sum(int(np.prod(s)) for _, s in unetres_weight_shapes(in_nc, out_nc, nc, nb)))


def random_weights(in_nc=10, out_nc=10, nc=(64, 128, 256, 512), nb=4, seed=1, gain=1.0) -> np.ndarray:
    """Procedural-random flat fp32 weights: tensor i ~ U(-a,a), a = gain*sqrt(3/fan_in) (variance-preserving
    through ReLU pairs only loosely; meant for single-pass kernel parity, not for ADMM runs)."""
    out = []
    for i, (name, shp) in enumerate(unetres_weight_shapes(in_nc, out_nc, nc, nb)):
        n = int(np.prod(shp))
        if ".0.weight" in name and name.startswith("m_up"):
            fan_in = shp[0]                       # each output pixel of a 2x2/s2 transposed conv sees Cin inputs
        else:
            fan_in = shp[1] * shp[2] * shp[3]
        a = gain * np.sqrt(3.0 / fan_in)
        u = uniform01(seed * 1000003 + i, n)
        out.append(((2.0 * u - 1.0) * a).astype(np.float32))
    return np.concatenate(out)


def golden224_input(in_nc: int = 10, seed: int = 9300) -> np.ndarray:
    """[in_nc][224][224] float32 input of the random-weight 224 x 224 golden vectors (tools/gen_golden.py,
    `unetres_full_224_random_*`): uniform [0,1) image channels; with 11 channels the last plane is the constant
    noise map 0.01 (`PnP_ADMM.m:132`, `build_noise_map.m:19`)."""
    x = uniform01(seed + in_nc, in_nc * 224 * 224).astype(np.float32).reshape(in_nc, 224, 224)
    if in_nc == 11:
        x[10] = np.float32(0.01)
    return x


def structured_weights(in_nc=10, out_nc=10, nc=(64, 128, 256, 512), nb=4, seed=2, eps=0.02) -> np.ndarray:
    """Structured-synthetic denoiser (SURVEY.md section 7 "hard parts"): head = [1 2 1]^2/16 blur of input
    channel c into feature channel c, tail = pick feature channel c back, everything else eps * random.
    The resulting map is a mild linear smoother plus a small perturbation through every layer, so 100 ADMM
    iterations stay bounded while every kernel is exercised."""
    w = []
    blur = np.outer([1.0, 2.0, 1.0], [1.0, 2.0, 1.0]) / 16.0
    for i, (name, shp) in enumerate(unetres_weight_shapes(in_nc, out_nc, nc, nb)):
        n = int(np.prod(shp))
        fan_in = shp[0] if (name.startswith("m_up") and name.endswith(".0.weight")) else shp[1] * shp[2] * shp[3]
        r = ((2.0 * uniform01(seed * 7368787 + i, n) - 1.0) * eps * np.sqrt(3.0 / fan_in)).reshape(shp)
        if name == "m_head.weight":
            for c in range(min(out_nc, shp[0], shp[1])):
                r[c, c] += blur
        elif name == "m_tail.weight":
            for c in range(min(out_nc, shp[1])):
                r[c, c, 1, 1] += 1.0
        w.append(r.astype(np.float32).ravel())
    return np.concatenate(w)


# ----------------------------------------------------------------------------------------------------
# one complete synthetic case
# ----------------------------------------------------------------------------------------------------
def make_case(N=224, T=200, s=10, K=(128, 64), slice_seed=0, dict_seed=0):
    """Dictionary + phantom + ground-truth TSMI for one slice (no operator / measurements: the caller
    builds those through the product API or through the oracle)."""
    dic = make_dictionary(T=T, n_t1=K[0], n_t2=K[1], s=s, seed=dict_seed)
    q = make_phantom_qmaps(N, seed=slice_seed)
    X0 = synthesize_tsmi(q, dic)
    return dic, q, X0
