"""Slice batches: the only parallelism the hot path has.

Slices are independent (`PnP_ADMM.m:76-146` touches only its own y, x, v, u), so a batch is sharded over GPUs
with no collective: `shard_slices` is the rank -> slice assignment used by bench.py under torchrun (one process
per GPU), `recon_batch` drives `qmri_recon_batch` (one host thread + one context per device inside one process,
the shape a single MATLAB session needs).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import AdmmParams, NetDesc, Problem


def shard_slices(nslices: int, world: int, rank: int) -> list:
    """Static block partition of slice ids over ranks: contiguous, sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world / rank")
    base, extra = divmod(nslices, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def recon_batch(devices, Y, N, M, V, frame_ptr, kidx, weights, in_nc=10, out_nc=10, nc=(64, 128, 256, 512), nb=4,
                dictionary=None, gamma=0.05, iters=100, cg_tol=1e-4, cg_maxit=100, solver="lsqr", multi_level=False,
                noise_std=0.01, slices_per_launch=1):
    """Reconstruct Y[nslices, m] on the given devices; returns dict(X [nslices,N,M,s], qmap, pd)."""
    L = _lib.lib()
    Y = np.ascontiguousarray(Y, dtype=np.complex128)
    nsl, m = Y.shape
    from .engine import real_dictionary_array
    V = real_dictionary_array(V, "V", np.float64)
    T, s = V.shape
    Vf = np.ascontiguousarray(V.ravel(order="F"))
    fp = np.ascontiguousarray(frame_ptr, dtype=np.int32)
    kk = np.ascontiguousarray(kidx, dtype=np.int32)
    if int(fp[-1]) != m:
        raise ValueError("Y does not match the operator's measurement count")
    w = np.ascontiguousarray(weights, dtype=np.float32)
    desc = NetDesc(0, in_nc, out_nc, (C.c_int32 * 4)(*[int(v) for v in nc]), nb, 0)
    f, d, i32 = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int32)
    pb = Problem()
    pb.N, pb.M, pb.s, pb.T = N, M, s, T
    pb.V, pb.frame_ptr, pb.kidx = Vf.ctypes.data_as(d), fp.ctypes.data_as(i32), kk.ctypes.data_as(i32)
    pb.net = C.pointer(desc)
    pb.weights, pb.weights_nbytes = w.ctypes.data_as(f), w.nbytes
    keep = []
    if dictionary is not None:
        D = np.ascontiguousarray(real_dictionary_array(dictionary["D"], "dict.D", np.float32).ravel(order="F"))
        lut = np.asarray(dictionary["lut"], np.float32)
        lf = np.ascontiguousarray(lut.ravel(order="F"))
        nd = np.ascontiguousarray(dictionary["normD"], dtype=np.float32)
        keep += [D, lf, nd]
        pb.K, pb.Q = int(np.asarray(dictionary["D"]).shape[0]), int(lut.shape[1])
        pb.D, pb.normD, pb.lut = D.ctypes.data_as(f), nd.ctypes.data_as(f), lf.ctypes.data_as(f)
    else:
        pb.K, pb.Q = 0, 0
    pb.admm = AdmmParams(float(gamma), int(iters), float(cg_tol), int(cg_maxit), 0 if solver == "lsqr" else 1,
                         int(bool(multi_level)), float(noise_std), 0)
    pb.slices_per_launch = int(slices_per_launch)
    n = N * M * s
    X = np.empty((nsl, n), np.complex128)
    qmap = np.empty((nsl, N * M * max(pb.Q, 1)), np.float32) if dictionary is not None else None
    pd = np.empty((nsl, N * M), np.complex64) if dictionary is not None else None
    devs = (C.c_int * len(devices))(*[int(v) for v in devices])
    err = C.create_string_buffer(1024)
    st = L.qmri_recon_batch(len(devices), devs, nsl, C.byref(pb), Y.ctypes.data_as(C.c_void_p), X.ctypes.data_as(C.c_void_p),
                            qmap.ctypes.data_as(f) if qmap is not None else None,
                            pd.ctypes.data_as(f) if pd is not None else None, err, len(err))
    if st != 0:
        from .engine import QmriError
        raise QmriError(st, err.value.decode())
    out = {"X": X.reshape((nsl, N, M, s), order="C").copy()}
    out["X"] = np.stack([X[i].reshape((N, M, s), order="F") for i in range(nsl)])
    if dictionary is not None:
        out["qmap"] = np.stack([qmap[i].reshape((N, M, pb.Q), order="F") for i in range(nsl)])
        out["pd"] = np.stack([pd[i].reshape((N, M), order="F") for i in range(nsl)])
    return out
