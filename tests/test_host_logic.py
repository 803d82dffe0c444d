"""CPU: host-side logic that needs no GPU -- C-ABI symbols, FFT codelets, synthetic generators, slice sharding."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    from qmri_pnp_recon_poc_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "qmri.h")).read()
    declared = sorted(set(re.findall(r"\b(qmri_[a-z0-9_]+)\s*\(", hdr)))
    assert sorted(_lib.SYMBOLS) == declared, "ctypes symbol table is out of sync with include/qmri.h"
    L = _lib.lib()                                   # builds with hipcc if needed; loads without a GPU
    for s in declared:
        assert hasattr(L, s), f"libqmri.so does not export {s}"
    assert L.qmri_abi_version() == 1


def test_no_gpu_fails_loudly():
    """Without a usable gfx950 device the product refuses to run (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from qmri_pnp_recon_poc_amd import engine
    with pytest.raises(engine.QmriError) as ei:
        engine.Engine(0)
    assert "no HIP device" in str(ei.value) or "gfx950" in str(ei.value) or "failed" in str(ei.value)


def test_host_mask_builders_match_oracle(oracle):
    """qmri_build_spiral / qmri_build_epi are host integer code: bit-exact against the oracle, no GPU needed."""
    from qmri_pnp_recon_poc_amd import engine
    for (N, S, T) in [(224, 771, 200), (32, 120, 24), (64, 50, 7)]:
        fp, k = engine.build_spiral(N, S, T)
        fo, ko = oracle.spiral_mask(N, S, T)
        assert np.array_equal(fp, fo) and np.array_equal(k, ko)
    for (N, M, pct, T) in [(224, 224, 1 / 65, 200), (32, 32, 1 / 8, 40), (8, 4, 1 / 3, 5)]:
        fp, k = engine.build_epi(N, M, pct, T)
        fo, ko = oracle.epi_mask(N, M, pct, T)
        assert np.array_equal(fp, fo) and np.array_equal(k, ko)


def test_fft_codelets_on_host(tmp_path):
    exe = tmp_path / "fft_codelets_test"
    subprocess.run(["g++", "-O2", "-I", os.path.join(ROOT, "qmri_pnp_recon_poc_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "fft_codelets_test.cpp"), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout


def test_synth_generators_are_deterministic(synth):
    a = synth.splitmix64(1234567, 4)
    assert a.tolist() == [6457827717110365317, 3203168211198807973, 9817491932198370423, 4593380528125082431]
    dic = synth.make_dictionary(T=60, n_t1=12, n_t2=8)
    V = dic["V"]
    assert np.abs(V.T @ V - np.eye(10)).max() < 1e-12                  # orthonormal columns (a7 relies on it)
    assert np.allclose(np.linalg.norm(dic["D"], axis=1), 1.0, atol=1e-6)
    q = synth.make_phantom_qmaps(64, seed=3)
    X = synth.synthesize_tsmi(q, dic)
    assert X.shape == (64, 64, 10) and np.all(X[:, :, 0] >= 0)         # sign-aligned to channel 1 (main_synthesize_tsmis.m:97-98)
    assert np.all(X[q[:, :, 2] == 0] == 0)
    y = np.ones(1000, complex)
    n = synth.awgn_measured(y, 30.0, seed=5) - y
    assert abs(10 * np.log10(1.0 / np.mean(np.abs(n) ** 2)) - 30.0) < 0.5   # measured SNR


def test_shard_slices():
    from qmri_pnp_recon_poc_amd.batch import shard_slices
    for nsl, world in [(120, 8), (15, 4), (3, 8), (0, 2)]:
        parts = [shard_slices(nsl, world, r) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(nsl))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert shard_slices(120, 8, 3) == list(range(45, 60))
