"""CPU: the MATLAB-sourced oracle rows against an independent, literal restatement of the .m lines.

tests/golden/matlab_rows_*.npz come from tools/gen_matlab_rows.py, which builds the masks and the sparse matrix P exactly as
setup_subsampling_spiralgrided.m:7-42 / setup_subsampling_epi.m:20-35 write them (dense mask, fftshift, find, sparse, kron, vertical
concatenation) and applies main_recon_tsmis_FFT.m:228-229 with numpy's fft2 -- no code or data structure in common with the oracle.
Indices must agree bit for bit, values to 1e-13 relative (both fp64; the two FFTs and summation orders differ)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

CASES = ["spiral_32", "epi_32", "spiral_64", "spiral_224", "epi_224"]


def _seeded_complex(synth, seed, shape):
    n = int(np.prod(shape))
    return ((synth.uniform01(seed, n) - 0.5) + 1j * (synth.uniform01(seed + 7919, n) - 0.5)).reshape(shape, order="F")


@pytest.mark.parametrize("name", CASES)
def test_masks_and_operator_match_the_literal_restatement(oracle, synth, name):
    g = np.load(os.path.join(GOLDEN, f"matlab_rows_{name}.npz"))
    N, T, s, seed = int(g["N"]), int(g["T"]), int(g["s"]), int(g["seed"])
    if str(g["pattern"]) == "spiral":
        fp, k = oracle.spiral_mask(N, int(g["param"]), T)
    else:
        fp, k = oracle.epi_mask(N, N, float(g["param"]), T)
    assert np.array_equal(fp, g["frame_ptr"]) and np.array_equal(k, g["kidx"])          # index work: bit for bit
    assert int(g["nnz"]) == k.size * s                                                    # one V(t,c) per (sample, channel) in P
    op = oracle.Operator(N, N, g["V"], fp, k)
    x = _seeded_complex(synth, seed + 1, (N, N, s))
    y = _seeded_complex(synth, seed + 2, (k.size,))
    Ax, Aty = op.forward(x), op.adjoint(y)
    sy, sx = int(g["stride_y"]), int(g["stride_x"])
    assert np.abs(Ax[::sy] - g["Ax"]).max() <= 1e-13 * np.abs(g["Ax"]).max()
    assert np.abs(Aty.ravel(order="F")[::sx] - g["Aty"]).max() <= 1e-13 * np.abs(g["Aty"]).max()
    assert abs(np.linalg.norm(Ax) - float(g["Ax_norm"])) <= 1e-13 * float(g["Ax_norm"])
    assert abs(np.linalg.norm(Aty) - float(g["Aty_norm"])) <= 1e-13 * float(g["Aty_norm"])


def test_host_mask_builders_match_the_literal_restatement():
    """The product's host-side integer builders (qmri_build_spiral / qmri_build_epi) against the same fixtures."""
    from qmri_pnp_recon_poc_amd import engine
    for name in CASES:
        g = np.load(os.path.join(GOLDEN, f"matlab_rows_{name}.npz"))
        N, T = int(g["N"]), int(g["T"])
        fp, k = (engine.build_spiral(N, int(g["param"]), T) if str(g["pattern"]) == "spiral" else engine.build_epi(N, N, float(g["param"]), T))
        assert np.array_equal(fp, g["frame_ptr"]) and np.array_equal(k, g["kidx"])


def test_dictionary_match_against_the_literal_restatement(oracle):
    """mrf_dtm_cpu.m:91-96 restated with numpy in single precision (max over abs(ip), first index among equal magnitudes)."""
    g = np.load(os.path.join(GOLDEN, "matlab_rows_dictmatch.npz"))
    K = g["D"].shape[0]
    lut = np.stack([np.arange(K), -np.arange(K)], axis=1).astype(np.float32)
    o = oracle.dict_match(g["x"], g["D"], g["normD"], lut)
    assert np.array_equal(o["dm"], g["dm"])                                # random data: no magnitude ties, same atoms
    assert np.allclose(o["mt"], g["mt"], rtol=2e-6) and np.allclose(o["pd"], g["pd"], rtol=2e-5, atol=1e-7)
    assert o["dm"][3] == 18 and o["dm"][4] == 251 and o["dm"][5] == 1      # exact atoms; the all-zero pixel ties at 0 -> first atom


def test_dictionary_match_magnitude_tie_goes_to_the_first_atom(oracle):
    """Two atoms whose |ip|^2 differ in the last bit of single precision while abs(ip) is the same single: max(abs(ip)) ties and
    MATLAB returns the first index (mrf_dtm_cpu.m:92).  An argmax of |ip|^2 would return the later atom B."""
    g = np.load(os.path.join(GOLDEN, "matlab_rows_dictmatch.npz"))
    ones, lut = np.ones(6, np.float32), np.zeros((6, 2), np.float32)
    Dt = g["Dt"]
    re, im = np.float32(Dt[2, 0]), np.float32(Dt[2, 1])
    m2_b = np.float32(np.float32(re * re) + np.float32(im * im))          # = fmaf(im, im, re*re) here: im*im is exact in single
    assert m2_b > np.float32(1.0) and np.sqrt(m2_b, dtype=np.float32) == np.float32(1.0)    # larger square, equal magnitude
    o = oracle.dict_match(g["xt"], Dt, ones, lut)
    assert o["dm"][0] == int(g["dm_t"][0]) == 2 and o["mt"][0] == np.float32(1.0)
    o2 = oracle.dict_match(g["xt"], g["Dt2"], ones, lut)
    assert o2["dm"][0] == int(g["dm_t2"][0]) == 1
