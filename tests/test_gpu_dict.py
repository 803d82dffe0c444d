"""GPU parity: dictionary match through the C ABI vs the CPU oracle (bit-exact atom indices)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K", [(128, 64), (37, 5)])
def test_dict_match_bit_exact(engine_mod, oracle, synth, case224, K):
    dic = synth.make_dictionary(T=200, n_t1=K[0], n_t2=K[1])
    X = synth.synthesize_tsmi(case224["q"], dic)
    rng = np.random.default_rng(3)
    X = X + 0.01 * (rng.standard_normal(X.shape) + 1j * rng.standard_normal(X.shape))
    e = engine_mod.Engine(0)
    e.set_dictionary(dic["D"], dic["normD"], dic["lut"])
    g = e.dict_match(X)
    o = oracle.dict_match(X, dic["D"], dic["normD"], dic["lut"])
    assert np.array_equal(g["dm"], o["dm"])                       # index work: bit-exact
    assert np.array_equal(g["qmap"], o["qmap"])
    assert np.array_equal(g["mt"], o["mt"])
    assert np.array_equal(g["pd"], o["pd"])                         # same bits (no NaNs involved)
    e.close()


def test_dict_match_edge_cases(engine_mod, oracle, synth):
    dic = synth.make_dictionary(T=100, n_t1=9, n_t2=7)
    lut = dic["lut"].copy()
    lut[3, 1] = np.nan                                             # NaN in lut -> 0 (mrf_dtm_cpu.m:138)
    e = engine_mod.Engine(0)
    e.set_dictionary(dic["D"], dic["normD"], lut)
    X = np.zeros((5, 7, 10), np.complex128)                        # all-zero pixels: every |ip| ties at 0 -> first atom
    X[1, 1] = dic["D"][3] * 2.0                                    # exact atom 3 (1-based 4)
    X[2, 2] = -1j * dic["D"][10]
    g = e.dict_match(X)
    o = oracle.dict_match(X, dic["D"], dic["normD"], lut)
    assert np.array_equal(g["dm"], o["dm"])
    assert g["dm"][0, 0] == 1 and g["dm"][1, 1] == 4 and g["dm"][2, 2] == 11
    assert g["qmap"][1, 1, 1] == 0.0
    assert np.array_equal(g["pd"], o["pd"])                         # same bits (no NaNs involved)
    e.close()
