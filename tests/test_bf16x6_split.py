"""CPU model of the operand splitting used by the 3x3 conv kernel (conv6_kernels.hip): the three-way bf16 split is
exact and six piece products reproduce an fp32 product sum to fp32 accumulation accuracy."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import bf16x6_check as m  # noqa: E402


def test_split_is_exact():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.standard_normal(4096) * s for s in (1e-6, 1e-3, 1.0, 1e3)]).astype(np.float32)
    a0, a1, a2 = m.split3(x)
    assert np.all(a0.astype(np.float64) + a1.astype(np.float64) + a2.astype(np.float64) == x.astype(np.float64))
    for a in (a0, a1, a2):                                     # every piece is a bf16 value
        assert np.all((a.view(np.uint32) & 0xFFFF) == 0)


def test_six_products_match_fp32_accuracy():
    rng = np.random.default_rng(2)
    A = (rng.standard_normal((64, 576)) * 0.05).astype(np.float32)
    B = np.maximum(rng.standard_normal((576, 128)), 0).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    e6, e32, e3 = m.rel(m.six_product(A, B), ref), m.rel(m.mm32(A, B), ref), m.rel(m.three_product(A, B), ref)
    assert e6 < 5e-7 and e6 < 2.0 * e32 + 1e-7                 # as accurate as an fp32 matmul
    assert e3 > 10 * e6                                         # the three-product shortcut is not


def test_f16_split_and_three_products():
    """The default scheme: x = hi + lo'/2^11 in f16 pieces, products hi*hi + (hi*lo' + lo'*hi)/2^11."""
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.standard_normal(4096) * s for s in (1e-6, 1e-3, 1.0, 1e3)]).astype(np.float32)
    hi, lo = m.split_f16(x)
    back = hi.astype(np.float64) + lo.astype(np.float64) / 2048.0
    # 22 bits + the sign of lo' in the normal range; below 2^-14 the f16 pieces are subnormal and the error is absolute, 2^-36
    assert np.all(np.abs(back - x) <= np.maximum(2.0 ** -22 * np.abs(x), 2.0 ** -36))
    A = (rng.standard_normal((64, 576)) * 0.05).astype(np.float32)
    B = np.maximum(rng.standard_normal((576, 128)), 0).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    e3, e32 = m.rel(m.f16x3_product(A, B), ref), m.rel(m.mm32(A, B), ref)
    assert e3 < 5e-7 and e3 < 1.5 * e32                              # as accurate as an fp32 matmul
    assert m.f16x3_representation_error(A, B) < 1.2e-7               # the dropped lo*lo terms and lo' rounding: fp32 rounding level
    # wide dynamic range inside one operand (weights spanning 8 decades): the scaled residual keeps every element's precision
    A2 = (A * np.exp(rng.standard_normal(A.shape) * 3)).astype(np.float32)
    assert m.f16x3_representation_error(A2, B) < 1.2e-7
