"""CPU: the oracle's UNetRes restatement against golden vectors produced by the reference's own PyTorch module
(tools/gen_golden.py).  This is what pins the oracle for the denoiser."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_err


@pytest.mark.parametrize("in_nc", [10, 11])
def test_tiny_unetres_matches_reference(oracle, in_nc):
    g = np.load(os.path.join(GOLDEN, f"unetres_tiny_{in_nc}ch.npz"))
    net = oracle.Net(g["weights"], in_nc=in_nc, out_nc=10, nc=tuple(int(v) for v in g["nc"]), nb=int(g["nb"]))
    y = net.forward_f32(g["x"].transpose(1, 2, 0)).transpose(2, 0, 1)          # CHW (torch) <-> HWC (MATLAB dims)
    assert rel_err(y, g["y"]) < 1e-5                                            # fp32 summation-order noise only


def test_state_dict_order_and_procedural_weights(oracle, synth):
    g = np.load(os.path.join(GOLDEN, "unetres_full_64.npz"))
    w = synth.random_weights(seed=1)
    assert w.size == 32648448 == synth.unetres_nparams()                        # network_unet.py:68-117, 10-channel
    assert synth.unetres_nparams(in_nc=11) == 32649024
    assert np.array_equal(w[:5760], g["weights_head"])
    assert abs(float(w.astype(np.float64).sum()) - float(g["weights_sum"])) < 1e-6
    shapes = synth.unetres_weight_shapes()
    assert len(shapes) == 64 and shapes[0][0] == "m_head.weight" and shapes[-1][0] == "m_tail.weight"
    assert shapes[9] == ("m_down1.4.weight", (128, 64, 2, 2)) and shapes[36] == ("m_up3.0.weight", (512, 256, 2, 2))


def test_full_unetres_64_matches_reference(oracle, synth):
    g = np.load(os.path.join(GOLDEN, "unetres_full_64.npz"))
    net = oracle.Net(synth.random_weights(seed=1))
    y = net.forward_f32(g["x"].transpose(1, 2, 0)).transpose(2, 0, 1)
    assert rel_err(y, g["y"]) < 2e-5


def test_full_unetres_224_matches_reference(oracle, synth):
    g = np.load(os.path.join(GOLDEN, "unetres_full_224.npz"))
    net = oracle.Net(synth.structured_weights(seed=int(g["weight_seed"]), eps=float(g["eps"])))
    x = synth.uniform01(int(g["input_seed"]), 10 * 224 * 224).astype(np.float32).reshape(10, 224, 224)
    y = net.forward_f32(x.transpose(1, 2, 0)).transpose(2, 0, 1)
    assert rel_err(y[:, 96:128, 64:96], g["crop"]) < 1e-5
    assert np.abs(y.astype(np.float64).sum(axis=(1, 2)) - g["ch_sum"]).max() / np.abs(g["ch_sum"]).max() < 1e-6
    assert np.abs(np.sqrt((y.astype(np.float64) ** 2).sum(axis=(1, 2))) - g["ch_l2"]).max() / g["ch_l2"].max() < 1e-6
    # G4: positive homogeneity (bias-free + ReLU); the reference's own deviation is recorded in the fixture
    y3 = net.forward_f32(3.0 * x.transpose(1, 2, 0)).transpose(2, 0, 1)
    dev = float(np.abs(y3 - 3.0 * y).max() / np.abs(3.0 * y).max())
    ref_dev = float(np.load(os.path.join(GOLDEN, "unetres_homogeneity.npz"))["rel_dev"])
    assert dev < 10 * max(ref_dev, 1e-6)


def test_denoise_wrapper_casts_and_residual(oracle, synth):
    g = np.load(os.path.join(GOLDEN, "unetres_tiny_10ch.npz"))
    net = oracle.Net(g["weights"], in_nc=10, out_nc=10, nc=tuple(int(v) for v in g["nc"]), nb=int(g["nb"]))
    x = g["x"].transpose(1, 2, 0).astype(np.float64)
    y = net.denoise(x)
    assert y.dtype == np.float64 and rel_err(y.transpose(2, 0, 1), g["y"]) < 1e-5     # denoiseImage_PnP_ADMM.m:111-115
    r = net.denoise(x, residual_noise=True)                                            # :99-104
    assert np.allclose(r, x.astype(np.float32) - y.astype(np.float32), atol=1e-6)
    yb = net.denoise(np.stack([x, 2 * x], axis=3))                                     # H x W x C x N batches (:13-17)
    assert rel_err(yb[..., 0], y) < 1e-7 and rel_err(yb[..., 1], 2 * y) < 1e-5
