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


@pytest.mark.parametrize("in_nc", [10, 11])
def test_full_unetres_224_random_weights_matches_reference(oracle, synth, in_nc):
    """G3b: weights under which every level of the network matters (random_weights(seed=1, gain=0.7)); the structured
    weights of G3 leave the 62 interior layers at 1.5e-4 of the output.  The sensitivity is asserted here so that the fixture
    can never go blind: one layer of the 28 x 28 body times 1.01 must move the oracle's output by what it moved the reference's."""
    g = np.load(os.path.join(GOLDEN, f"unetres_full_224_random_{in_nc}ch.npz"))
    w = synth.random_weights(in_nc=in_nc, seed=int(g["weight_seed"]), gain=float(g["gain"]))
    x = synth.golden224_input(in_nc)
    y = oracle.Net(w, in_nc=in_nc).forward_f32(x.transpose(1, 2, 0)).transpose(2, 0, 1)
    assert abs(float(np.abs(y).max()) - float(g["absmax"])) < 1e-3 * float(g["absmax"])
    assert rel_err(y[:, 96:128, 64:96], g["crop_a"]) < 1e-5 and rel_err(y[:, 0:32, 192:224], g["crop_b"]) < 1e-5
    assert rel_err(y[:, ::37, ::41], g["rows"]) < 1e-5
    y64 = y.astype(np.float64)
    assert np.abs(y64.sum(axis=(1, 2)) - g["ch_sum"]).max() / np.abs(g["ch_sum"]).max() < 1e-5
    assert np.abs(np.sqrt((y64 ** 2).sum(axis=(1, 2))) - g["ch_l2"]).max() / g["ch_l2"].max() < 1e-6
    if in_nc == 10:
        w2 = w.copy()
        w2[synth.unetres_weight_slice("m_body.1.res.2.weight", in_nc=in_nc)] *= np.float32(1.01)
        y2 = oracle.Net(w2, in_nc=in_nc).forward_f32(x.transpose(1, 2, 0)).transpose(2, 0, 1)
        sens = rel_err(y2, y)
        assert sens > 1e-4 and abs(sens - float(g["sens_body_1pct"])) < 0.02 * float(g["sens_body_1pct"])


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
