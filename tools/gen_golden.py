#!/usr/bin/env python3
"""Generate golden vectors for the denoiser from the reference's own UNetRes.

Runs ONLY in the build container (needs /root/reference, which never travels).  It imports the reference's
importable PyTorch module `PyTorch_Denoiser/zhang_dpir_testing_code/network_unet.py::UNetRes`, pushes seeded
inputs through it and writes small `.npz` fixtures under tests/golden/.  The fixtures hold DATA only:
weights (flat fp32, state-dict order), inputs and expected outputs.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py

Fixtures
  unetres_tiny_10ch.npz / unetres_tiny_11ch.npz   G1: UNetRes(in,10,nc=[4,8,16,32],nb=4), torch-seeded weights,
                                                  input C x 32 x 32, full output
  unetres_full_64.npz                             G2: full-size UNetRes(10,10,[64,128,256,512],4) with the
                                                  procedural weights of synth.random_weights(seed=1); input
                                                  10 x 64 x 64, full output + L2 norms of x1..x4 / body
  unetres_full_224.npz                            G3: same arch, synth.structured_weights(seed=2), input
                                                  10 x 224 x 224 in [0,1]; per-channel sums / norms + a 32x32 crop
  unetres_full_224_random_{10,11}ch.npz           G3b (round 3): same arch (in_nc 10 / 11), synth.random_weights(seed=1,
                                                  gain=0.7) -- weights under which EVERY level matters (1 % of one layer
                                                  moves the output by > 1e-3); input C x 224 x 224 in [0,1] (11-ch: last
                                                  plane = 0.01, the noise map of PnP_ADMM.m:132); per-channel sums / norms,
                                                  two 32x32 crops, L2 norms of x1..x4 / body / up3..up1 / pre-tail, and the
                                                  output's change when m_body.1.res.2.weight is scaled by 1.01 (the
                                                  sensitivity figure the tests re-check on the oracle)
  unetres_homogeneity.npz                         G4: net(3x) vs 3 net(x) relative deviation of the reference itself
  checkpoint_small.pt (+ checkpoint_small.npz)    G5: a checkpoint in the layout main_train.py:407-411 saves (epoch,
                                                  model_state_dict, optimizer_state_dict after one Adam step, loss) of
                                                  UNetRes(11,10,[4,8,8,16],nb=2), written by torch.save; the .npz holds
                                                  the expected flat weights, an input and the reference's output
  unetres_small_torch_export.onnx (+ .npz)        G6 (round 5): the same small UNetRes written by torch.onnx.export with the arguments of
                                                  export_to_onnx (PyTorch_Denoiser/utils.py:468-481: opset 9, constant folding, input /
                                                  output names, a dynamic batch axis) -- a file from the exporter the reference uses, not
                                                  from tests/onnx_writer.py.  The image has no `onnx` package; the TorchScript exporter
                                                  needs it for ONE step after its C++ serialiser has produced the file's bytes
                                                  (onnx_proto_utils._add_onnxscript_fn: inserts custom onnxscript functions, returns the
                                                  bytes unchanged when there are none, as here), so that step is replaced by the identity
                                                  for the call.  The .npz holds the expected flat weights, an input and the reference's output
  training_pickle_small.npz                       G7 (round 5): the `.mat` -> training-pickle step run by the reference's OWN function
                                                  (PyTorch_Denoiser/main_save_python_tsmis.py:98-205, ready_real_data) on a scratch directory
                                                  of small per-slice `.mat` files (2 subjects x 3 slices, X: 7 x 6 x 4): the slices in the
                                                  order the script sorts them and the arrays it pickled (all channels; first 1; first 2),
                                                  with the file names it chose
`python tools/gen_golden.py [tiny full64 full224 full224random checkpoint onnx pickle]` regenerates a subset.
Tensor layout in the fixtures is PyTorch's [C][H][W]; tests transpose to the MATLAB order.
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/PyTorch_Denoiser")
from zhang_dpir_testing_code.network_unet import UNetRes  # noqa: E402  (reference module, read-only)

from qmri_pnp_recon_poc_amd import synth  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_grad_enabled(False)
torch.set_num_threads(8)


def flat_weights(net):
    return np.concatenate([v.detach().cpu().numpy().astype(np.float32).ravel() for v in net.state_dict().values()])


def load_flat(net, flat):
    off = 0
    sd = net.state_dict()
    for k, v in sd.items():
        n = v.numel()
        sd[k] = torch.from_numpy(flat[off:off + n].reshape(tuple(v.shape)).copy())
        off += n
    assert off == flat.size
    net.load_state_dict(sd)


def check_order(net, in_nc, out_nc, nc, nb):
    names = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    mine = synth.unetres_weight_shapes(in_nc, out_nc, nc, nb)
    assert names == [(n, tuple(s)) for n, s in mine], "state-dict order mismatch"


def tiny(in_nc):
    nc, nb = [4, 8, 16, 32], 4
    torch.manual_seed(1234 + in_nc)
    net = UNetRes(in_nc=in_nc, out_nc=10, nc=nc, nb=nb, act_mode="R", downsample_mode="strideconv",
                  upsample_mode="convtranspose").eval()
    check_order(net, in_nc, 10, nc, nb)
    x = synth.uniform01(77 + in_nc, in_nc * 32 * 32).astype(np.float32).reshape(in_nc, 32, 32)
    y = net(torch.from_numpy(x)[None])[0].numpy()
    np.savez_compressed(os.path.join(OUT, f"unetres_tiny_{in_nc}ch.npz"), weights=flat_weights(net), x=x, y=y,
                        in_nc=in_nc, out_nc=10, nc=np.array(nc), nb=nb)
    print(f"tiny {in_nc}ch: params {flat_weights(net).size}, |y| {np.abs(y).max():.4g}")


def full64():
    nc, nb = [64, 128, 256, 512], 4
    net = UNetRes(in_nc=10, out_nc=10, nc=nc, nb=nb, act_mode="R", downsample_mode="strideconv",
                  upsample_mode="convtranspose").eval()
    check_order(net, 10, 10, nc, nb)
    w = synth.random_weights(10, 10, nc, nb, seed=1)
    assert w.size == 32648448
    load_flat(net, w)
    x = synth.uniform01(4242, 10 * 64 * 64).astype(np.float32).reshape(10, 64, 64)
    xt = torch.from_numpy(x)[None]
    x1 = net.m_head(xt); x2 = net.m_down1(x1); x3 = net.m_down2(x2); x4 = net.m_down3(x3); xb = net.m_body(x4)
    y = net(xt)[0].numpy()
    norms = np.array([float(t.norm()) for t in (x1, x2, x3, x4, xb)])
    np.savez_compressed(os.path.join(OUT, "unetres_full_64.npz"), x=x, y=y, norms=norms, weight_seed=1,
                        weights_head=w[:5760].copy(), weights_sum=float(w.astype(np.float64).sum()))
    print("full64: |y|", np.abs(y).max(), "norms", norms)


def full224():
    nc, nb = [64, 128, 256, 512], 4
    net = UNetRes(in_nc=10, out_nc=10, nc=nc, nb=nb, act_mode="R", downsample_mode="strideconv",
                  upsample_mode="convtranspose").eval()
    w = synth.structured_weights(10, 10, nc, nb, seed=2, eps=0.02)
    load_flat(net, w)
    x = synth.uniform01(9001, 10 * 224 * 224).astype(np.float32).reshape(10, 224, 224)
    y = net(torch.from_numpy(x)[None])[0].numpy()
    np.savez_compressed(os.path.join(OUT, "unetres_full_224.npz"), input_seed=9001, weight_seed=2, eps=0.02,
                        ch_sum=y.astype(np.float64).sum(axis=(1, 2)), ch_l2=np.sqrt((y.astype(np.float64) ** 2).sum(axis=(1, 2))),
                        crop=y[:, 96:128, 64:96].copy())
    # homogeneity of the reference itself: net(3x) == 3 net(x) up to fp32 rounding (bias-free + ReLU)
    y3 = net(torch.from_numpy(3.0 * x)[None])[0].numpy()
    dev = float(np.abs(y3 - 3.0 * y).max() / np.abs(3.0 * y).max())
    np.savez_compressed(os.path.join(OUT, "unetres_homogeneity.npz"), rel_dev=dev)
    print("full224: ch_sum", y.sum(axis=(1, 2))[:3], "homogeneity dev", dev)


def full224_random(in_nc):
    nc, nb = [64, 128, 256, 512], 4
    net = UNetRes(in_nc=in_nc, out_nc=10, nc=nc, nb=nb, act_mode="R", downsample_mode="strideconv",
                  upsample_mode="convtranspose").eval()
    check_order(net, in_nc, 10, nc, nb)
    w = synth.random_weights(in_nc, 10, nc, nb, seed=1, gain=0.7)
    load_flat(net, w)
    x = synth.golden224_input(in_nc)
    xt = torch.from_numpy(x)[None]
    x1 = net.m_head(xt); x2 = net.m_down1(x1); x3 = net.m_down2(x2); x4 = net.m_down3(x3); xb = net.m_body(x4)
    u3 = net.m_up3(xb + x4); u2 = net.m_up2(u3 + x3); u1 = net.m_up1(u2 + x2)      # network_unet.py:106-117
    y = net.m_tail(u1 + x1)[0].numpy()
    assert np.array_equal(y, net(xt)[0].numpy())
    norms = np.array([float(t.double().norm()) for t in (x1, x2, x3, x4, xb, u3, u2, u1, u1 + x1)])
    # sensitivity of the reference itself: one deep layer times 1.01
    sd = net.state_dict()
    sd["m_body.1.res.2.weight"] = sd["m_body.1.res.2.weight"] * 1.01
    net.load_state_dict(sd)
    y2 = net(xt)[0].numpy()
    sens = float(np.linalg.norm(y2 - y) / np.linalg.norm(y))
    y64 = y.astype(np.float64)
    np.savez_compressed(os.path.join(OUT, f"unetres_full_224_random_{in_nc}ch.npz"), in_nc=in_nc, weight_seed=1, gain=0.7,
                        ch_sum=y64.sum(axis=(1, 2)), ch_l2=np.sqrt((y64 ** 2).sum(axis=(1, 2))),
                        crop_a=y[:, 96:128, 64:96].copy(), crop_b=y[:, 0:32, 192:224].copy(),
                        rows=y[:, ::37, ::41].copy(), norms=norms, sens_body_1pct=sens, absmax=float(np.abs(y).max()))
    print(f"full224 random {in_nc}ch: |y| {np.abs(y).max():.4g}, norms {norms}, 1% of m_body.1.res.2 moves y by {sens:.3e}")


def checkpoint():
    in_nc, nc, nb = 11, [4, 8, 8, 16], 2
    torch.manual_seed(4321)
    net = UNetRes(in_nc=in_nc, out_nc=10, nc=nc, nb=nb, act_mode="R", downsample_mode="strideconv",
                  upsample_mode="convtranspose")
    check_order(net, in_nc, 10, nc, nb)
    with torch.enable_grad():                                    # one optimiser step so the Adam state holds tensors
        opt = torch.optim.Adam(params=net.parameters(), lr=1e-3)          # main_train.py:274
        xin = torch.from_numpy(synth.uniform01(55, in_nc * 16 * 16).astype(np.float32).reshape(1, in_nc, 16, 16))
        loss = torch.nn.functional.mse_loss(net(xin), xin[:, :10])
        loss.backward()
        opt.step()
    net.eval()
    torch.save({"epoch": 1, "model_state_dict": net.state_dict(), "optimizer_state_dict": opt.state_dict(),
                "loss": loss.item()}, os.path.join(OUT, "checkpoint_small.pt"))      # main_train.py:407-411
    x = synth.uniform01(56, in_nc * 32 * 32).astype(np.float32).reshape(in_nc, 32, 32)
    y = net(torch.from_numpy(x)[None])[0].numpy()
    np.savez_compressed(os.path.join(OUT, "checkpoint_small.npz"), weights=flat_weights(net), x=x, y=y,
                        in_nc=in_nc, out_nc=10, nc=np.array(nc), nb=nb, loss=loss.item())
    print(f"checkpoint: params {flat_weights(net).size}, file {os.path.getsize(os.path.join(OUT, 'checkpoint_small.pt'))} B")


def onnx_export():
    import warnings
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    onnx_proto_utils._add_onnxscript_fn = lambda model_bytes, custom_opsets: model_bytes     # (see the module docstring, G6)
    in_nc, nc, nb = 11, [4, 8, 8, 16], 2
    torch.manual_seed(8765)
    net = UNetRes(in_nc=in_nc, out_nc=10, nc=nc, nb=nb, act_mode="R", downsample_mode="strideconv", upsample_mode="convtranspose").eval()
    check_order(net, in_nc, 10, nc, nb)
    path = os.path.join(OUT, "unetres_small_torch_export.onnx")
    with torch.enable_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        xe = torch.randn(1, in_nc, 16, 16, requires_grad=True)                              # utils.py:466
        torch.onnx.export(net, xe, path, export_params=True, opset_version=9, do_constant_folding=True, input_names=["input"],
                          output_names=["output"], dynamic_axes={"input": {0: "batch_size"}, "output": {0: "batch_size"}}, dynamo=False)   # utils.py:469-481
    x = synth.uniform01(57, in_nc * 32 * 32).astype(np.float32).reshape(in_nc, 32, 32)
    y = net(torch.from_numpy(x)[None])[0].numpy()
    np.savez_compressed(os.path.join(OUT, "unetres_small_torch_export.npz"), weights=flat_weights(net), x=x, y=y, in_nc=in_nc, out_nc=10,
                        nc=np.array(nc), nb=nb, torch_version=str(torch.__version__))
    print(f"onnx: params {flat_weights(net).size}, file {os.path.getsize(path)} B")


def training_pickle():
    import argparse
    import contextlib
    import io
    import pickle
    import tempfile
    import scipy.io as scio
    import main_save_python_tsmis as ref                      # (the reference's script: definitions only at import, read-only)
    rng = np.random.default_rng(77)
    nvol, nsl = 2, 3
    X = rng.random((nvol, nsl, 7, 6, 4))                        # [subject][slice] N x M x C, float64 like the script's data['X']
    out = {}
    with tempfile.TemporaryDirectory() as td:
        tr, te = os.path.join(td, "in_train") + os.sep, os.path.join(td, "in_test") + os.sep
        os.makedirs(tr); os.makedirs(te)
        for v in range(nvol):
            for sl in range(nsl):                               # subject 1 -> training input, subject 2 -> testing input (matlab_train_test_split = 2)
                scio.savemat((tr if v == 0 else te) + f"vol{v + 1}s{sl + 1}.mat", {"X": X[v, sl]})
        for tag, kw in (("all", dict(select_channels=False)), ("first1", dict(select_channels=True, channels_to_save=1)),
                        ("first2", dict(select_channels=True, channels_to_save=2))):
            otr, ote = os.path.join(td, "out_train_" + tag) + os.sep, os.path.join(td, "out_test_" + tag) + os.sep
            os.makedirs(otr); os.makedirs(ote)
            args = argparse.Namespace(scan_type="fisp", cut=3, num_slices=nsl, training_data_input_path=tr, testing_data_input_path=te,
                                      training_data_output_path=otr, testing_data_output_path=ote, matlab_train_test_split=2, python_train_test_split=2)
            with contextlib.redirect_stdout(io.StringIO()):
                ref.ready_real_data(args, **kw)
            names = []
            for d in (otr, ote):
                for fn in sorted(os.listdir(d)):
                    with open(d + fn, "rb") as f:
                        out[f"{tag}_{fn}"] = pickle.load(f)
                    names.append(("train/" if d == otr else "test/") + fn)
            out[tag + "_files"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "training_pickle_small.npz"), X=X, **out)
    print("pickle:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


if __name__ == "__main__":
    todo = sys.argv[1:] or ["tiny", "full64", "full224", "full224random", "checkpoint", "onnx", "pickle"]
    if "tiny" in todo:
        tiny(10)
        tiny(11)
    if "full64" in todo:
        full64()
    if "full224" in todo:
        full224()
    if "full224random" in todo:
        full224_random(10)
        full224_random(11)
    if "checkpoint" in todo:
        checkpoint()
    if "onnx" in todo:
        onnx_export()
    if "pickle" in todo:
        training_pickle()
