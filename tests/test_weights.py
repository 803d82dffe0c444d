"""Weight ingestion: the `.pt` checkpoint and ONNX readers, host side only.

Pinned by: tests/golden/checkpoint_small.pt - written by torch.save from the reference's own UNetRes in the layout
main_train.py:407-411 uses (tools/gen_golden.py checkpoint) - with the expected flat weights in checkpoint_small.npz.
ONNX: tests/golden/unetres_small_torch_export.onnx is a file from torch.onnx.export itself, called on the reference's UNetRes with the
arguments of export_to_onnx (PyTorch_Denoiser/utils.py:468-481; tools/gen_golden.py onnx, round 5) -- both readers are pinned to it;
the encodings an exporter MAY use but this one did not (float_data, packed dims, renamed initializers, other float types, mangled
files) come from tests/onnx_writer.py (schema-built) and are checked against the known blob and between the two readers.
"""
import os
import pickle
import sys
import zipfile

import numpy as np
import pytest

import onnx_writer as ow

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
CKPT = os.path.join(GOLD, "checkpoint_small.pt")


@pytest.fixture(scope="module")
def W():
    from qmri_pnp_recon_poc_amd import weights
    return weights


@pytest.fixture(scope="module")
def ck():
    return np.load(os.path.join(GOLD, "checkpoint_small.npz"))


def arch_of(g):
    return {"in_nc": int(g["in_nc"]), "out_nc": int(g["out_nc"]), "nc": tuple(int(v) for v in g["nc"]), "nb": int(g["nb"])}


# ---- .pt checkpoints --------------------------------------------------------------------------------------
def test_checkpoint_state_dict_matches_reference(W, ck, synth):
    sd = W.read_checkpoint(CKPT)
    a = arch_of(ck)
    want = synth.unetres_weight_shapes(a["in_nc"], a["out_nc"], a["nc"], a["nb"])
    assert [(k, v.shape) for k, v in sd.items()] == [(n, tuple(s)) for n, s in want]
    assert all(v.dtype == np.float32 for v in sd.values())
    blob, arch = W.unetres_blob(sd)
    assert arch == a
    assert np.array_equal(blob, ck["weights"])                           # bit-exact
    blob2, arch2 = W.load_denoiser_weights(CKPT)
    assert arch2 == a and np.array_equal(blob2, blob)


def test_checkpoint_other_entries(W, ck):
    top = W.read_checkpoint(CKPT, everything=True)
    assert set(top) == {"epoch", "model_state_dict", "optimizer_state_dict", "loss"}        # main_train.py:407-411
    assert top["epoch"] == 1 and top["loss"] == float(ck["loss"])
    st = top["optimizer_state_dict"]["state"]
    assert len(st) == len(top["model_state_dict"])
    assert st[0]["exp_avg"].shape == top["model_state_dict"]["m_head.weight"].shape


def test_checkpoint_reader_does_not_need_torch(tmp_path):
    import subprocess
    code = ("import sys; sys.modules['torch'] = None\n"
            "from qmri_pnp_recon_poc_amd import weights as W\n"
            f"b, a = W.load_denoiser_weights({CKPT!r}); print(b.size, a['nb'])\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=os.path.dirname(HERE))
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["22132", "2"]


def _rewrite_pickle(tmp_path, mutate):
    """Copy the checkpoint with data.pkl replaced by pickle.dumps(mutate(decoded top-level object)) - plain containers only."""
    out = str(tmp_path / "m.pt")
    with zipfile.ZipFile(CKPT) as zin, zipfile.ZipFile(out, "w") as zout:
        for n in zin.namelist():
            data = zin.read(n)
            if n.endswith("data.pkl"):
                data = mutate(data)
            zout.writestr(n, data)
    return out


def test_checkpoint_refuses_foreign_globals(W, tmp_path):
    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    bad = _rewrite_pickle(tmp_path, lambda _: pickle.dumps({"model_state_dict": Evil()}, protocol=2))
    with pytest.raises(W.WeightFileError, match="references"):
        W.read_checkpoint(bad)


def test_checkpoint_errors(W, tmp_path):
    p = tmp_path / "x.pt"
    p.write_bytes(b"not a zip at all")
    with pytest.raises(W.WeightFileError, match="not a zip"):
        W.read_checkpoint(str(p))
    nokey = _rewrite_pickle(tmp_path, lambda _: pickle.dumps({"epoch": 3}, protocol=2))
    with pytest.raises(W.WeightFileError, match="model_state_dict"):
        W.read_checkpoint(nokey)


def test_blob_dataparallel_prefix_and_bare_state_dict(W, ck):
    sd = W.read_checkpoint(CKPT)
    from collections import OrderedDict
    blob, arch = W.unetres_blob(OrderedDict(("module." + k, v) for k, v in sd.items()))
    assert np.array_equal(blob, ck["weights"]) and arch == arch_of(ck)


def test_blob_rejects_non_unetres(W, ck):
    sd = W.read_checkpoint(CKPT)
    from collections import OrderedDict
    items = list(sd.items())
    with pytest.raises(W.WeightFileError, match="14\\*nb"):
        W.unetres_blob(OrderedDict(items[:-1]))
    with pytest.raises(W.WeightFileError, match="bias-free"):
        W.unetres_blob(OrderedDict(items + [("m_tail.bias", np.zeros(10, np.float32))]))
    swapped = OrderedDict(items)
    swapped["m_down1.2.weight"] = np.zeros((8, 4, 3, 3), np.float32)
    with pytest.raises(W.WeightFileError, match="expects"):
        W.unetres_blob(swapped)
    renamed = OrderedDict((("m_head.conv.weight" if k == "m_head.weight" else k), v) for k, v in items)
    with pytest.raises(W.WeightFileError, match="named"):
        W.unetres_blob(renamed)
    nonfinite = OrderedDict(items)
    nonfinite["m_tail.weight"] = np.full_like(items[-1][1], np.nan)
    with pytest.raises(W.WeightFileError, match="non-finite"):
        W.unetres_blob(nonfinite)


# ---- ONNX --------------------------------------------------------------------------------------------------
def test_onnx_file_written_by_the_torch_exporter(W, engine_mod, oracle):
    """The file torch.onnx.export wrote for the reference's own UNetRes (opset 9, constant folding, named input / output, dynamic batch axis:
    utils.py:468-481).  Both readers return the state dict's weights bit for bit and recognise the architecture; the CPU restatement of the
    network with those weights reproduces the reference network's output on the fixture's input."""
    g = np.load(os.path.join(GOLD, "unetres_small_torch_export.npz"))
    path = os.path.join(GOLD, "unetres_small_torch_export.onnx")
    a = arch_of(g)
    m = W.read_onnx(path)
    assert m["opset"] == 9 and m["inputs"] == ["input"] and m["outputs"] == ["output"]
    assert len(m["convs"]) == 14 * a["nb"] + 8
    blob, arch = W.load_denoiser_weights(path)
    assert arch == a and np.array_equal(blob, g["weights"])
    nblob, narch = engine_mod.read_onnx_unetres(path)                   # libqmri's own reader (C ABI, host only)
    assert narch == a and np.array_equal(nblob, g["weights"])
    net = oracle.Net(nblob, in_nc=a["in_nc"], out_nc=a["out_nc"], nc=a["nc"], nb=a["nb"])
    y = net.forward_f32(g["x"].transpose(1, 2, 0)).transpose(2, 0, 1)          # CHW (torch) <-> HWC (MATLAB dims)
    err = np.linalg.norm(y.astype(np.float64) - g["y"]) / np.linalg.norm(g["y"])
    assert err < 2e-5, err


ONNX_VARIANTS = [
    dict(mode="raw"),                                                # what torch.onnx.export writes
    dict(mode="raw", packed_dims=True, initializers_first=True),
    dict(mode="float_data"),
    dict(mode="float_data_unpacked"),
    dict(mode="raw", names="renamed"),                               # constant folding may rename initializers
]


@pytest.fixture(scope="module")
def small_ws(ck):
    a = arch_of(ck)
    return a, ow.split_blob(ck["weights"], a["in_nc"], a["out_nc"], a["nc"], a["nb"])


@pytest.mark.parametrize("variant", ONNX_VARIANTS)
def test_onnx_python_and_native_readers(W, engine_mod, ck, small_ws, tmp_path, variant):
    a, ws = small_ws
    path = str(tmp_path / "net.onnx")
    with open(path, "wb") as f:
        f.write(ow.unetres_model(ws, a["in_nc"], a["out_nc"], a["nc"], a["nb"], **variant))
    m = W.read_onnx(path)
    assert m["opset"] == 9 and m["inputs"] == ["input"] and m["outputs"] == ["output"]       # utils.py:476-478
    assert len(m["convs"]) == 14 * a["nb"] + 8
    blob, arch = W.load_denoiser_weights(path)
    assert arch == a and np.array_equal(blob, ck["weights"])
    nblob, narch = engine_mod.read_onnx_unetres(path)                   # libqmri's own reader, no GPU involved
    assert narch == a and np.array_equal(nblob, ck["weights"])


@pytest.mark.parametrize("dtype,mode", [(np.float16, "raw"), (np.float64, "raw"), (np.float64, "double_data")])
def test_onnx_other_float_types(W, engine_mod, small_ws, tmp_path, dtype, mode):
    a, ws = small_ws
    cast = [w.astype(dtype) for w in ws]
    path = str(tmp_path / "net.onnx")
    with open(path, "wb") as f:
        f.write(ow.unetres_model(cast, a["in_nc"], a["out_nc"], a["nc"], a["nb"], mode=mode))
    want = np.concatenate([w.astype(np.float32).ravel() for w in cast])
    blob, _ = W.load_denoiser_weights(path)
    nblob, _ = engine_mod.read_onnx_unetres(path)
    assert np.array_equal(blob, want) and np.array_equal(nblob, want)


def test_onnx_full_size_architecture(W, engine_mod, synth, tmp_path):
    """The 11-channel full-size net (main_test.py:251): 32.6 M parameters, 130 MB file."""
    nc, nb = (64, 128, 256, 512), 4
    blob = synth.random_weights(11, 10, nc, nb, seed=3)
    path = str(tmp_path / "full.onnx")
    with open(path, "wb") as f:
        f.write(ow.unetres_model(ow.split_blob(blob, 11, 10, nc, nb), 11, 10, nc, nb, hw=224))
    got, arch = engine_mod.read_onnx_unetres(path)
    assert arch == {"in_nc": 11, "out_nc": 10, "nc": nc, "nb": nb}
    assert np.array_equal(got, blob)
    got2, arch2 = W.load_denoiser_weights(path)
    assert arch2 == arch and np.array_equal(got2, blob)


def test_onnx_errors(W, engine_mod, small_ws, tmp_path):
    a, ws = small_ws
    good = ow.unetres_model(ws, a["in_nc"], a["out_nc"], a["nc"], a["nb"])

    def both(data, match_py, code):
        p = str(tmp_path / "bad.onnx")
        with open(p, "wb") as f:
            f.write(data)
        with pytest.raises(W.WeightFileError, match=match_py):
            W.load_denoiser_weights(p)
        with pytest.raises(engine_mod.QmriError) as ei:
            engine_mod.read_onnx_unetres(p)
        assert ei.value.code == code
    both(ow.unetres_model(ws, a["in_nc"], a["out_nc"], a["nc"], a["nb"], bias_on=3), "bias", -4)
    both(good[: len(good) // 2], "past the end|truncated|initializers|UNetRes", -4)          # cut in the middle
    both(ow.f_varint(1, 6) + ow.f_bytes(2, "pytorch"), "GraphProto", -4)                      # a model without a graph
    # one ResBlock short: 14*nb + 8 no longer holds
    short = ow.unetres_model(ws, a["in_nc"], a["out_nc"], a["nc"], a["nb"])
    both(short.replace(b"ConvTranspose", b"ConvTranspos_"), "14\\*nb|UNetRes", -4)            # op types no longer recognised
    # a 3x3 tensor where the strided 2x2 belongs
    wrong = list(ws)
    wrong[2 * a["nb"] + 1] = np.zeros((a["nc"][1], a["nc"][0], 3, 3), np.float32)
    both(ow.unetres_model(wrong, a["in_nc"], a["out_nc"], a["nc"], a["nb"]), "expects", -4)
    with pytest.raises(engine_mod.QmriError) as ei:
        engine_mod.read_onnx_unetres(str(tmp_path / "does_not_exist.onnx"))
    assert ei.value.code == -1
