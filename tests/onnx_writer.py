"""Test helper: assemble an ONNX ModelProto byte string for a UNetRes the way torch.onnx.export (opset 9,
export_params=True; PyTorch_Denoiser/utils.py:470-481) lays it out - Conv / Relu / Add / ConvTranspose nodes in
execution order, the weights as graph initializers named after the state-dict keys, graph input 'input', output
'output'.  The `onnx` package is not installed in this image and torch's exporter refuses to run without it, so the
files the weight readers are tested on are written here from the published onnx.proto3 schema (field numbers below);
no exporter-written file exists to pin against.

Wire format: tag = (field_number << 3) | wire_type; 0 = varint, 2 = length-delimited, 5 = fixed32.
"""
import struct

import numpy as np


def varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def f_varint(no, v):
    return varint(no << 3) + varint(v)


def f_bytes(no, b):
    if isinstance(b, str):
        b = b.encode()
    return varint((no << 3) | 2) + varint(len(b)) + bytes(b)


def tensor(name, arr, mode="raw", packed_dims=False):
    """TensorProto: dims=1, data_type=2, float_data=4, name=8, raw_data=9, double_data=10."""
    arr = np.asarray(arr)
    dt = {np.dtype("float32"): 1, np.dtype("float16"): 10, np.dtype("float64"): 11, np.dtype("int64"): 7}[arr.dtype]
    if packed_dims:
        out = f_bytes(1, b"".join(varint(d) for d in arr.shape))
    else:
        out = b"".join(f_varint(1, d) for d in arr.shape)
    out += f_varint(2, dt)
    if mode == "raw":
        out += f_bytes(8, name) + f_bytes(9, arr.astype(arr.dtype.newbyteorder("<")).tobytes())
    elif mode == "float_data":                                   # packed repeated float
        assert dt == 1
        out += f_bytes(4, arr.astype("<f4").tobytes()) + f_bytes(8, name)
    elif mode == "float_data_unpacked":                          # one fixed32 field per element
        assert dt == 1
        out += b"".join(varint((4 << 3) | 5) + struct.pack("<f", v) for v in arr.ravel()) + f_bytes(8, name)
    elif mode == "double_data":
        assert dt == 11
        out += f_bytes(10, arr.astype("<f8").tobytes()) + f_bytes(8, name)
    else:
        raise ValueError(mode)
    return out


def attr_ints(name, vals):
    """AttributeProto: name=1, ints=8, type=20 (INTS=7)."""
    return f_bytes(1, name) + b"".join(f_varint(8, v) for v in vals) + f_varint(20, 7)


def attr_int(name, v):
    return f_bytes(1, name) + f_varint(3, v) + f_varint(20, 2)


def node(op, ins, outs, attrs=()):
    """NodeProto: input=1, output=2, name=3, op_type=4, attribute=5."""
    out = b"".join(f_bytes(1, s) for s in ins) + b"".join(f_bytes(2, s) for s in outs)
    out += f_bytes(3, f"{op}_{outs[0]}") + f_bytes(4, op)
    return out + b"".join(f_bytes(5, a) for a in attrs)


def value_info(name, shape):
    """ValueInfoProto: name=1, type=2 -> TypeProto.tensor_type=1 -> elem_type=1, shape=2 -> dim=1 -> dim_value=1 / dim_param=2."""
    dims = b"".join(f_bytes(1, f_bytes(2, d) if isinstance(d, str) else f_varint(1, d)) for d in shape)
    return f_bytes(1, name) + f_bytes(2, f_bytes(1, f_varint(1, 1) + f_bytes(2, dims)))


def unetres_model(weights, in_nc, out_nc, nc, nb, hw=32, mode="raw", packed_dims=False, names="state_dict", bias_on=None,
                  initializers_first=False, extra_int_initializer=True):
    """ModelProto bytes.  weights: list of arrays in state-dict order (synth.unetres_weight_shapes)."""
    from qmri_pnp_recon_poc_amd.synth import unetres_weight_shapes
    spec = unetres_weight_shapes(in_nc, out_nc, tuple(nc), nb)
    assert len(spec) == len(weights)
    wname = [n if names == "state_dict" else f"onnx::Conv_{100 + i}" for i, (n, _) in enumerate(spec)]
    nodes, inits, cnt = [], [], [0]

    def fresh():
        cnt[0] += 1
        return str(cnt[0])

    it = iter(range(len(spec)))

    def conv(x, k, stride, transposed=False):
        i = next(it)
        ins = [x, wname[i]]
        inits.append(tensor(wname[i], weights[i], mode, packed_dims))
        if bias_on == i:
            bname = wname[i].replace("weight", "bias")
            inits.append(tensor(bname, np.zeros(weights[i].shape[1 if transposed else 0], np.float32), mode, packed_dims))
            ins.append(bname)
        y = fresh()
        pad = 1 if k == 3 else 0
        nodes.append(node("ConvTranspose" if transposed else "Conv", ins, [y],
                          [attr_ints("dilations", [1, 1]), attr_int("group", 1), attr_ints("kernel_shape", [k, k]),
                           attr_ints("pads", [pad] * 4), attr_ints("strides", [stride, stride])]))
        return y

    def resblock(x):
        a = conv(x, 3, 1)
        r = fresh()
        nodes.append(node("Relu", [a], [r]))
        b = conv(r, 3, 1)
        y = fresh()
        nodes.append(node("Add", [x, b], [y]))
        return y

    x1 = conv("input", 3, 1)
    skips, x = [x1], x1
    for lvl in range(3):
        for _ in range(nb):
            x = resblock(x)
        x = conv(x, 2, 2)
        skips.append(x)
    for _ in range(nb):
        x = resblock(x)
    for lvl in (3, 2, 1):
        s = fresh()
        nodes.append(node("Add", [x, skips[lvl]], [s]))
        x = conv(s, 2, 2, transposed=True)
        for _ in range(nb):
            x = resblock(x)
    s = fresh()
    nodes.append(node("Add", [x, skips[0]], [s]))
    i = next(it)
    inits.append(tensor(wname[i], weights[i], mode, packed_dims))
    nodes.append(node("Conv", [s, wname[i]], ["output"], [attr_ints("kernel_shape", [3, 3]), attr_ints("pads", [1] * 4)]))
    if extra_int_initializer:                                   # exporters leave integer constants (pads, shapes) around
        inits.append(tensor("onnx::Pad_7", np.array([0, 0, 1, 1], np.int64)))
    # GraphProto: node=1, name=2, initializer=5, input=11, output=12
    body_nodes = b"".join(f_bytes(1, n) for n in nodes)
    body_inits = b"".join(f_bytes(5, t) for t in inits)
    graph = (body_inits + body_nodes) if initializers_first else (body_nodes + f_bytes(2, "torch-jit-export") + body_inits)
    graph += f_bytes(11, value_info("input", ["batch_size", in_nc, hw, hw])) + f_bytes(12, value_info("output", ["batch_size", out_nc, hw, hw]))
    # ModelProto: ir_version=1, producer_name=2, producer_version=3, graph=7, opset_import=8 (domain=1, version=2)
    return f_varint(1, 6) + f_bytes(2, "pytorch") + f_bytes(3, "1.7") + f_bytes(7, graph) + f_bytes(8, f_varint(2, 9))


def split_blob(blob, in_nc, out_nc, nc, nb):
    from qmri_pnp_recon_poc_amd.synth import unetres_weight_shapes
    out, off = [], 0
    for _, shp in unetres_weight_shapes(in_nc, out_nc, tuple(nc), nb):
        n = int(np.prod(shp))
        out.append(np.asarray(blob[off:off + n], np.float32).reshape(shp))
        off += n
    assert off == len(blob)
    return out
