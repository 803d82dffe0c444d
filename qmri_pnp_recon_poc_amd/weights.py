"""Weight ingestion for the UNetRes denoiser: the two files the reference's training kit writes -> the flat fp32 blob
`qmri_set_denoiser` takes (include/qmri.h), with the architecture read off the tensor shapes.

  read_checkpoint(path)   the `.pt` dict `main_train.py:407-411 / 431-435` saves with torch.save
                          ({'epoch', 'model_state_dict', 'optimizer_state_dict', 'loss'}); `main_test.py:260-261` is the
                          reference-side reader (torch.load + load_state_dict).
  read_onnx(path)         the ONNX file `utils.py:468-481 export_to_onnx` writes (opset 9, input 'input', output
                          'output', weights as graph initializers); the reference-side reader is MATLAB's
                          importONNXNetwork (main_recon_tsmis_FFT.m:138).  The native twin of this function is
                          `qmri_onnx_read_unetres` in the shared library (csrc/onnx_reader.cpp) for the MEX route.
  unetres_blob(tensors)   orders / validates the tensors as UNetRes (network_unet.py:164-211: bias-free convs, head,
                          3 x (nb ResBlocks + strideconv 2x2), body, 3 x (convtranspose 2x2 + nb ResBlocks), tail) and
                          returns (flat fp32 weights, dict(in_nc,out_nc,nc,nb)).

Neither reader imports torch or onnx: a checkpoint is a zip of raw little-endian storages plus a pickle that is decoded
with a closed allow-list (nothing from the file is ever executed), an ONNX file is walked as protobuf wire format using
the handful of field numbers of onnx.proto3 quoted below.  Host logic only - no GPU, no oracle.
"""
from __future__ import annotations

import io
import pickle
import struct
import zipfile
from collections import OrderedDict

import numpy as np

__all__ = ["read_checkpoint", "read_onnx", "unetres_blob", "load_denoiser_weights", "WeightFileError"]


class WeightFileError(ValueError):
    """The file is not something the reference's training kit writes, or does not hold a UNetRes."""


# ------------------------------------------------------------------------------------------------------------
# torch.save zip checkpoints
# ------------------------------------------------------------------------------------------------------------
_STORAGE_DTYPES = {
    "FloatStorage": np.dtype("<f4"), "DoubleStorage": np.dtype("<f8"), "HalfStorage": np.dtype("<f2"),
    "LongStorage": np.dtype("<i8"), "IntStorage": np.dtype("<i4"), "ShortStorage": np.dtype("<i2"),
    "CharStorage": np.dtype("i1"), "ByteStorage": np.dtype("u1"), "BoolStorage": np.dtype("?"),
    "BFloat16Storage": np.dtype("<u2"),          # kept as raw 16-bit words; widened in _to_f32
}


class _StorageType:
    def __init__(self, name):
        self.name = name
        self.dtype = _STORAGE_DTYPES[name]


class _LazyStorage:
    def __init__(self, zf, prefix, stype, key, numel):
        self.zf, self.prefix, self.stype, self.key, self.numel = zf, prefix, stype, key, int(numel)

    def array(self):
        raw = self.zf.read(f"{self.prefix}/data/{self.key}")
        a = np.frombuffer(raw, dtype=self.stype.dtype)
        if a.size < self.numel:
            raise WeightFileError(f"storage {self.key}: {a.size} elements in the archive, {self.numel} declared")
        return a


class _Tensor:
    """What `_rebuild_tensor_v2` leaves behind: a view description, materialised on demand."""

    def __init__(self, storage, offset, size, stride):
        self.storage, self.offset, self.size, self.stride = storage, int(offset), tuple(int(v) for v in size), tuple(int(v) for v in stride)

    def numpy(self):
        base = self.storage.array()
        if not self.size:
            return base[self.offset:self.offset + 1].reshape(()).copy()
        n = int(np.prod(self.size))
        if n == 0:
            return np.zeros(self.size, base.dtype)
        last = self.offset + sum((d - 1) * s for d, s in zip(self.size, self.stride))
        if self.offset < 0 or last >= base.size or any(s < 0 for s in self.stride):
            raise WeightFileError("tensor view reaches outside its storage")
        item = base.dtype.itemsize
        v = np.lib.stride_tricks.as_strided(base[self.offset:], shape=self.size, strides=tuple(s * item for s in self.stride), writeable=False)
        a = np.ascontiguousarray(v)
        if self.storage.stype.name == "BFloat16Storage":
            a = (a.astype(np.uint32) << 16).view(np.float32)
        return a


def _rebuild_tensor_v2(storage, storage_offset, size, stride, requires_grad=False, backward_hooks=None, metadata=None):
    return _Tensor(storage, storage_offset, size, stride)


def _rebuild_parameter(data, requires_grad=False, backward_hooks=None):
    return data


class _CheckpointUnpickler(pickle.Unpickler):
    """Unpickler with a closed allow-list: containers, tensor rebuild stubs, storage type tags. Anything else raises."""

    def __init__(self, f, zf, prefix):
        super().__init__(f)
        self.zf, self.prefix = zf, prefix

    def find_class(self, module, name):
        if module == "collections" and name == "OrderedDict":
            return OrderedDict
        if module == "torch._utils" and name in ("_rebuild_tensor_v2", "_rebuild_tensor"):
            return _rebuild_tensor_v2
        if module == "torch._utils" and name == "_rebuild_parameter":
            return _rebuild_parameter
        if module == "torch" and name in _STORAGE_DTYPES:
            return _StorageType(name)
        raise WeightFileError(f"checkpoint pickle references {module}.{name}, which a state-dict checkpoint does not need")

    def persistent_load(self, pid):
        if not (isinstance(pid, tuple) and len(pid) >= 5 and pid[0] == "storage" and isinstance(pid[1], _StorageType)):
            raise WeightFileError("unsupported persistent id in the checkpoint pickle")
        _, stype, key, _location, numel = pid[:5]             # the device the tensor lived on ('cuda:0', 'cpu') is irrelevant here
        return _LazyStorage(self.zf, self.prefix, stype, str(key), numel)


def _materialise(obj):
    if isinstance(obj, _Tensor):
        return obj.numpy()
    if isinstance(obj, OrderedDict):
        return OrderedDict((k, _materialise(v)) for k, v in obj.items())
    if isinstance(obj, dict):
        return {k: _materialise(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_materialise(v) for v in obj)
    return obj


def read_checkpoint(path, key="model_state_dict", everything=False):
    """Tensors of a torch.save zip checkpoint as numpy arrays, in the order they were saved.

    key         the entry of the top-level dict that holds the state dict (main_train.py:408); a file that IS a bare
                state dict is accepted too.
    everything  return the whole decoded object (epoch, loss, optimizer state ...) instead of the state dict.
    """
    try:
        zf = zipfile.ZipFile(path)
    except zipfile.BadZipFile as e:
        raise WeightFileError(f"{path}: not a zip checkpoint (torch.save's legacy pre-1.6 stream format is not supported)") from e
    with zf:
        pkl = [n for n in zf.namelist() if n.endswith("/data.pkl")]
        if len(pkl) != 1:
            raise WeightFileError(f"{path}: expected one */data.pkl entry, found {len(pkl)}")
        prefix = pkl[0][:-len("/data.pkl")]
        bo = f"{prefix}/byteorder"
        if bo in zf.namelist() and zf.read(bo).strip() != b"little":
            raise WeightFileError(f"{path}: big-endian checkpoint")
        try:
            top = _CheckpointUnpickler(io.BytesIO(zf.read(pkl[0])), zf, prefix).load()
        except pickle.UnpicklingError as e:
            raise WeightFileError(f"{path}: {e}") from e
        if everything:
            return _materialise(top)
        if isinstance(top, dict) and key in top:
            sd = top[key]
        elif isinstance(top, dict) and top and all(isinstance(v, _Tensor) for v in top.values()):
            sd = top
        else:
            raise WeightFileError(f"{path}: no '{key}' entry and not a bare state dict")
        if not isinstance(sd, dict) or not all(isinstance(v, _Tensor) for v in sd.values()):
            raise WeightFileError(f"{path}: '{key}' is not a name -> tensor mapping")
        return OrderedDict((str(k), v.numpy()) for k, v in sd.items())


# ------------------------------------------------------------------------------------------------------------
# ONNX (protobuf wire format, onnx.proto3 field numbers)
# ------------------------------------------------------------------------------------------------------------
# ModelProto : ir_version=1 (varint), opset_import=8 (OperatorSetIdProto: domain=1, version=2), graph=7
# GraphProto : node=1, name=2, initializer=5, input=11, output=12
# NodeProto  : input=1 (repeated string), output=2, name=3, op_type=4, attribute=5
# TensorProto: dims=1 (repeated int64, packed or not), data_type=2, float_data=4 (packed float), name=8, raw_data=9,
#              double_data=10;  data_type 1=FLOAT 10=FLOAT16 11=DOUBLE
def _varint(buf, pos):
    val = shift = 0
    while True:
        if pos >= len(buf) or shift > 63:
            raise WeightFileError("truncated or malformed varint")
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _fields(buf):
    """Yield (field_number, wire_type, value) over one message; value is an int (varint / fixed) or a memoryview."""
    pos, n = 0, len(buf)
    while pos < n:
        tag, pos = _varint(buf, pos)
        fno, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            if pos + 8 > n:
                raise WeightFileError("truncated fixed64 field")
            v, pos = struct.unpack_from("<Q", buf, pos)[0], pos + 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            if pos + ln > n:
                raise WeightFileError("length-delimited field runs past the end of its message")
            v, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            if pos + 4 > n:
                raise WeightFileError("truncated fixed32 field")
            v, pos = struct.unpack_from("<I", buf, pos)[0], pos + 4
        else:
            raise WeightFileError(f"unsupported protobuf wire type {wt}")
        yield fno, wt, v


def _tensor_proto(buf):
    dims, dtype, name, raw, fdata, ddata = [], 0, "", None, [], []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            if wt == 2:                                        # packed
                p = 0
                while p < len(v):
                    d, p = _varint(v, p)
                    dims.append(d)
            else:
                dims.append(v)
        elif fno == 2:
            dtype = v
        elif fno == 8:
            name = bytes(v).decode("utf-8")
        elif fno == 9:
            raw = v
        elif fno == 4:
            fdata.append(np.frombuffer(v, "<f4") if wt == 2 else np.array([struct.unpack("<f", struct.pack("<I", v))[0]], "<f4"))
        elif fno == 10:
            ddata.append(np.frombuffer(v, "<f8") if wt == 2 else np.array([struct.unpack("<d", struct.pack("<Q", v))[0]], "<f8"))
    np_dtype = {1: np.dtype("<f4"), 10: np.dtype("<f2"), 11: np.dtype("<f8")}.get(dtype)
    if np_dtype is None:
        return name, None                                      # integer constants (shapes, pads): not weights
    count = int(np.prod(dims)) if dims else 1
    if raw is not None:
        a = np.frombuffer(raw, np_dtype)
    elif dtype == 1 and fdata:
        a = np.concatenate(fdata)
    elif dtype == 11 and ddata:
        a = np.concatenate(ddata)
    else:
        a = np.zeros(0, np_dtype)
    if a.size != count:
        raise WeightFileError(f"initializer '{name}': {a.size} values for dims {dims}")
    return name, a.reshape(dims).copy()


def read_onnx(path):
    """dict(opset, input, output, initializers{name: array}, convs[(op_type, weight_name, has_bias)] in graph order)."""
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    graph, opset = None, None
    try:
        for fno, wt, v in _fields(buf):
            if fno == 7 and wt == 2:
                graph = v
            elif fno == 8 and wt == 2:
                dom, ver = "", None
                for f2, w2, v2 in _fields(v):
                    if f2 == 1:
                        dom = bytes(v2).decode()
                    elif f2 == 2:
                        ver = v2
                if dom in ("", "ai.onnx"):
                    opset = ver
        if graph is None:
            raise WeightFileError(f"{path}: no GraphProto - not an ONNX model")
        inits, convs, inputs, outputs = OrderedDict(), [], [], []
        for fno, wt, v in _fields(graph):
            if fno == 5 and wt == 2:
                name, a = _tensor_proto(v)
                if a is not None:
                    inits[name] = a
            elif fno == 1 and wt == 2:
                ins, op = [], ""
                for f2, w2, v2 in _fields(v):
                    if f2 == 1:
                        ins.append(bytes(v2).decode("utf-8"))
                    elif f2 == 4:
                        op = bytes(v2).decode("utf-8")
                if op in ("Conv", "ConvTranspose"):
                    if len(ins) < 2:
                        raise WeightFileError(f"{path}: {op} node without a weight input")
                    convs.append((op, ins[1], len(ins) > 2))
            elif fno in (11, 12) and wt == 2:
                for f2, w2, v2 in _fields(v):
                    if f2 == 1:
                        (inputs if fno == 11 else outputs).append(bytes(v2).decode("utf-8"))
    except WeightFileError as e:
        raise WeightFileError(f"{path}: {e}") from None
    return {"opset": opset, "inputs": inputs, "outputs": outputs, "initializers": inits, "convs": convs}


# ------------------------------------------------------------------------------------------------------------
# UNetRes ordering
# ------------------------------------------------------------------------------------------------------------
def _to_f32(a, name):
    a = np.asarray(a)
    if a.dtype.kind != "f":
        raise WeightFileError(f"'{name}' is {a.dtype}, not a floating-point weight")
    if not np.all(np.isfinite(a)):
        raise WeightFileError(f"'{name}' holds non-finite values")
    return np.ascontiguousarray(a, dtype=np.float32)


def unetres_blob(tensors, kinds=None):
    """Flat fp32 weights in state-dict order + the architecture, from an ordered sequence of conv weights.

    tensors  ordered mapping name -> array (a state dict), or a list of arrays, in execution (= registration) order
    kinds    optional parallel list of 'Conv' / 'ConvTranspose' (known from an ONNX graph); a state dict carries no op
             type, there the 2x2 layers of the up path are the transposed ones.
    Raises WeightFileError when the list is not a UNetRes (network_unet.py:164-211): any bias, a wrong kernel size, or
    channel counts that do not chain.
    """
    if isinstance(tensors, dict):
        names, arrs = list(tensors.keys()), list(tensors.values())
        if all(n.startswith("module.") for n in names):          # nn.DataParallel wrapper
            names = [n[len("module."):] for n in names]
    else:
        arrs = list(tensors)
        names = [f"#{i}" for i in range(len(arrs))]
    for n, a in zip(names, arrs):
        if np.ndim(a) != 4:
            raise WeightFileError(f"'{n}' has shape {np.shape(a)}: UNetRes is bias-free and holds 4-D conv weights only (network_unet.py:172-205, bias=False)")
    L = len(arrs)
    if L < 22 or (L - 8) % 14:
        raise WeightFileError(f"{L} conv weights cannot be a UNetRes (expected 14*nb + 8)")
    nb = (L - 8) // 14
    shp = [tuple(int(v) for v in np.shape(a)) for a in arrs]
    if shp[0][2:] != (3, 3) or shp[-1][2:] != (3, 3):
        raise WeightFileError("head / tail are not 3x3 convolutions")
    in_nc, out_nc = shp[0][1], shp[-1][0]
    nc = [shp[0][0]]
    for lvl in range(3):
        nc.append(shp[1 + lvl * (2 * nb + 1) + 2 * nb][0])       # strideconv out channels
    from .synth import unetres_weight_shapes                     # the canonical order / shapes (same as the engine's add_layer walk)
    want = unetres_weight_shapes(in_nc, out_nc, tuple(nc), nb)
    for i, ((wname, wshape), s) in enumerate(zip(want, shp)):
        if tuple(wshape) != s:
            raise WeightFileError(f"tensor {i} ('{names[i]}') has shape {s}; UNetRes(in_nc={in_nc}, out_nc={out_nc}, nc={nc}, nb={nb}) "
                                  f"expects {tuple(wshape)} for {wname}")
        transposed = wname.startswith("m_up") and wname.count(".") == 2          # m_up<l>.0.weight
        if kinds is not None and (kinds[i] == "ConvTranspose") != transposed:
            raise WeightFileError(f"tensor {i} ('{names[i]}') is a {kinds[i]}; UNetRes has {'ConvTranspose' if transposed else 'Conv'} at {wname}")
        if isinstance(tensors, dict) and kinds is None and names[i] != wname:
            raise WeightFileError(f"tensor {i} is named '{names[i]}', expected '{wname}'")
    blob = np.concatenate([_to_f32(a, n).ravel() for n, a in zip(names, arrs)])
    return blob, {"in_nc": in_nc, "out_nc": out_nc, "nc": tuple(nc), "nb": nb}


def load_denoiser_weights(path):
    """(flat fp32 weights, dict(in_nc,out_nc,nc,nb)) from a `.pt` checkpoint or an `.onnx` export, by file signature."""
    with open(path, "rb") as f:
        magic = f.read(4)
    if magic[:2] == b"PK":
        return unetres_blob(read_checkpoint(path))
    m = read_onnx(path)
    missing = [w for _, w, _ in m["convs"] if w not in m["initializers"]]
    if missing:
        raise WeightFileError(f"{path}: conv weights {missing[:3]} are not stored as initializers (export_params=False?)")
    if any(b for _, _, b in m["convs"]):
        raise WeightFileError(f"{path}: a convolution carries a bias; UNetRes is bias-free")
    ws = OrderedDict((w, m["initializers"][w]) for _, w, _ in m["convs"])
    if len(ws) != len(m["convs"]):
        raise WeightFileError(f"{path}: a weight initializer is shared between convolutions")
    return unetres_blob(ws, kinds=[k for k, _, _ in m["convs"]])
