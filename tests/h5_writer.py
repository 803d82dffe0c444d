"""Test-side writer of MATLAB -v7.3 style files (HDF5 1.8 'earliest' structures: user block + superblock v0, object headers v1,
symbol-table groups, B-tree v1 chunk indexes, deflate + shuffle), used to build fixtures for tests/test_mat73.py -- the image has no
HDF5 library.  Written from the published HDF5 File Format Specification like the reader it exercises; the structures the two share
are pinned independently by a file MATLAB itself wrote (tests/golden/testhdf5_7.4_GLNX86.mat).

    write_mat73(path, {"name": value, ...}, chunk_elems=4096, deflate=3, shuffle=True)
        value: numpy array (real or complex; float32/64, integers, bool), str (char), dict (struct), list (cell, 1 x n)
"""
import struct
import zlib

import numpy as np

O = L = 8
BASE = 512
UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K, INT_K, CHUNK_K = 4, 16, 32


class _W:
    def __init__(self):
        self.buf = bytearray()

    def alloc(self, data, align=8):
        while len(self.buf) % align: self.buf.append(0)
        a = len(self.buf)
        self.buf += data
        return a                                                   # relative to BASE

    def reserve(self, n):
        return self.alloc(bytes(n))

    def put(self, a, data):
        self.buf[a:a + len(data)] = data


def _pad8(b):
    return b + bytes((-len(b)) % 8)


def _msg(t, body, flags=0):
    body = _pad8(body)
    return struct.pack("<HHB3x", t, len(body), flags) + body


def _dt_float(size):
    if size == 8: props = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
    else: props = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
    bits = 0x20 | ((size * 8 - 1) << 8)                           # mantissa normalisation 2 (implied msb), sign bit position
    return struct.pack("<B3sI", 0x11, bits.to_bytes(3, "little"), size) + props


def _dt_int(size, signed):
    return struct.pack("<B3sI", 0x10, (8 if signed else 0).to_bytes(3, "little"), size) + struct.pack("<HH", 0, size * 8)


def _dt_string(n):
    return struct.pack("<B3sI", 0x13, (0).to_bytes(3, "little"), n)


def _dt_ref():
    return struct.pack("<B3sI", 0x17, (0).to_bytes(3, "little"), 8)


def _dt_of(dtype):
    dtype = np.dtype(dtype)
    if dtype.kind == "f": return _dt_float(dtype.itemsize)
    if dtype.kind in "iu": return _dt_int(dtype.itemsize, dtype.kind == "i")
    raise ValueError(dtype)


def _dt_complex(real_dtype):
    real_dtype = np.dtype(real_dtype)
    sz = real_dtype.itemsize
    body = b""
    for k, name in enumerate((b"real", b"imag")):
        body += _pad8(name + b"\0") + struct.pack("<IB3xII4I", k * sz, 0, 0, 0, 0, 0, 0, 0) + _dt_of(real_dtype)
    return struct.pack("<B3sI", 0x16, (2).to_bytes(3, "little"), 2 * sz) + body


def _dataspace(shape):
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", int(d)) for d in shape)


def _attr(name, dt, shape, data):
    nm = name.encode() + b"\0"
    ds = _dataspace(shape)
    return _msg(0x0C, struct.pack("<BxHHH", 1, len(nm), len(dt), len(ds)) + _pad8(nm) + _pad8(dt) + _pad8(ds) + data)


def _attr_str(name, s):
    return _attr(name, _dt_string(len(s)), (), s.encode())


def _attr_i32(name, v):
    return _attr(name, _dt_int(4, True), (), struct.pack("<i", v))


def _object_header(w, msgs):
    body = b"".join(msgs)
    hdr = struct.pack("<BxHII4x", 1, len(msgs), 1, len(body))
    return w.alloc(hdr + body)


def _chunk_btree(w, entries, rank):
    """entries: list of (offsets tuple, nbytes, addr), in row-major order of the offsets.  Returns the root node's address."""
    def node(level, keys, children, last_key):
        data = b"TREE" + struct.pack("<BBH", 1, level, len(children)) + struct.pack("<QQ", UNDEF, UNDEF)
        for (offs, nbytes), c in zip(keys, children):
            data += struct.pack("<II", nbytes, 0) + b"".join(struct.pack("<Q", o) for o in offs) + struct.pack("<Q", 0) + struct.pack("<Q", c)
        offs, nbytes = last_key
        data += struct.pack("<II", nbytes, 0) + b"".join(struct.pack("<Q", o) for o in offs) + struct.pack("<Q", 0)
        return w.alloc(data)
    cap = 2 * CHUNK_K
    level = 0
    items = [((e[0], e[1]), e[2]) for e in entries]
    end_key = (tuple(entries[-1][0]), 0)
    while True:
        groups = [items[i:i + cap] for i in range(0, len(items), cap)]
        nodes = []
        for gi, g in enumerate(groups):
            last = groups[gi + 1][0][0] if gi + 1 < len(groups) else end_key
            nodes.append((g[0][0], node(level, [k for k, _ in g], [c for _, c in g], last)))
        if len(nodes) == 1: return nodes[0][1]
        items = nodes
        level += 1


def _dataset(w, arr, matlab_class, extra_attrs=(), chunk_elems=4096, deflate=3, shuffle=True, force_chunked=False):
    """arr: numpy array in MATLAB shape; stored with reversed dimensions (its transpose, C order)"""
    arr = np.asarray(arr)
    if arr.ndim < 2: arr = arr.reshape((1, -1)) if arr.ndim == 1 else arr.reshape((1, 1))
    h5 = np.ascontiguousarray(arr.T)
    if np.iscomplexobj(h5):
        rd = np.float32 if h5.dtype == np.complex64 else np.float64
        rec = np.empty(h5.shape, dtype=[("real", rd), ("imag", rd)])
        rec["real"], rec["imag"] = h5.real, h5.imag
        dt, raw_arr = _dt_complex(rd), rec
    elif h5.dtype == object:
        raise ValueError("object arrays go through _cell")
    else:
        dt, raw_arr = (_dt_ref() if matlab_class == "cell" else _dt_of(h5.dtype)), h5
    es = raw_arr.dtype.itemsize
    msgs = [_msg(0x01, _dataspace(h5.shape)), _msg(0x03, dt, 1)]
    if h5.size > chunk_elems or force_chunked:
        # chunks: whole trailing dimensions, the leading one cut so that a chunk has about chunk_elems elements
        inner = int(np.prod(h5.shape[1:])) if h5.ndim > 1 else 1
        c0 = max(1, min(h5.shape[0], chunk_elems // max(inner, 1)))
        cdims = (c0,) + tuple(h5.shape[1:])
        if inner > chunk_elems and h5.ndim > 1:                    # also cut the second dimension (ragged edge chunks in two dimensions)
            c1 = max(1, chunk_elems // max(int(np.prod(h5.shape[2:])) if h5.ndim > 2 else 1, 1))
            cdims = (1, min(c1, h5.shape[1])) + tuple(h5.shape[2:])
        entries = []
        grid = [range(0, s, c) for s, c in zip(h5.shape, cdims)]
        for offs in np.ndindex(*[len(g) for g in grid]):
            o = tuple(g[i] for g, i in zip(grid, offs))
            blk = np.zeros(cdims, dtype=raw_arr.dtype)
            sel = tuple(slice(oo, min(oo + c, s)) for oo, c, s in zip(o, cdims, h5.shape))
            blk[tuple(slice(0, s.stop - s.start) for s in sel)] = raw_arr[sel]
            raw = blk.tobytes()
            if shuffle:
                k = len(raw) // es
                raw = np.frombuffer(raw, np.uint8).reshape(k, es).T.tobytes()
            if deflate: raw = zlib.compress(raw, deflate)
            entries.append((o, len(raw), w.alloc(raw)))
        bt = _chunk_btree(w, entries, h5.ndim)
        filt = b""
        nf = 0
        if shuffle: filt += struct.pack("<HHHH", 2, 0, 1, 1) + struct.pack("<I", es) + bytes(4); nf += 1
        if deflate: filt += struct.pack("<HHHH", 1, 0, 1, 1) + struct.pack("<I", deflate) + bytes(4); nf += 1
        if nf: msgs.append(_msg(0x0B, struct.pack("<BB6x", 1, nf) + filt))
        lay = struct.pack("<BBB", 3, 2, h5.ndim + 1) + struct.pack("<Q", bt) + b"".join(struct.pack("<I", c) for c in cdims) + struct.pack("<I", es)
        msgs.append(_msg(0x08, lay))
    else:
        raw = raw_arr.tobytes()
        a = w.alloc(raw) if raw else UNDEF
        msgs.append(_msg(0x08, struct.pack("<BB", 3, 1) + struct.pack("<QQ", a, len(raw))))
    msgs.append(_attr_str("MATLAB_class", matlab_class))
    msgs += list(extra_attrs)
    return _object_header(w, msgs)


def _group(w, members, matlab_class=None):
    """members: {name: object header address}.  Symbol-table group: local heap + B-tree (one level, or two when many) + SNODs."""
    names = sorted(members)
    heap = bytearray(b"\0" * 8)                                     # offset 0: the empty string
    offs = {}
    for n in names:
        offs[n] = len(heap)
        heap += n.encode() + b"\0"
        while len(heap) % 8: heap.append(0)
    heap_data = w.alloc(bytes(heap) + bytes(16))
    heap_hdr = w.alloc(b"HEAP" + struct.pack("<B3x", 0) + struct.pack("<QQQ", len(heap) + 16, len(heap), heap_data))
    w.put(heap_data + len(heap), struct.pack("<QQ", 1, 16))        # free block: next = 1 (none), size
    per = 2 * LEAF_K
    snods = []
    for i in range(0, max(len(names), 1), per):
        part = names[i:i + per]
        data = b"SNOD" + struct.pack("<BxH", 1, len(part))
        for n in part:
            data += struct.pack("<QQII16x", offs[n], members[n], 0, 0)
        data += bytes((per - len(part)) * 40)
        snods.append((offs[part[-1]] if part else 0, w.alloc(data)))
    node = b"TREE" + struct.pack("<BBH", 0, 0, len(snods)) + struct.pack("<QQ", UNDEF, UNDEF) + struct.pack("<Q", 0)
    for key, a in snods:
        node += struct.pack("<Q", a) + struct.pack("<Q", key)
    bt = w.alloc(node)
    msgs = [_msg(0x11, struct.pack("<QQ", bt, heap_hdr))]
    if matlab_class: msgs.append(_attr_str("MATLAB_class", matlab_class))
    return _object_header(w, msgs), bt, heap_hdr


_CLASS = {np.dtype(np.float64): "double", np.dtype(np.float32): "single", np.dtype(np.complex128): "double", np.dtype(np.complex64): "single",
          np.dtype(np.int8): "int8", np.dtype(np.uint8): "uint8", np.dtype(np.int16): "int16", np.dtype(np.uint16): "uint16",
          np.dtype(np.int32): "int32", np.dtype(np.uint32): "uint32", np.dtype(np.int64): "int64", np.dtype(np.uint64): "uint64"}


def _value(w, v, refs, opts):
    if isinstance(v, dict):
        return _group(w, {k: _value(w, x, refs, opts) for k, x in v.items()}, "struct")[0]
    if isinstance(v, str):
        if not v:
            return _dataset(w, np.array([0, 0], np.uint64), "char", [_attr_i32("MATLAB_empty", 1)], **opts)
        return _dataset(w, np.array([[ord(c) for c in v]], np.uint16), "char", [_attr_i32("MATLAB_int_decode", 2)], **opts)
    if isinstance(v, list):                                        # 1 x n cell
        addrs = []
        for x in v:
            a = _value(w, x, refs, opts)
            refs["r%d" % len(refs)] = a
            addrs.append(a)
        return _dataset(w, np.array([addrs], np.uint64), "cell", **opts)
    a = np.asarray(v)
    if a.dtype == bool:
        return _dataset(w, a.astype(np.uint8), "logical", [_attr_i32("MATLAB_int_decode", 1)], **opts)
    if a.size == 0:
        shp = a.shape if a.ndim >= 2 else (0, 0)
        return _dataset(w, np.array(shp, np.uint64), _CLASS[a.dtype], [_attr_i32("MATLAB_empty", 1)], **opts)
    return _dataset(w, a, _CLASS[a.dtype], **opts)


def write_mat73(path, variables, chunk_elems=4096, deflate=3, shuffle=True, force_chunked=False):
    w = _W()
    sb = w.reserve(24 + 4 * 8 + 40)                                # superblock v0 at BASE
    opts = dict(chunk_elems=chunk_elems, deflate=deflate, shuffle=shuffle, force_chunked=force_chunked)
    refs = {}
    members = {k: _value(w, v, refs, opts) for k, v in variables.items()}
    if refs: members["#refs#"] = _group(w, refs)[0]
    root, bt, heap = _group(w, members)
    head = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, O, L, 0, LEAF_K, INT_K, 0)
    head += struct.pack("<QQQQ", BASE, UNDEF, BASE + len(w.buf), UNDEF)
    head += struct.pack("<QQII", 0, root, 1, 0) + struct.pack("<QQ", bt, heap)
    w.put(sb, head)
    text = b"MATLAB 7.3 MAT-file, Platform: GLNXA64, Created on: Sun Oct  4 00:00:00 2026 HDF5 schema 1.00 ."
    user = text.ljust(116, b" ") + bytes(8) + struct.pack("<H", 0x0200) + b"IM"
    with open(path, "wb") as f:
        f.write(user.ljust(BASE, b"\0") + bytes(w.buf))
