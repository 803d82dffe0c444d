"""Reader for MATLAB `-v7.3` MAT-files (HDF5 containers) -- the format MATLAB needs for variables above 2 GB and the one
`save` may default to, so a dictionary or TSMI volume of the reference (`load(dict_dir)`, `load(tsmi_dir)`, `load(MRFmaps_dir)`,
main_recon_tsmis_FFT.m:128,180,207) can arrive in it.  The image has no HDF5 library (no h5py / libhdf5), so this is a
dependency-free reader of the subset of the HDF5 file format MATLAB writes, restated from the published "HDF5 File Format
Specification" (version 1.1 structures of HDF5 1.8: MATLAB creates its files with the library's earliest-format defaults, plus
the version 2 structures a newer library may choose):

  superblock v0/v1 (behind MATLAB's 512-byte user block) and v2/v3; object headers v1 and v2 (with continuation blocks);
  old-style groups (symbol-table message -> B-tree v1 'TREE' / 'SNOD' nodes + local 'HEAP') and new-style compact groups (link messages);
  dataspace messages v1/v2; datatypes: fixed-point, floating-point, string, compound (MATLAB's complex = {real, imag}),
  object reference (cell arrays, '#refs#'); data layout v3 compact / contiguous / chunked (B-tree v1 chunk index) and v1/v2;
  filter pipeline: deflate (zlib), shuffle, fletcher32 (checksum dropped); attribute messages v1-v3 (MATLAB_class, MATLAB_empty,
  MATLAB_int_decode).

What is NOT read (raises NotImplementedError naming the feature): dense (fractal-heap) groups, B-tree v2 / extensible-array /
fixed-array chunk indexes of the 1.10 format, variable-length data other than MATLAB's own attributes, external storage, sparse
matrices, function handles, objects.

MATLAB's conventions on top of HDF5, as far as the path needs them: a numeric array is a dataset whose dimensions are the
MATLAB sizes reversed (column-major data stored row-major), class in the MATLAB_class attribute; complex data is a compound of
'real' and 'imag'; a struct is a group (MATLAB_class 'struct') with one member per field; char is uint16 code units; logical is
uint8 with MATLAB_int_decode; an empty array carries MATLAB_empty and stores its size vector as data; a cell array is a dataset
of object references into '/#refs#'.

    load_mat73(path) -> {name: value}   numeric -> numpy array with MATLAB's shape (complex for compounds), struct -> Struct
                                        (attribute access, like scipy's struct_as_record=False), char -> str, cell -> object ndarray
    is_mat73(path)                      the file has MATLAB's 7.3 header / an HDF5 signature

Pinned by the MATLAB-written file of scipy's own test-suite (tests/golden/testhdf5_7.4_GLNX86.mat: `testdouble = 0:pi/4:2*pi`) and
by files produced by tests/h5_writer.py (structs, complex, chunked + deflate + shuffle), tests/test_mat73.py.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

__all__ = ["load_mat73", "is_mat73", "Struct"]

SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class Struct:
    """MATLAB struct: fields as attributes (and `_fieldnames`, as scipy's mat_struct)."""

    def __init__(self, fields):
        self._fieldnames = list(fields)
        for k, v in fields.items():
            setattr(self, k, v)

    def __repr__(self):
        return "Struct(" + ", ".join(self._fieldnames) + ")"


def is_mat73(path):
    with open(path, "rb") as f:
        head = f.read(2048)
    return head.startswith(b"MATLAB 7.3 MAT-file") or any(head[o:o + 8] == SIG for o in (0, 512, 1024))


class _Dtype:
    """decoded datatype message"""

    def __init__(self, kind, size, np_dtype=None, members=None, strpad=0):
        self.kind, self.size, self.np, self.members, self.strpad = kind, size, np_dtype, members, strpad


class _File:
    def __init__(self, path):
        with open(path, "rb") as f:
            self.b = f.read()
        self.path = path
        self.sb = -1
        for o in (0, 512, 1024, 2048, 4096):
            if self.b[o:o + 8] == SIG:
                self.sb = o
                break
        if self.sb < 0:
            raise ValueError(f"{path}: no HDF5 signature (not a MATLAB v7.3 file)")
        self._superblock()

    # ---- primitives
    def u(self, off, n):
        return int.from_bytes(self.b[off:off + n], "little")

    def addr(self, off):
        v = self.u(off, self.O)
        return v if v == (1 << (8 * self.O)) - 1 else v + self.base      # undefined address stays all-ones

    def undefined(self, a):
        return a == (1 << (8 * self.O)) - 1

    def _superblock(self):
        s, b = self.sb, self.b
        ver = b[s + 8]
        if ver in (0, 1):
            self.O, self.L = b[s + 13], b[s + 14]
            p = s + 24 + (4 if ver == 1 else 0)
            self.base = self.u(p, self.O)
            # MATLAB's files: base address 0 with a 512-byte user block -- addresses then count from the superblock's position
            if self.base == 0 and s != 0:
                self.base = s if self._relative_to_superblock(p) else 0
            p += 4 * self.O                                        # base, free-space, end-of-file, driver info
            # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
            self.root = self.addr(p + self.O)
        elif ver in (2, 3):
            self.O, self.L = b[s + 9], b[s + 10]
            self.base = self.u(s + 12, self.O)
            self.root = self.u(s + 12 + 3 * self.O, self.O) + self.base
        else:
            raise NotImplementedError(f"{self.path}: HDF5 superblock version {ver}")

    def _relative_to_superblock(self, p):
        """With a user block the library writes base address = user block size; some writers leave 0 and mean the same.  Decide by
        looking where the root object header is: the candidate whose first byte looks like an object header wins."""
        rel = self.u(p + 4 * self.O + self.O, self.O)
        for base in (0, self.sb):
            a = rel + base
            if a + 16 <= len(self.b) and (self.b[a] == 1 and self.b[a + 1] == 0 or self.b[a:a + 4] == b"OHDR"):
                return base == self.sb
        return False

    # ---- object headers -> list of (type, flags, bytes)
    def messages(self, a):
        b = self.b
        out = []
        if b[a:a + 4] == b"OHDR":                                  # version 2
            flags = b[a + 5]
            p = a + 6
            if flags & 0x20: p += 16                               # times
            if flags & 0x10: p += 4                                # attribute storage phase change
            szlen = 1 << (flags & 3)
            chunk0 = self.u(p, szlen); p += szlen
            blocks = [(p, p + chunk0)]
            track = bool(flags & 0x04)
            while blocks:
                q, end = blocks.pop(0)
                while q + 4 <= end - 0:                            # (a gap smaller than a message header may precede the checksum)
                    t = b[q]; sz = self.u(q + 1, 2); fl = b[q + 3]; q += 4
                    if track: q += 2
                    if q + sz > end: break
                    body = b[q:q + sz]
                    q += sz
                    if t == 0x10:
                        ca, cl = self.addr_in(body, 0), int.from_bytes(body[self.O:self.O + self.L], "little")
                        if b[ca:ca + 4] != b"OCHK": raise ValueError("object header continuation without OCHK signature")
                        blocks.append((ca + 4, ca + cl - 4))       # (checksum at the end)
                    elif t != 0:
                        out.append((t, fl, body))
            return out
        if b[a] != 1:
            raise ValueError(f"{self.path}: no object header at {a}")
        nmsg = self.u(a + 2, 2)
        size = self.u(a + 8, 4)
        blocks = [(a + 16, a + 16 + size)]
        while blocks and len(out) < 4096:
            q, end = blocks.pop(0)
            while q + 8 <= end and nmsg > 0:
                t = self.u(q, 2); sz = self.u(q + 2, 2); fl = b[q + 4]; q += 8
                body = b[q:q + sz]
                q += sz
                nmsg -= 1
                if t == 0x10:
                    blocks.append((self.addr_in(body, 0), 0))
                    ca = self.addr_in(body, 0); cl = int.from_bytes(body[self.O:self.O + self.L], "little")
                    blocks[-1] = (ca, ca + cl)
                elif t != 0:
                    out.append((t, fl, body))
        return out

    def addr_in(self, body, off):
        v = int.from_bytes(body[off:off + self.O], "little")
        return v if v == (1 << (8 * self.O)) - 1 else v + self.base

    # ---- groups
    def links(self, a):
        """{name: object header address} of the group whose header is at a"""
        out = {}
        for t, fl, body in self.messages(a):
            if t == 0x11:                                          # symbol table: B-tree address, local heap address
                bt, hp = self.addr_in(body, 0), self.addr_in(body, self.O)
                heap = self._heap(hp)
                self._group_btree(bt, heap, out)
            elif t == 0x06:                                        # link message
                name, target = self._link(body)
                if target is not None: out[name] = target
            elif t == 0x02:                                        # link info: dense storage?
                p = 2 + (8 if body[1] & 1 else 0)
                fh = int.from_bytes(body[p:p + self.O], "little")
                if fh != (1 << (8 * self.O)) - 1:
                    raise NotImplementedError(f"{self.path}: group with dense (fractal heap) link storage")
        return out

    def _heap(self, a):
        if self.b[a:a + 4] != b"HEAP": raise ValueError("local heap signature missing")
        return self.addr(a + 8 + 2 * self.L)                        # address of the data segment

    def _group_btree(self, a, heap, out):
        b = self.b
        if b[a:a + 4] != b"TREE": raise ValueError("group B-tree signature missing")
        level, n = b[a + 5], self.u(a + 6, 2)
        p = a + 8 + 2 * self.O
        for i in range(n):
            child = self.addr(p + self.L)                          # key i (L bytes), child i
            p += self.L + self.O
            if level > 0: self._group_btree(child, heap, out)
            else: self._snod(child, heap, out)

    def _snod(self, a, heap, out):
        b = self.b
        if b[a:a + 4] != b"SNOD": raise ValueError("symbol table node signature missing")
        n = self.u(a + 6, 2)
        p = a + 8
        esz = 2 * self.O + 4 + 4 + 16
        for i in range(n):
            noff = self.u(p, self.O)
            oh = self.addr(p + self.O)
            q = heap + noff
            name = b[q:b.index(b"\0", q)].decode("utf-8")
            out[name] = oh
            p += esz

    def _link(self, body):
        flags = body[1]
        p = 2
        ltype = 0
        if flags & 0x08: ltype = body[p]; p += 1
        if flags & 0x04: p += 8
        if flags & 0x10: p += 1
        nl = 1 << (flags & 3)
        n = int.from_bytes(body[p:p + nl], "little"); p += nl
        name = body[p:p + n].decode("utf-8"); p += n
        if ltype != 0: return name, None                           # soft / external links: not followed
        return name, self.addr_in(body, p)

    # ---- datatypes
    def dtype(self, body, off=0):
        """-> (_Dtype, bytes consumed)"""
        cv = body[off]
        cls, ver = cv & 15, cv >> 4
        bits = int.from_bytes(body[off + 1:off + 4], "little")
        size = int.from_bytes(body[off + 4:off + 8], "little")
        p = off + 8
        if cls == 0:                                               # fixed point
            if bits & 1: raise NotImplementedError("big-endian integers")
            signed = bool(bits & 8)
            return _Dtype("int", size, np.dtype(("<i" if signed else "<u") + str(size))), p + 4 - off
        if cls == 1:                                               # floating point
            if bits & 1: raise NotImplementedError("big-endian floats")
            if size not in (4, 8): raise NotImplementedError(f"{size}-byte floats")
            return _Dtype("float", size, np.dtype("<f" + str(size))), p + 12 - off
        if cls == 3:                                               # fixed-length string
            return _Dtype("string", size, np.dtype("S" + str(size)), strpad=bits & 15), p - off
        if cls == 6:                                               # compound
            nmem = bits & 0xFFFF
            members = []
            for _ in range(nmem):
                e = body.index(b"\0", p)
                name = body[p:e].decode("ascii")
                if ver < 3: p += ((e - p + 1) + 7) // 8 * 8
                else: p = e + 1
                if ver < 3:
                    moff = int.from_bytes(body[p:p + 4], "little"); p += 4
                    if ver == 1: p += 1 + 3 + 4 + 4 + 16           # dimensionality, reserved, permutation, reserved, 4 dim sizes
                else:
                    nb = 1 if size < 256 else 2 if size < 65536 else 3 if size < (1 << 24) else 4
                    moff = int.from_bytes(body[p:p + nb], "little"); p += nb
                mt, used = self.dtype(body, p)
                p += used
                members.append((name, moff, mt))
            npd = np.dtype({"names": [m[0] for m in members], "formats": [m[2].np for m in members],
                            "offsets": [m[1] for m in members], "itemsize": size})
            return _Dtype("compound", size, npd, members=members), p - off
        if cls == 7:                                               # reference (object reference = an address)
            return _Dtype("ref", size, np.dtype("<u8")), p - off
        if cls == 9:
            return _Dtype("vlen", size, None), len(body) - off     # (only met in attributes that are skipped)
        raise NotImplementedError(f"HDF5 datatype class {cls}")

    def dataspace(self, body):
        ver, rank, flags = body[0], body[1], body[2]
        if ver == 1: p = 8
        elif ver == 2:
            p = 4
            if body[3] == 2: return None                           # null dataspace
        else: raise NotImplementedError(f"dataspace message version {ver}")
        return tuple(int.from_bytes(body[p + i * self.L:p + (i + 1) * self.L], "little") for i in range(rank))

    # ---- attributes of an object: {name: value} (strings decoded, small numeric arrays as numpy)
    def attributes(self, msgs):
        out = {}
        for t, fl, body in msgs:
            if t != 0x0C: continue
            ver = body[0]
            nsz, tsz, ssz = (int.from_bytes(body[2 + 2 * i:4 + 2 * i], "little") for i in range(3))
            p = 8
            if ver == 3: p = 9
            pad = (lambda n: (n + 7) // 8 * 8) if ver == 1 else (lambda n: n)
            name = body[p:p + nsz].split(b"\0")[0].decode("utf-8"); p += pad(nsz)
            try:
                dt, _ = self.dtype(body[p:p + tsz])
            except NotImplementedError:
                continue
            p += pad(tsz)
            shape = self.dataspace(body[p:p + ssz]); p += pad(ssz)
            if dt.kind == "vlen" or shape is None:
                continue
            n = int(np.prod(shape)) if shape else 1
            raw = body[p:p + n * dt.size]
            if dt.kind == "string":
                out[name] = raw.split(b"\0")[0].decode("utf-8")
            else:
                v = np.frombuffer(raw, dtype=dt.np, count=n)
                out[name] = v[0] if n == 1 else v.copy()
        return out

    # ---- dataset raw data -> numpy array in HDF5 (row-major) shape
    def dataset(self, msgs):
        dt = shape = layout = None
        filters = []
        for t, fl, body in msgs:
            if t == 0x01: shape = self.dataspace(body)
            elif t == 0x03: dt, _ = self.dtype(body)
            elif t == 0x08: layout = body
            elif t == 0x0B: filters = self._filters(body)
        if dt is None or layout is None:
            raise ValueError("object is not a dataset")
        if shape is None: shape = (0,)
        n = int(np.prod(shape)) if shape else 1
        ver = layout[0]
        if ver == 3:
            cls = layout[1]
            if cls == 0:
                sz = int.from_bytes(layout[2:4], "little")
                raw = layout[4:4 + sz]
            elif cls == 1:
                a = self.addr_in(layout, 2)
                sz = int.from_bytes(layout[2 + self.O:2 + self.O + self.L], "little")
                raw = b"" if self.undefined(a) else self.b[a:a + sz]
            elif cls == 2:
                nd = layout[2]
                bt = self.addr_in(layout, 3)
                cdims = [int.from_bytes(layout[3 + self.O + 4 * i:7 + self.O + 4 * i], "little") for i in range(nd)]
                return self._chunked(bt, cdims[:-1], shape, dt, filters)
            else:
                raise NotImplementedError(f"data layout class {cls}")
        elif ver in (1, 2):
            nd, cls = layout[1], layout[2]
            p = 8
            a = None
            if cls != 0: a = self.addr_in(layout, p); p += self.O
            dims = [int.from_bytes(layout[p + 4 * i:p + 4 * i + 4], "little") for i in range(nd)]
            p += 4 * nd
            if cls == 2:
                return self._chunked(a, dims[:-1], shape, dt, filters)   # (nd counts the element-size dimension; dims[-1] = element size)
            if cls == 0:
                sz = int.from_bytes(layout[p:p + 4], "little"); raw = layout[p + 4:p + 4 + sz]
            else:
                raw = b"" if self.undefined(a) else self.b[a:a + n * dt.size]
        elif ver == 4:
            raise NotImplementedError(f"{self.path}: version 4 data layout (HDF5 1.10 chunk indexes)")
        else:
            raise NotImplementedError(f"data layout message version {ver}")
        if len(raw) < n * dt.size:                                 # never written: fill value zero
            raw = raw + b"\0" * (n * dt.size - len(raw))
        return np.frombuffer(raw, dtype=dt.np, count=n).reshape(shape), dt

    def _filters(self, body):
        ver, nf = body[0], body[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(nf):
            fid = int.from_bytes(body[p:p + 2], "little"); p += 2
            nlen = 0
            if ver == 1 or fid >= 256: nlen = int.from_bytes(body[p:p + 2], "little"); p += 2
            p += 2                                                  # flags
            ncd = int.from_bytes(body[p:p + 2], "little"); p += 2
            if nlen: p += (nlen + 7) // 8 * 8 if ver == 1 else nlen
            cd = [int.from_bytes(body[p + 4 * i:p + 4 * i + 4], "little") for i in range(ncd)]
            p += 4 * ncd
            if ver == 1 and ncd % 2: p += 4
            out.append((fid, cd))
        return out

    def _chunked(self, bt, cdims, shape, dt, filters):
        rank = len(shape)
        out = np.zeros(shape, dtype=dt.np)
        if self.undefined(bt) or out.size == 0:
            return out, dt
        csize = int(np.prod(cdims)) * dt.size
        for offs, fmask, a, nbytes in self._chunk_btree(bt, rank):
            raw = self.b[a:a + nbytes]
            for i, (fid, cd) in reversed(list(enumerate(filters))):     # undo the pipeline back to front
                if fmask & (1 << i): continue
                if fid == 1: raw = zlib.decompress(raw)
                elif fid == 2:
                    es = cd[0] if cd else dt.size
                    k = len(raw) // es
                    raw = np.frombuffer(raw[:k * es], np.uint8).reshape(es, k).T.tobytes() + raw[k * es:]
                elif fid == 3: raw = raw[:-4]
                else: raise NotImplementedError(f"HDF5 filter {fid}")
            if len(raw) < csize: raise ValueError("short chunk")
            blk = np.frombuffer(raw, dtype=dt.np, count=int(np.prod(cdims))).reshape(cdims)
            sel_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shape))
            sel_in = tuple(slice(0, so.stop - so.start) for so in sel_out)
            out[sel_out] = blk[sel_in]
        return out, dt

    def _chunk_btree(self, a, rank):
        b = self.b
        if b[a:a + 4] != b"TREE": raise ValueError("chunk B-tree signature missing")
        if b[a + 4] != 1: raise ValueError("not a raw-data B-tree")
        level, n = b[a + 5], self.u(a + 6, 2)
        p = a + 8 + 2 * self.O
        ksz = 8 + 8 * (rank + 1)
        for i in range(n):
            nbytes, fmask = self.u(p, 4), self.u(p + 4, 4)
            offs = [self.u(p + 8 + 8 * j, 8) for j in range(rank)]
            child = self.addr(p + ksz)
            p += ksz + self.O
            if level > 0: yield from self._chunk_btree(child, rank)
            else: yield offs, fmask, child, nbytes


_INT_CLASSES = {"int8": np.int8, "uint8": np.uint8, "int16": np.int16, "uint16": np.uint16, "int32": np.int32, "uint32": np.uint32,
                "int64": np.int64, "uint64": np.uint64, "double": np.float64, "single": np.float32}


def _convert(f, a, depth=0):
    """object at header address a -> MATLAB value"""
    if depth > 64: raise ValueError("nesting too deep")
    msgs = f.messages(a)
    types = {t for t, _, _ in msgs}
    att = f.attributes(msgs)
    cls = att.get("MATLAB_class", "")
    if 0x08 not in types:                                          # a group: struct (or the file's root / '#refs#')
        if cls not in ("struct", ""):
            raise NotImplementedError(f"MATLAB class '{cls}' stored as a group (objects, function handles, sparse matrices are not read)")
        return Struct({k: _convert(f, v, depth + 1) for k, v in f.links(a).items()})
    arr, dt = f.dataset(msgs)
    if "MATLAB_empty" in att and int(att["MATLAB_empty"]):          # data = the size vector
        shape = tuple(int(v) for v in np.asarray(arr).ravel())
        base = _INT_CLASSES.get(cls, np.float64)
        if cls == "char": return ""
        if cls == "cell": return np.empty(shape, dtype=object)
        return np.zeros(shape, dtype=base)
    if dt.kind == "compound":
        names = [m[0] for m in dt.members]
        if set(names) == {"real", "imag"}:
            arr = arr["real"] + 1j * arr["imag"] if arr["real"].dtype == np.float64 else (arr["real"] + 1j * arr["imag"]).astype(np.complex64 if arr["real"].dtype == np.float32 else np.complex128)
        else:
            raise NotImplementedError(f"compound datatype with members {names}")
    arr = np.ascontiguousarray(arr.T)                              # HDF5 dims are MATLAB's reversed: the transpose has MATLAB's shape and values
    if dt.kind == "ref":
        if cls not in ("cell", ""):
            raise NotImplementedError(f"references in a '{cls}' array")
        out = np.empty(arr.shape, dtype=object)
        for idx, r in np.ndenumerate(arr):
            out[idx] = _convert(f, int(r) + f.base, depth + 1)
        return out
    if cls == "char":
        return "".join(chr(int(c)) for c in arr.ravel(order="F")) if arr.ndim <= 2 and min(arr.shape or (1,)) <= 1 else \
            np.array(["".join(chr(int(c)) for c in row) for row in arr.reshape(arr.shape[0], -1)])
    if cls == "logical":
        return arr.astype(bool)
    if cls in _INT_CLASSES and dt.kind in ("int", "float"):
        return arr.astype(_INT_CLASSES[cls], copy=False)
    return arr


def load_mat73(path, squeeze_me=True):
    """{variable name: value} of a MATLAB -v7.3 file.  squeeze_me: drop singleton dimensions (as harness.load_mat asks of scipy)."""
    f = _File(path)
    out = {}
    for name, a in f.links(f.root).items():
        if name.startswith("#"): continue                          # '#refs#', '#subsystem#'
        out[name] = _convert(f, a)
    if squeeze_me:
        def sq(v):
            if isinstance(v, np.ndarray) and v.dtype != object:
                v = np.squeeze(v)
                return v[()] if v.ndim == 0 else v
            if isinstance(v, Struct):
                for k in v._fieldnames: setattr(v, k, sq(getattr(v, k)))
            return v
        out = {k: sq(v) for k, v in out.items()}
    return out
