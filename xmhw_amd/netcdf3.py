"""Minimal netCDF-3 ("classic", CDF-1 / CDF-2 64-bit-offset / CDF-5) reader in numpy.

The reference leaves I/O to xarray (docs/gettingstarted.rst:30-33); none of xarray / netCDF4 /
h5netcdf is installed in this image, so the ingest path of xmhw_amd (xmhw_amd/ingest.py) ships the
small reader it needs: the header is parsed here, and every variable is returned as a
**memory-mapped, zero-copy view** of the file (big-endian dtype, strided for record variables), with
its attributes (``scale_factor`` / ``add_offset`` / ``_FillValue`` are applied on the DEVICE by the
ingest kernels, not here).  netCDF-4 / HDF5 files are refused with a clear message.

Format: "The NetCDF Classic Format Specification" (header = magic, numrecs, dim_list, gatt_list,
var_list; all integers big-endian; names and values padded to 4 bytes; record variables interleaved
record by record).  A writer for small files (tests, tools) is included; it writes CDF-2.
"""
import mmap
import struct

import numpy as np

_NC_DIMENSION, _NC_VARIABLE, _NC_ATTRIBUTE = 0x0A, 0x0B, 0x0C
_TYPES = {1: ">i1", 2: "S1", 3: ">i2", 4: ">i4", 5: ">f4", 6: ">f8",
          7: ">u1", 8: ">u2", 9: ">u4", 10: ">i8", 11: ">u8"}
_CODES = {"i1": 1, "S1": 2, "i2": 3, "i4": 4, "f4": 5, "f8": 6}


class NetCDF3Error(ValueError):
    pass


class Variable:
    def __init__(self, name, dims, shape, dtype, attrs, data, is_record):
        self.name, self.dims, self.shape, self.dtype = name, tuple(dims), tuple(shape), np.dtype(dtype)
        self.attrs, self.data, self.is_record = attrs, data, is_record

    def __repr__(self):
        return f"<netcdf3.Variable {self.name}{self.dims} {self.dtype} shape={self.shape}>"


class _Reader:
    def __init__(self, buf, version):
        self.buf, self.pos, self.v = buf, 4, version

    def u32(self):
        (x,) = struct.unpack_from(">I", self.buf, self.pos)
        self.pos += 4
        return x

    def u64(self):
        (x,) = struct.unpack_from(">Q", self.buf, self.pos)
        self.pos += 8
        return x

    def count(self):                      # "NON_NEG": 32-bit, 64-bit in CDF-5
        return self.u64() if self.v == 5 else self.u32()

    def name(self):
        n = self.count()
        s = bytes(self.buf[self.pos:self.pos + n]).decode("utf-8")
        self.pos += (n + 3) & ~3
        return s

    def values(self, nc_type, n):
        dt = np.dtype(_TYPES[nc_type])
        nbytes = dt.itemsize * n
        raw = bytes(self.buf[self.pos:self.pos + nbytes])
        self.pos += (nbytes + 3) & ~3
        if nc_type == 2:
            return raw.rstrip(b"\x00").decode("utf-8", "replace")
        a = np.frombuffer(raw, dtype=dt).astype(dt.newbyteorder("="))
        return a[0] if n == 1 else a

    def attrs(self):
        tag, n = self.u32(), self.count()
        if tag == 0 and n == 0:
            return {}
        if tag != _NC_ATTRIBUTE:
            raise NetCDF3Error("corrupt header: attribute list expected")
        out = {}
        for _ in range(n):
            k = self.name()
            t = self.u32()
            out[k] = self.values(t, self.count())
        return out


class File:
    """``File(path)``: ``.dimensions`` {name: length, None for the record dimension's declared
    length}, ``.numrecs``, ``.attrs``, ``.variables`` {name: Variable}.  Keep the object alive while
    the variable views are in use (they map the file)."""

    def __init__(self, path):
        self._f = open(path, "rb")
        head = self._f.read(4)
        if head[:3] != b"CDF":
            self._f.close()
            if head == b"\x89HDF":
                raise NetCDF3Error(f"{path}: netCDF-4 / HDF5 file; only the classic format (CDF-1/2/5) is read here")
            raise NetCDF3Error(f"{path}: not a netCDF classic file")
        self.version = head[3]
        if self.version not in (1, 2, 5):
            self._f.close()
            raise NetCDF3Error(f"{path}: unknown classic format version {self.version}")
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        r = _Reader(self._mm, self.version)
        numrecs = r.count()
        self.dimensions, dim_names, dim_len = {}, [], []
        tag, n = r.u32(), r.count()
        if not (tag == 0 and n == 0):
            if tag != _NC_DIMENSION:
                raise NetCDF3Error("corrupt header: dimension list expected")
            for _ in range(n):
                nm = r.name()
                ln = r.count()
                dim_names.append(nm)
                dim_len.append(ln)
                self.dimensions[nm] = ln if ln else None
        self.attrs = r.attrs()
        tag, n = r.u32(), r.count()
        raw_vars = []
        if not (tag == 0 and n == 0):
            if tag != _NC_VARIABLE:
                raise NetCDF3Error("corrupt header: variable list expected")
            for _ in range(n):
                nm = r.name()
                nd = r.count()
                dimids = [r.count() for _ in range(nd)]
                at = r.attrs()
                t = r.u32()
                vsize = r.count()
                begin = r.u32() if self.version == 1 else r.u64()
                raw_vars.append((nm, dimids, at, t, vsize, begin))
        is_rec = [bool(d) and dim_len[d[0]] == 0 for _, d, *_ in raw_vars]
        rec_vars = [v for v, rr in zip(raw_vars, is_rec) if rr]
        # record size: the sum of the record variables' vsize; a single record variable is NOT padded
        if len(rec_vars) == 1:
            nm, dimids, _, t, _, _ = rec_vars[0]
            recsize = int(np.prod([dim_len[d] for d in dimids[1:]], dtype=np.int64)) * np.dtype(_TYPES[t]).itemsize
        else:
            recsize = sum(v[4] for v in rec_vars)
        if numrecs == 0xFFFFFFFF and rec_vars:        # "streaming": derive from the file size
            numrecs = (len(self._mm) - min(v[5] for v in rec_vars)) // max(recsize, 1)
        self.numrecs = numrecs
        self.variables = {}
        base = np.frombuffer(self._mm, dtype=np.uint8)
        # for readers that want to pread() instead of touching the mapping (xmhw_amd.device._staged_upload)
        self.map_address = base.__array_interface__["data"][0]
        self.map_length = len(self._mm)
        for (nm, dimids, at, t, vsize, begin), rr in zip(raw_vars, is_rec):
            dt = np.dtype(_TYPES[t])
            shape = [dim_len[d] for d in dimids]
            if rr:
                shape[0] = numrecs
                inner = int(np.prod(shape[1:], dtype=np.int64))
                if numrecs == 0:
                    data = np.empty(shape, dtype=dt)
                else:
                    flat = np.lib.stride_tricks.as_strided(
                        base[begin:].view(np.uint8), shape=(numrecs, inner * dt.itemsize), strides=(recsize, 1),
                        writeable=False)
                    if recsize == inner * dt.itemsize:
                        data = base[begin:begin + numrecs * recsize].view(dt).reshape(shape)
                    else:
                        # interleaved with other record variables: rows are recsize bytes apart
                        data = np.ndarray(shape=tuple(shape), dtype=dt, buffer=self._mm, offset=begin,
                                          strides=(recsize,) + tuple(
                                              int(np.prod(shape[k + 1:], dtype=np.int64)) * dt.itemsize
                                              for k in range(1, len(shape))))
                    del flat
            else:
                count = int(np.prod(shape, dtype=np.int64)) if shape else 1
                data = base[begin:begin + count * dt.itemsize].view(dt).reshape(shape)
            self.variables[nm] = Variable(nm, [dim_names[d] for d in dimids], shape, dt, at, data, rr)

    def fileno(self):
        return self._f.fileno()

    def close(self):
        self.variables = {}
        try:
            self._mm.close()
        except (BufferError, ValueError):
            pass                                  # views still alive: the map goes with them
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _pad4(b):
    return b + b"\x00" * (-len(b) % 4)


def _pack_attrs(attrs):
    if not attrs:
        return struct.pack(">II", 0, 0)
    out = struct.pack(">II", _NC_ATTRIBUTE, len(attrs))
    for k, v in attrs.items():
        kb = k.encode()
        out += struct.pack(">I", len(kb)) + _pad4(kb)
        if isinstance(v, str):
            vb = v.encode()
            out += struct.pack(">II", 2, len(vb)) + _pad4(vb)
        else:
            a = np.atleast_1d(np.asarray(v))
            code = _CODES[a.dtype.newbyteorder("=").str[1:]]
            out += struct.pack(">II", code, a.size) + _pad4(a.astype(a.dtype.newbyteorder(">")).tobytes())
    return out


def write_classic(path, dims, variables, attrs=None, record_dim=None):
    """Write a small CDF-2 file.  dims: {name: length}; variables: {name: (dim names, array, attrs)};
    record_dim: name of the unlimited dimension (variables whose first dim it is are interleaved)."""
    names = list(dims)
    numrecs = 0
    head = b"CDF\x02"
    dim_block = struct.pack(">II", _NC_DIMENSION, len(names))
    for n in names:
        nb = n.encode()
        dim_block += struct.pack(">I", len(nb)) + _pad4(nb) + struct.pack(">I", 0 if n == record_dim else dims[n])
    if record_dim is not None:
        numrecs = dims[record_dim]
    gatt = _pack_attrs(attrs)
    entries = []
    for vn, (vd, arr, vat) in variables.items():
        arr = np.asarray(arr)
        code = _CODES[arr.dtype.newbyteorder("=").str[1:]]
        be = arr.astype(arr.dtype.newbyteorder(">"))
        rec = bool(vd) and vd[0] == record_dim
        per = int(np.prod(arr.shape[1:], dtype=np.int64)) * arr.dtype.itemsize if rec else arr.nbytes
        entries.append(dict(name=vn, dims=vd, code=code, data=be, rec=rec, vsize=(per + 3) & ~3, attrs=vat or {}))
    nrec = sum(e["rec"] for e in entries)
    if nrec == 1:
        for e in entries:
            if e["rec"]:
                e["rec_stride"] = int(np.prod(e["data"].shape[1:], dtype=np.int64)) * e["data"].dtype.itemsize
    def var_block(begins):
        out = struct.pack(">II", _NC_VARIABLE, len(entries)) if entries else struct.pack(">II", 0, 0)
        for e, b in zip(entries, begins):
            nb = e["name"].encode()
            out += struct.pack(">I", len(nb)) + _pad4(nb) + struct.pack(">I", len(e["dims"]))
            for d in e["dims"]:
                out += struct.pack(">I", names.index(d))
            out += _pack_attrs(e["attrs"]) + struct.pack(">II", e["code"], e["vsize"]) + struct.pack(">Q", b)
        return out
    hlen = len(head) + 4 + len(dim_block) + len(gatt) + len(var_block([0] * len(entries)))
    pos = hlen
    begins = []
    for e in entries:
        if not e["rec"]:
            begins.append(pos)
            pos += e["vsize"]
        else:
            begins.append(None)
    recstart = pos
    recsize = 0
    for i, e in enumerate(entries):
        if e["rec"]:
            begins[i] = recstart + recsize
            recsize += e.get("rec_stride", e["vsize"])
    with open(path, "wb") as f:
        f.write(head + struct.pack(">I", numrecs) + dim_block + gatt + var_block(begins))
        for e in entries:
            if not e["rec"]:
                f.write(_pad4(e["data"].tobytes()))
        for r in range(numrecs):
            for e in entries:
                if e["rec"]:
                    b = e["data"][r:r + 1].tobytes()
                    f.write(b if "rec_stride" in e else _pad4(b))
    return path
