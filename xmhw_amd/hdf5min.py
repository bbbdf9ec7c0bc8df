"""A minimal reader of the HDF5 subset netCDF-4 files use -- enough to hand a (time, y, x) variable and its
coordinate variables to the device ingest without h5py / netCDF4 (neither is in the image).

The reference's users open netCDF-4 through xarray (docs/gettingstarted.rst:30-33) and its own fixtures
(test/testdata/*.nc) are netCDF-4, i.e. HDF5 written by netCDF-C: superblock 0..3, version 1 and 2 object
headers (with continuation blocks), old-style groups (symbol table: B-tree v1 + local heap) and new-style ones
(link messages; dense links and dense attributes in fractal heaps), contiguous / compact / chunked (layout 3:
B-tree v1) datasets, the deflate, shuffle and fletcher32 filters, fixed-point / float / fixed-length string
datatypes (variable-length strings through the global heap).  Everything else -- layout version 4 (files
written with libver='latest'), other filters (szip, zstd, blosc ...), compound / enum / array element types --
is refused by name.  Format: "HDF5 File Format Specification Version 3.0" (restated from the published
specification; no HDF5 library source was consulted).

    f = File(path); v = f["sst"]; v.shape, v.dtype, v.attrs, v.read() / v.contiguous_offset()
"""
import mmap
import struct
import zlib

import numpy as np

from .exception import XmhwException

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class _Buf:
    def __init__(self, mm, osz, lsz):
        self.mm, self.osz, self.lsz = mm, osz, lsz

    def u(self, pos, n):
        return int.from_bytes(self.mm[pos:pos + n], "little")

    def off(self, pos):
        return self.u(pos, self.osz)

    def ln(self, pos):
        return self.u(pos, self.lsz)


def _pad8(n):
    return (n + 7) & ~7


class Datatype:
    def __init__(self, cls, size, np_dtype=None, vlen_string=False, base=None):
        self.cls, self.size, self.np_dtype, self.vlen_string, self.base = cls, size, np_dtype, vlen_string, base


def _parse_datatype(b, pos):
    """-> (Datatype, bytes consumed)"""
    cv = b.mm[pos]
    cls, ver = cv & 0x0F, cv >> 4
    bits = b.u(pos + 1, 3)
    size = b.u(pos + 4, 4)
    p = pos + 8
    if cls == 0:                                        # fixed point
        order = ">" if bits & 1 else "<"
        kind = "i" if bits & 8 else "u"
        return Datatype(0, size, np.dtype(f"{order}{kind}{size}")), 8 + 4
    if cls == 1:                                        # floating point
        order = ">" if bits & 1 else "<"
        return Datatype(1, size, np.dtype(f"{order}f{size}")), 8 + 12
    if cls == 3:                                        # fixed-length string
        return Datatype(3, size, np.dtype(f"S{size}")), 8
    if cls == 9:                                        # variable length: properties = the base type
        base, used = _parse_datatype(b, p)
        return Datatype(9, size, None, vlen_string=(bits & 0x0F) == 1, base=base), 8 + used
    if cls == 7:                                        # reference
        return Datatype(7, size, None), 8
    if cls == 6:                                        # compound: members are skipped (only its size matters here)
        return Datatype(6, size, None), None
    if cls == 8:                                        # enum
        base, used = _parse_datatype(b, p)
        return Datatype(8, size, None, base=base), None
    return Datatype(cls, size, None), None


def _parse_dataspace(b, pos):
    """-> (shape or None for a null dataspace, bytes consumed)"""
    ver = b.mm[pos]
    rank = b.mm[pos + 1]
    flags = b.mm[pos + 2]
    if ver == 1:
        p = pos + 8
    elif ver == 2:
        if b.mm[pos + 3] == 2:                          # null dataspace
            return None, 4
        p = pos + 4
    else:
        raise XmhwException(f"HDF5: dataspace message version {ver}")
    shape = tuple(b.ln(p + i * b.lsz) for i in range(rank))
    p += rank * b.lsz
    if flags & 1:
        p += rank * b.lsz
    return shape, p - pos


class Dataset:
    def __init__(self, f, name, addr):
        self.file, self.name, self.addr = f, name, addr
        self.shape, self.dt, self.layout, self.filters, self.attrs = None, None, None, [], {}
        self.fill = None
        f._read_object(addr, self)

    is_group = False

    @property
    def dtype(self):
        if self.dt is None or self.dt.np_dtype is None:
            raise XmhwException(f"{self.file.path}: {self.name}: element type class {getattr(self.dt, 'cls', None)} is not supported")
        return self.dt.np_dtype

    def contiguous_offset(self):
        """file offset of the raw array when it is stored contiguously and unfiltered, else None"""
        if self.layout and self.layout[0] == "contiguous" and not self.filters and self.layout[1] != _UNDEF:
            return self.layout[1]
        return None

    def read(self):
        """the whole array in its stored dtype / byte order"""
        f, b = self.file, self.file.b
        dt = self.dtype
        shape = self.shape or ()
        out = np.empty(shape, dtype=dt)
        kind = self.layout[0]
        if kind == "compact":
            return np.frombuffer(self.layout[1], dtype=dt, count=out.size).reshape(shape).copy()
        if kind == "contiguous":
            addr, size = self.layout[1], self.layout[2]
            if addr == _UNDEF:                          # never written: all fill value
                out[...] = self.fill if self.fill is not None else 0
                return out
            return np.frombuffer(b.mm, dtype=dt, count=out.size, offset=addr).reshape(shape).copy()
        if kind not in ("chunked", "single", "implicit"):
            raise XmhwException(f"{f.path}: {self.name}: data layout {kind!r} is not supported")
        btree, cshape = self.layout[1], self.layout[2]
        out[...] = self.fill if self.fill is not None else 0
        if kind == "single":
            n = self.layout[3] if self.layout[3] is not None else int(np.prod(cshape)) * dt.itemsize
            chunks = [((0,) * len(cshape), self.layout[4], n, btree)]
        elif kind == "implicit":
            n = int(np.prod(cshape)) * dt.itemsize
            counts = [-(-s // c) for s, c in zip(shape, cshape)]
            chunks = [(tuple(i * c for i, c in zip(idx, cshape)), 0xFFFFFFFF, n, btree + k * n)
                      for k, idx in enumerate(np.ndindex(*counts))]
        else:
            chunks = f._chunks(btree, len(cshape)) if btree != _UNDEF else []
        if btree != _UNDEF:
            for offs, fmask, size, addr in chunks:
                raw = bytes(b.mm[addr:addr + size])
                for i in reversed(range(len(self.filters))):
                    if fmask & (1 << i):
                        continue
                    fid, cd = self.filters[i]
                    if fid == 1:
                        raw = zlib.decompress(raw)
                    elif fid == 2:                      # shuffle: bytes of all elements plane by plane
                        es = cd[0] if cd else dt.itemsize
                        n = len(raw) // es
                        raw = np.frombuffer(raw[:n * es], dtype=np.uint8).reshape(es, n).T.tobytes() + raw[n * es:]
                    elif fid == 3:                      # fletcher32: a 4-byte checksum behind the data
                        raw = raw[:-4]
                    else:
                        raise XmhwException(f"{f.path}: {self.name}: HDF5 filter id {fid} is not available here "
                                            f"(deflate, shuffle and fletcher32 are)")
                chunk = np.frombuffer(raw, dtype=dt, count=int(np.prod(cshape))).reshape(cshape)
                sel = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cshape, shape))
                if any(s.start >= s.stop for s in sel):
                    continue
                out[sel] = chunk[tuple(slice(0, s.stop - s.start) for s in sel)]
        return out


class Group:
    is_group = True

    def __init__(self, f, name, addr):
        self.file, self.name, self.addr = f, name, addr
        self.links, self.attrs = {}, {}
        self.shape = self.dt = self.layout = None
        self.filters, self.fill = [], None
        f._read_object(addr, self)


class File:
    def __init__(self, path):
        self.path = path
        self._fh = open(path, "rb")
        try:
            self._mm = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        except ValueError:
            self._fh.close()
            raise XmhwException(f"{path}: empty file")
        mm = self._mm
        if mm[:8] != _SIG:
            raise XmhwException(f"{path}: not an HDF5 / netCDF-4 file")
        ver = mm[8]
        if ver in (0, 1):
            osz, lsz = mm[13], mm[14]
            self.b = _Buf(mm, osz, lsz)
            p = 24 + (4 if ver == 1 else 0)
            self.base = self.b.off(p)
            ste = p + 4 * osz                           # root group symbol table entry
            root = self.b.off(ste + osz)
        elif ver in (2, 3):
            osz, lsz = mm[9], mm[10]
            self.b = _Buf(mm, osz, lsz)
            self.base = self.b.off(12)
            root = self.b.off(12 + 3 * osz)
        else:
            raise XmhwException(f"{path}: HDF5 superblock version {ver}")
        if self.base != 0:
            raise XmhwException(f"{path}: HDF5 base address {self.base} (a user block) is not supported")
        self.root = Group(self, "/", root)
        self._cache = {}
        # the whole mapping as a numpy view: zero-copy windows for contiguous datasets (and its address, for
        # uploads that pread() instead of faulting the mapping in -- xmhw_amd/device.py)
        self._base = np.frombuffer(self._mm, dtype=np.uint8)
        self.map_address = self._base.__array_interface__["data"][0]
        self.map_length = len(self._mm)

    def window(self, offset, dtype, shape):
        """a zero-copy array over the mapped file"""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return self._base[offset:offset + n].view(dtype).reshape(shape)

    def fileno(self):
        return self._fh.fileno()

    def close(self):
        self._base = None
        try:
            self._mm.close()
        except BufferError:          # views handed out are still alive: the mapping goes with them
            pass
        finally:
            self._fh.close()

    def keys(self):
        return list(self.root.links)

    def __contains__(self, name):
        return name in self.root.links

    def __getitem__(self, name):
        if name not in self._cache:
            if name not in self.root.links:
                raise KeyError(name)
            addr = self.root.links[name]
            probe = Group(self, name, addr)
            # a dataset has a layout message, a group does not
            self._cache[name] = probe if probe.layout is None else Dataset(self, name, addr)
        return self._cache[name]

    # ---- object headers ---------------------------------------------------------------------
    def _read_object(self, addr, obj):
        b = self.b
        mm = b.mm
        msgs = []
        if mm[addr:addr + 4] == b"OHDR":
            flags = mm[addr + 5]
            p = addr + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            n = 1 << (flags & 3)
            size0 = b.u(p, n)
            p += n
            blocks = [(p, p + size0)]
            extra = 2 if flags & 0x04 else 0
            while blocks:
                q, end = blocks.pop(0)
                while q + 4 + extra <= end:
                    mtype, msize = mm[q], b.u(q + 1, 2)
                    data = q + 4 + extra
                    if data + msize > end:
                        break
                    if mtype == 0x10:                   # continuation: an OCHK block (signature ... checksum)
                        co, cl = b.off(data), b.ln(data + b.osz)
                        if mm[co:co + 4] != b"OCHK":
                            raise XmhwException(f"{self.path}: bad object header continuation at {co}")
                        blocks.append((co + 4, co + cl - 4))
                    elif mtype != 0:
                        msgs.append((mtype, data, msize))
                    q = data + msize
        elif mm[addr] == 1:
            nmsg = b.u(addr + 2, 2)
            hsize = b.u(addr + 8, 4)
            blocks = [(addr + 16, addr + 16 + hsize)]
            while blocks and len(msgs) < nmsg + 64:
                q, end = blocks.pop(0)
                while q + 8 <= end:
                    mtype, msize = b.u(q, 2), b.u(q + 2, 2)
                    data = q + 8
                    if mtype == 0x10:
                        co, cl = b.off(data), b.ln(data + b.osz)
                        blocks.append((co, co + cl))
                    elif mtype != 0:
                        msgs.append((mtype, data, msize))
                    q = data + _pad8(msize)
        else:
            raise XmhwException(f"{self.path}: unknown object header at {addr}")
        for mtype, data, msize in msgs:
            self._message(obj, mtype, data, msize)

    def _message(self, obj, mtype, p, size):
        b = self.b
        mm = b.mm
        if mtype == 0x01:
            obj.shape, _ = _parse_dataspace(b, p)
        elif mtype == 0x03:
            obj.dt, _ = _parse_datatype(b, p)
        elif mtype == 0x05:                             # fill value
            ver = mm[p]
            if ver in (1, 2):
                defined = mm[p + 3]
                if ver == 1 or defined:
                    n = b.u(p + 4, 4)
                    obj._fill_raw = bytes(mm[p + 8:p + 8 + n]) if n else None
                else:
                    obj._fill_raw = None
            elif ver == 3:
                fl = mm[p + 1]
                obj._fill_raw = None
                if fl & 0x20:
                    n = b.u(p + 2, 4)
                    obj._fill_raw = bytes(mm[p + 6:p + 6 + n])
            raw = getattr(obj, "_fill_raw", None)
            if raw and obj.dt is not None and obj.dt.np_dtype is not None and len(raw) == obj.dt.size:
                obj.fill = np.frombuffer(raw, dtype=obj.dt.np_dtype)[0]
        elif mtype == 0x08:                             # data layout
            ver = mm[p]
            if ver not in (3, 4):
                raise XmhwException(f"{self.path}: {obj.name}: HDF5 data layout message version {ver}")
            cls = mm[p + 1]
            if cls == 0:
                n = b.u(p + 2, 2)
                obj.layout = ("compact", bytes(mm[p + 4:p + 4 + n]))
            elif cls == 1:
                obj.layout = ("contiguous", b.off(p + 2), b.ln(p + 2 + b.osz))
            elif cls == 2 and ver == 3:
                nd = mm[p + 2]
                bt = b.off(p + 3)
                dims = tuple(b.u(p + 3 + b.osz + 4 * i, 4) for i in range(nd))
                obj.layout = ("chunked", bt, dims[:-1])
            elif cls == 2:
                # version 4 (libver='latest'): the chunk index is one of five structures; the two trivial ones are read
                fl, nd, enc = mm[p + 2], mm[p + 3], mm[p + 4]
                dims = tuple(b.u(p + 5 + enc * i, enc) for i in range(nd))
                q = p + 5 + enc * nd
                itype = mm[q]
                q += 1
                if itype == 1:                          # single chunk
                    size, fmask = None, 0
                    if fl & 2:
                        size, fmask = b.ln(q), b.u(q + b.lsz, 4)
                        q += b.lsz + 4
                    obj.layout = ("single", b.off(q), dims[:-1], size, fmask)
                elif itype == 2:                        # implicit: the chunks one after the other, unfiltered
                    obj.layout = ("implicit", b.off(q), dims[:-1])
                else:
                    names = {3: "fixed array", 4: "extensible array", 5: "version-2 B-tree"}
                    raise XmhwException(f"{self.path}: {obj.name}: chunk index '{names.get(itype, itype)}' of the new HDF5 "
                                        f"file format (libver='latest') is not supported; files written with the "
                                        f"default format (version-1 B-tree) are")
            else:
                obj.layout = (f"class {cls}",)
        elif mtype == 0x0B:                             # filter pipeline
            ver, nf = mm[p], mm[p + 1]
            q = p + (8 if ver == 1 else 2)
            for _ in range(nf):
                fid = b.u(q, 2)
                if ver == 1 or fid >= 256:
                    nlen = b.u(q + 2, 2)
                    q += 2
                else:
                    nlen = 0
                ncd = b.u(q + 4, 2)
                q += 6
                q += _pad8(nlen) if ver == 1 else nlen
                cd = [b.u(q + 4 * i, 4) for i in range(ncd)]
                q += 4 * ncd
                if ver == 1 and ncd % 2:
                    q += 4
                obj.filters.append((fid, cd))
        elif mtype == 0x0C:
            name, val, _ = self._attribute(p)
            if name is not None:
                obj.attrs[name] = val
        elif mtype == 0x15:                             # attribute info: dense attributes live in a fractal heap
            fl = mm[p + 1]
            q = p + 2 + (2 if fl & 1 else 0)
            heap = b.off(q)
            if heap != _UNDEF:
                for pos in self._heap_objects(heap, b.off(q + b.osz)):
                    name, val, _ = self._attribute(pos)
                    if name is not None:
                        obj.attrs[name] = val
        elif mtype == 0x06:
            name, addr, _ = self._link(p)
            if name is not None:
                obj.links[name] = addr
        elif mtype == 0x02:                             # link info: dense links
            fl = mm[p + 1]
            q = p + 2 + (8 if fl & 1 else 0)
            heap = b.off(q)
            if heap != _UNDEF:
                for pos in self._heap_objects(heap, b.off(q + b.osz)):
                    name, addr, _ = self._link(pos)
                    if name is not None:
                        obj.links[name] = addr
        elif mtype == 0x11:                             # symbol table: an old-style group
            self._symbol_table(obj, b.off(p), b.off(p + b.osz))

    # ---- attributes, links --------------------------------------------------------------------
    def _attribute(self, p):
        """-> (name, value or None when the type is not decoded, bytes consumed)"""
        b = self.b
        mm = b.mm
        ver = mm[p]
        nsz, dsz, ssz = b.u(p + 2, 2), b.u(p + 4, 2), b.u(p + 6, 2)
        q = p + 8 + (1 if ver == 3 else 0)
        pad = _pad8 if ver == 1 else (lambda n: n)
        name = bytes(mm[q:q + nsz]).split(b"\0")[0].decode("utf-8", "replace")
        q += pad(nsz)
        dt, _ = _parse_datatype(b, q)
        q += pad(dsz)
        shape, _ = _parse_dataspace(b, q) if ssz else ((), 0)
        q += pad(ssz)
        count = int(np.prod(shape)) if shape is not None else 0
        nbytes = count * dt.size
        val = None
        if shape is None:
            val = None
        elif dt.np_dtype is not None and dt.cls in (0, 1):
            a = np.frombuffer(mm, dtype=dt.np_dtype, count=count, offset=q).astype(dt.np_dtype.newbyteorder("="))
            val = a[0] if shape == () or count == 1 else a.copy()
        elif dt.cls == 3:
            s = [bytes(mm[q + i * dt.size:q + (i + 1) * dt.size]).split(b"\0")[0].decode("utf-8", "replace") for i in range(count)]
            val = s[0] if count == 1 else s
        elif dt.cls == 9 and dt.vlen_string:
            s = [self._global_heap_bytes(q + i * dt.size).decode("utf-8", "replace") for i in range(count)]
            val = s[0] if count == 1 else s
        return name, val, q + nbytes - p

    def _link(self, p):
        b = self.b
        mm = b.mm
        fl = mm[p + 1]
        q = p + 2
        ltype = 0
        if fl & 0x08:
            ltype = mm[q]
            q += 1
        if fl & 0x04:
            q += 8
        if fl & 0x10:
            q += 1
        n = 1 << (fl & 3)
        nlen = b.u(q, n)
        q += n
        name = bytes(mm[q:q + nlen]).decode("utf-8", "replace")
        q += nlen
        if ltype != 0:                                  # soft / external links: skipped
            return None, None, None
        return name, b.off(q), q + b.osz - p

    def _global_heap_bytes(self, p):
        """a variable-length element: length(4), collection address, object index(4)"""
        b = self.b
        mm = b.mm
        n = b.u(p, 4)
        coll, idx = b.off(p + 4), b.u(p + 4 + b.osz, 4)
        if n == 0 or coll in (0, _UNDEF):
            return b""
        if mm[coll:coll + 4] != b"GCOL":
            raise XmhwException(f"{self.path}: bad global heap collection at {coll}")
        size = b.ln(coll + 8)
        q, end = coll + 8 + b.lsz, coll + size
        while q + 8 + b.lsz <= end:
            oid, osize = b.u(q, 2), b.ln(q + 8)
            if oid == idx:
                return bytes(mm[q + 8 + b.lsz:q + 8 + b.lsz + n])
            if oid == 0:
                break
            q += 8 + b.lsz + _pad8(osize)
        return b""

    # ---- old-style groups ----------------------------------------------------------------------
    def _symbol_table(self, obj, btree, heap):
        b = self.b
        mm = b.mm
        if mm[heap:heap + 4] != b"HEAP":
            raise XmhwException(f"{self.path}: bad local heap at {heap}")
        data = b.off(heap + 8 + 2 * b.lsz)

        def walk(node):
            if mm[node:node + 4] == b"SNOD":
                n = b.u(node + 6, 2)
                q = node + 8
                for _ in range(n):
                    noff, oaddr = b.off(q), b.off(q + b.osz)
                    name = bytes(mm[data + noff:data + noff + 256]).split(b"\0")[0].decode("utf-8", "replace")
                    obj.links[name] = oaddr
                    q += 2 * b.osz + 24
                return
            if mm[node:node + 4] != b"TREE":
                raise XmhwException(f"{self.path}: bad group B-tree node at {node}")
            used = b.u(node + 6, 2)
            q = node + 8 + 2 * b.osz
            for i in range(used):
                q += b.lsz                              # key i
                walk(b.off(q))
                q += b.osz
        walk(btree)

    # ---- chunk index (B-tree v1, node type 1) ---------------------------------------------------
    def _chunks(self, node, rank):
        b = self.b
        mm = b.mm
        if mm[node:node + 4] != b"TREE" or mm[node + 4] != 1:
            raise XmhwException(f"{self.path}: bad chunk B-tree node at {node}")
        level, used = mm[node + 5], b.u(node + 6, 2)
        q = node + 8 + 2 * b.osz
        ksz = 8 + 8 * (rank + 1)
        for _ in range(used):
            size, fmask = b.u(q, 4), b.u(q + 4, 4)
            offs = tuple(b.u(q + 8 + 8 * i, 8) for i in range(rank))
            child = b.off(q + ksz)
            if level == 0:
                yield offs, fmask, size, child
            else:
                yield from self._chunks(child, rank)
            q += ksz + b.osz

    # ---- fractal heaps (dense links / attributes): every managed object, in storage order ----------------
    def _btree2_records(self, addr):
        """the records of a version-2 B-tree (raw bytes, in key order): the name index of dense attributes (record
        type 8) or dense links (type 5)"""
        b = self.b
        mm = b.mm
        if mm[addr:addr + 4] != b"BTHD":
            raise XmhwException(f"{self.path}: bad version-2 B-tree header at {addr}")
        rtype = mm[addr + 5]
        node_size, rec_size, depth = b.u(addr + 6, 4), b.u(addr + 10, 2), b.u(addr + 12, 2)
        root, nroot = b.off(addr + 16), b.u(addr + 16 + b.osz, 2)
        if root == _UNDEF or nroot == 0:
            return rtype, []
        # H5B2hdr.c: ONE size for the "records in child" field of every node pointer at every depth, that of the LEAF
        # maximum (hdr->max_nrec_size); a pointer to a child of depth >= 1 carries the child's total as well, sized for
        # the most records a subtree of that depth can hold (cum_max_nrec_size)
        def enc(n):
            return (max(n, 1).bit_length() - 1) // 8 + 1
        max_leaf = (node_size - 10) // rec_size
        nrec_sz = enc(max_leaf)
        cum = [max_leaf]                        # [d]: most records a subtree of depth d can hold
        for d in range(1, depth + 1):
            ptr = b.osz + nrec_sz + (enc(cum[d - 1]) if d > 1 else 0)
            max_int = (node_size - 10 - ptr) // (rec_size + ptr)
            cum.append(max_int + (max_int + 1) * cum[d - 1])
        out = []

        def node(a, n, d):
            sig = b"BTLF" if d == 0 else b"BTIN"
            if mm[a:a + 4] != sig:
                raise XmhwException(f"{self.path}: bad version-2 B-tree node at {a}")
            q = a + 6
            recs = [bytes(mm[q + i * rec_size:q + (i + 1) * rec_size]) for i in range(n)]
            q += n * rec_size
            if d == 0:
                out.extend(recs)
                return
            for i in range(n + 1):
                child = b.off(q)
                cn = b.u(q + b.osz, nrec_sz)
                q += b.osz + nrec_sz + (enc(cum[d - 1]) if d > 1 else 0)
                node(child, cn, d - 1)
                if i < n:
                    out.append(recs[i])
        node(root, nroot, depth)
        return rtype, out

    def _heap_objects(self, heap, index=_UNDEF):
        """positions of the objects of a fractal heap: through the heap's name index (a version-2 B-tree whose records
        carry the heap IDs: offset and length inside the heap's address space) where the object header names one,
        otherwise by walking the direct blocks (the objects are self-describing messages) -- and then the number found
        must be the number the heap says it manages: a heap with a free-space gap or a stale rewritten message (files
        edited with ncatted / NCO / CDO) must not silently lose attributes such as scale_factor or _FillValue."""
        b = self.b
        mm = b.mm
        if mm[heap:heap + 4] != b"FRHP":
            raise XmhwException(f"{self.path}: bad fractal heap header at {heap}")
        p = heap + 5
        idlen, flt = b.u(p, 2), b.u(p + 2, 2)
        flags = mm[p + 4]
        p += 5
        max_obj = b.u(p, 4)
        p += 4 + b.lsz + b.osz + b.lsz + b.osz           # next huge id, huge btree, free space, free-space manager
        p += 4 * b.lsz                                   # managed space, allocated, iterator offset, managed objects
        nmanaged = b.ln(p - b.lsz)
        p += 4 * b.lsz                                   # huge size / count, tiny size / count
        width = b.u(p, 2)
        start = b.ln(p + 2)
        maxdirect = b.ln(p + 2 + b.lsz)
        maxheap_bits = b.u(p + 2 + 2 * b.lsz, 2)
        p += 2 + 2 * b.lsz + 2
        p += 2                                           # starting rows of the root indirect block
        root = b.off(p)
        nrows = b.u(p + b.osz, 2)
        if flt:
            raise XmhwException(f"{self.path}: filtered fractal heaps are not supported")
        boff = (maxheap_bits + 7) // 8
        hdr = 5 + b.osz + boff + (4 if flags & 2 else 0)

        def direct_blocks():
            if root == _UNDEF:
                return
            if nrows == 0:
                yield root, start
                return

            def indirect(addr, rows):
                if mm[addr:addr + 4] != b"FHIB":
                    raise XmhwException(f"{self.path}: bad fractal heap indirect block at {addr}")
                q = addr + 5 + b.osz + boff
                for r in range(rows):
                    size = start if r < 2 else start << (r - 1)
                    for _ in range(width):
                        a = b.off(q)
                        q += b.osz
                        if size <= maxdirect:
                            if a != _UNDEF:
                                yield a, size
                        elif a != _UNDEF:
                            # rows of a child indirect block: log2(size / start / width) + 1
                            sub = (size // start // width).bit_length()
                            yield from indirect(a, sub)
            yield from indirect(root, nrows)

        blocks = []
        for addr, size in direct_blocks():
            if mm[addr:addr + 4] != b"FHDB":
                raise XmhwException(f"{self.path}: bad fractal heap direct block at {addr}")
            blocks.append((b.u(addr + 5 + b.osz, boff), size, addr))      # (offset in the heap's address space, ...)

        if index != _UNDEF:
            rtype, recs = self._btree2_records(index)
            if rtype not in (5, 8):
                raise XmhwException(f"{self.path}: version-2 B-tree of type {rtype} is not a name index")
            # a managed heap ID: flags byte (version << 6 | type << 4), offset, length
            len_bytes = min((maxdirect.bit_length() - 1) // 8 + 1, (max(max_obj, 1).bit_length() - 1) // 8 + 1)
            nskipped = 0
            for rec in recs:
                hid = rec[:idlen] if rtype == 8 else rec[4:4 + idlen]
                kind = (hid[0] >> 4) & 3
                if kind != 0:
                    # a "huge" object (> the heap's largest managed size, e.g. a 100 KB NCO history string) lives outside
                    # the heap, a "tiny" one inside its ID: neither can be one of the small numeric attributes the decoding
                    # needs (scale_factor, add_offset, _FillValue, units ...) -- skipped with a warning, the rest of the
                    # object header stays readable
                    import warnings
                    warnings.warn(f"{self.path}: a {'huge' if kind == 1 else 'tiny'} fractal heap object (a very long or "
                                  f"very short attribute / link) is skipped", stacklevel=2)
                    nskipped += 1
                    continue
                off = int.from_bytes(hid[1:1 + boff], "little")
                ln = int.from_bytes(hid[1 + boff:1 + boff + len_bytes], "little")
                for blk_off, size, addr in blocks:
                    if blk_off <= off and off + ln <= blk_off + size:
                        yield addr + (off - blk_off)
                        break
                else:
                    raise XmhwException(f"{self.path}: fractal heap object at offset {off} is in no direct block")
            if len(recs) - nskipped != nmanaged:
                raise XmhwException(f"{self.path}: the name index lists {len(recs) - nskipped} managed objects, the fractal heap manages {nmanaged}")
            return

        found = 0
        for _, size, addr in blocks:
            q, end = addr + hdr, addr + size
            while q + 8 < end and found < nmanaged:
                ver = mm[q]
                if ver == 0 or ver > 3:
                    break
                # an attribute (version 1..3) or a link (version 1) message body; both know their own length
                used = self._try_len(q)
                if used is None or used <= 0:
                    break
                yield q
                found += 1
                q += used
        if found != nmanaged:
            raise XmhwException(f"{self.path}: {found} objects found in a fractal heap that manages {nmanaged} "
                                f"(a free-space gap or a rewritten message, and no name index to go by)")

    def _try_len(self, q):
        """length of the message body at q: a link message starts with version 1 and a flags byte whose unused
        bits are zero; an attribute message carries name / datatype / dataspace sizes"""
        try:
            mm = self.b.mm
            if mm[q] == 1 and (mm[q + 1] & 0xE0) == 0 and self._looks_like_link(q):
                return self._link_len(q)
            return self._attribute(q)[2]
        except Exception:       # noqa: BLE001 -- the end of the used part of a block
            return None

    def _looks_like_link(self, q):
        # (dense LINK heaps hold only links and dense ATTRIBUTE heaps only attributes; an attribute message of
        # version 1 has a zero reserved byte where a link has its flags, and netCDF-4 links always carry the
        # creation-order and link-type flags or at least a name length of 1..255)
        b = self.b
        fl = b.mm[q + 1]
        if fl == 0:
            # flags 0: link with a 1-byte name length directly behind -- or a version-1 attribute (reserved = 0,
            # then the 2-byte name size, whose high byte is 0 for names shorter than 256: the third byte tells)
            return b.mm[q + 3] != 0
        return True

    def _link_len(self, q):
        b = self.b
        fl = b.mm[q + 1]
        p = q + 2
        ltype = 0
        if fl & 0x08:
            ltype = b.mm[p]
            p += 1
        if fl & 0x04:
            p += 8
        if fl & 0x10:
            p += 1
        n = 1 << (fl & 3)
        nlen = b.u(p, n)
        p += n + nlen
        if ltype == 0:
            p += b.osz
        elif ltype == 1:
            p += 2 + b.u(p, 2)
        else:
            p += 2 + b.u(p, 2)
        return p - q
