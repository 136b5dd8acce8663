"""Reader for the reference's model files -- Keras 2.x HDF5 (``model.h5`` / ``inference_model.h5`` written by
``keras.Model.save`` at net.py:418-427, or ``model_weights.h5`` written by ``save_weights``), without h5py.

The reference keeps trained weights in HDF5 through Keras (net.py:418-494); h5py/libhdf5 are not a dependency of
this package, so the subset of the HDF5 file format Keras/h5py emit is parsed here in plain Python + numpy:
superblock v0-v3, v1/v2 object headers (+ continuation blocks), old-style groups (symbol table, v1 B-tree, local heap)
and compact new-style groups (link messages), contiguous / compact / chunked (v1 B-tree, optional shuffle + deflate)
datasets of fixed-point / floating-point numbers, and attributes holding numbers, fixed-length strings or
variable-length strings (global heap).  Anything else raises ``KerasH5Error`` naming the unsupported feature.

Layout Keras writes (keras/engine/saving.py, 2.2.x): ``model.save`` puts the weights under the group
``/model_weights`` (``save_weights`` puts them at the root): attribute ``layer_names`` lists the layers in model
order, each layer group has the attribute ``weight_names`` in ``layer.weights`` order and one dataset per name
(the names contain '/', so the datasets sit in nested groups).  ``read_keras_weights`` returns the arrays in exactly
``model.get_weights()`` order -- the flat parameter order of this package.
"""
import struct
import zlib

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class KerasH5Error(ValueError):
    pass


class _Dataset:
    def __init__(self, f, msgs):
        self._f, self._msgs = f, msgs

    def read(self):
        return self._f._read_dataset(self._msgs)


class _Group(dict):
    """name -> _Group | _Dataset, plus ``attrs``."""
    attrs = None


class H5File:
    def __init__(self, path):
        with open(path, "rb") as fh:
            self.buf = fh.read()
        self._parse_superblock()
        self.root = self._load_object(self.root_addr)

    # ------------------------------------------------------------------ low level
    def _u(self, off, size):
        return int.from_bytes(self.buf[off:off + size], "little")

    def _addr(self, off):
        v = self._u(off, self.so)
        return None if v == (1 << (8 * self.so)) - 1 else v + self.base

    def _parse_superblock(self):
        b = self.buf
        off = 0
        while b[off:off + 8] != SIGNATURE:
            off = 512 if off == 0 else off * 2
            if off + 8 > len(b):
                raise KerasH5Error("not an HDF5 file (signature not found)")
        ver = b[off + 8]
        if ver in (0, 1):
            self.so, self.sl = b[off + 13], b[off + 14]
            p = off + 24 + (4 if ver == 1 else 0)
            self.base = 0
            self.base = self._u(p, self.so)
            p += 4 * self.so                                   # base, free-space, end-of-file, driver block
            # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
            self.root_addr = self._u(p + self.so, self.so) + self.base
        elif ver in (2, 3):
            self.so, self.sl = b[off + 9], b[off + 10]
            p = off + 12
            self.base = self._u(p, self.so)
            self.root_addr = self._u(p + 3 * self.so, self.so) + self.base
        else:
            raise KerasH5Error(f"unsupported HDF5 superblock version {ver}")
        if self.so not in (4, 8) or self.sl not in (4, 8):
            raise KerasH5Error("unsupported offset/length size")

    # ------------------------------------------------------------------ object headers
    def _messages(self, addr):
        """[(type, flags, payload offset, payload size)] of the object header at addr (v1 or v2)."""
        b = self.buf
        out = []
        if b[addr:addr + 4] == b"OHDR":
            if b[addr + 4] != 2:
                raise KerasH5Error("unsupported object header version")
            flags = b[addr + 5]
            p = addr + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            nsz = 1 << (flags & 3)
            chunk0 = self._u(p, nsz)
            p += nsz
            blocks = [(p, chunk0)]
            track = bool(flags & 0x04)
            while blocks:
                p, size = blocks.pop(0)
                end = p + size
                while p + 4 <= end:
                    mtype, msize, mflags = b[p], self._u(p + 1, 2), b[p + 3]
                    p += 4 + (2 if track else 0)
                    if mtype == 0x10:
                        caddr, clen = self._addr(p), self._u(p + self.so, self.sl)
                        if b[caddr:caddr + 4] != b"OCHK":
                            raise KerasH5Error("bad object header continuation block")
                        blocks.append((caddr + 4, clen - 8))           # minus signature and checksum
                    elif mtype != 0:
                        out.append((mtype, mflags, p, msize))
                    p += msize
            return out
        if b[addr] != 1:
            raise KerasH5Error(f"unsupported object header version {b[addr]} at {addr}")
        nmsg = self._u(addr + 2, 2)
        blocks = [(addr + 16, self._u(addr + 8, 4))]
        while blocks and len(out) < nmsg + 64:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end:
                mtype, msize, mflags = self._u(p, 2), self._u(p + 2, 2), b[p + 4]
                p += 8
                if mtype == 0x10:
                    blocks.append((self._addr(p), self._u(p + self.so, self.sl)))
                elif mtype != 0:
                    out.append((mtype, mflags, p, msize))
                p += msize
        return out

    def _load_object(self, addr, depth=0):
        if depth > 32:
            raise KerasH5Error("group nesting too deep (cycle?)")
        msgs = self._messages(addr)
        types = {m[0] for m in msgs}
        if 0x08 in types:                                        # data layout message: a dataset
            return _Dataset(self, msgs)
        grp = _Group()
        grp.attrs = _LazyAttrs(self, msgs)
        for mtype, _, p, size in msgs:
            if mtype == 0x11:                                    # symbol table: old-style group
                for name, child in self._symbol_table(self._addr(p), self._addr(p + self.so)):
                    grp[name] = self._load_object(child, depth + 1)
            elif mtype == 0x06:                                  # link message: compact new-style group
                name, child = self._link(p)
                if child is not None:
                    grp[name] = self._load_object(child, depth + 1)
            elif mtype == 0x02:                                  # link info: dense storage if a fractal heap is named
                v_flags = self.buf[p + 1]
                q = p + 2 + (8 if v_flags & 1 else 0)
                if self._addr(q) is not None:
                    raise KerasH5Error("dense link storage (fractal heap groups) is not supported; re-save the model "
                                       "with h5py's default libver")
        return grp

    def _link(self, p):
        b = self.buf
        if b[p] != 1:
            raise KerasH5Error("unsupported link message version")
        flags = b[p + 1]
        q = p + 2
        ltype = 0
        if flags & 0x08:
            ltype = b[q]; q += 1
        if flags & 0x04:
            q += 8
        if flags & 0x10:
            q += 1
        nsz = 1 << (flags & 3)
        nlen = self._u(q, nsz); q += nsz
        name = b[q:q + nlen].decode("utf8"); q += nlen
        return name, (self._addr(q) if ltype == 0 else None)

    def _symbol_table(self, btree, heap):
        b = self.buf
        if b[heap:heap + 4] != b"HEAP":
            raise KerasH5Error("bad local heap")
        data = self._addr(heap + 8 + 2 * self.sl)
        out = []

        def walk(node):
            if b[node:node + 4] != b"TREE" or b[node + 4] != 0:
                raise KerasH5Error("bad group B-tree node")
            level, used = b[node + 5], self._u(node + 6, 2)
            p = node + 8 + 2 * self.so
            for i in range(used):
                child = self._addr(p + self.sl + i * (self.sl + self.so))
                if level > 0:
                    walk(child)
                    continue
                if b[child:child + 4] != b"SNOD":
                    raise KerasH5Error("bad symbol table node")
                nsym = self._u(child + 6, 2)
                e = child + 8
                for _ in range(nsym):
                    noff = self._u(e, self.so)
                    s = data + noff
                    name = b[s:b.index(b"\0", s)].decode("utf8")
                    out.append((name, self._addr(e + self.so)))
                    e += 2 * self.so + 24
        walk(btree)
        return out

    # ------------------------------------------------------------------ datatypes / dataspaces
    def _datatype(self, p):
        """-> (kind, numpy dtype or None, element size, total bytes of the message)"""
        b = self.buf
        cls, bits0 = b[p] & 0x0F, b[p + 1]
        size = self._u(p + 4, 4)
        if cls == 0:                                              # fixed point
            order = ">" if bits0 & 1 else "<"
            signed = bool(bits0 & 0x08)
            return "num", np.dtype(f"{order}{'i' if signed else 'u'}{size}"), size, 8 + 4
        if cls == 1:                                              # floating point
            order = ">" if bits0 & 1 else "<"
            if size not in (2, 4, 8):
                raise KerasH5Error(f"unsupported float size {size}")
            return "num", np.dtype(f"{order}f{size}"), size, 8 + 12
        if cls == 3:                                              # fixed-length string
            return "str", None, size, 8
        if cls == 9:                                              # variable length
            if (bits0 & 0x0F) != 1:
                raise KerasH5Error("variable-length sequences are not supported")
            _, _, _, base_len = self._datatype(p + 8)
            return "vstr", None, size, 8 + base_len
        raise KerasH5Error(f"unsupported datatype class {cls}")

    def _dataspace(self, p):
        b = self.buf
        ver, rank, flags = b[p], b[p + 1], b[p + 2]
        if ver == 1:
            q = p + 8
        elif ver == 2:
            if b[p + 3] == 2:                                     # null dataspace
                return None
            q = p + 4
        else:
            raise KerasH5Error("unsupported dataspace version")
        return tuple(self._u(q + i * self.sl, self.sl) for i in range(rank))

    def _decode(self, kind, dt, esize, shape, raw):
        n = int(np.prod(shape)) if shape else 1
        if kind == "num":
            a = np.frombuffer(raw, dtype=dt, count=n).reshape(shape)
            return a.astype(dt.newbyteorder("="))
        if kind == "str":
            vals = [raw[i * esize:(i + 1) * esize].split(b"\0")[0] for i in range(n)]
        else:                                                     # vlen strings: (length 4, heap address, index 4)
            vals = []
            step = 4 + self.so + 4
            for i in range(n):
                q = i * step
                ln = int.from_bytes(raw[q:q + 4], "little")
                addr = int.from_bytes(raw[q + 4:q + 4 + self.so], "little") + self.base
                idx = int.from_bytes(raw[q + 4 + self.so:q + step], "little")
                vals.append(self._global_heap_object(addr, idx)[:ln])
        if not shape:
            return vals[0]
        return np.array(vals, dtype=object).reshape(shape)

    def _global_heap_object(self, addr, idx):
        b = self.buf
        if b[addr:addr + 4] != b"GCOL":
            raise KerasH5Error("bad global heap collection")
        end = addr + self._u(addr + 8, self.sl)
        p = addr + 8 + self.sl
        while p + 8 + self.sl <= end:
            oidx, osize = self._u(p, 2), self._u(p + 8, self.sl)
            if oidx == 0:
                break
            if oidx == idx:
                return b[p + 8 + self.sl:p + 8 + self.sl + osize]
            p += 8 + self.sl + ((osize + 7) & ~7)
        raise KerasH5Error("global heap object not found")

    # ------------------------------------------------------------------ attributes
    def _attribute(self, p):
        b = self.buf
        ver = b[p]
        nlen, dlen, slen = self._u(p + 2, 2), self._u(p + 4, 2), self._u(p + 6, 2)
        if ver == 1:
            pad = lambda v: (v + 7) & ~7
            q = p + 8
        elif ver in (2, 3):
            if b[p + 1] & 3:
                raise KerasH5Error("shared attribute datatypes/dataspaces are not supported")
            pad = lambda v: v
            q = p + 8 + (1 if ver == 3 else 0)
        else:
            raise KerasH5Error("unsupported attribute message version")
        name = b[q:q + nlen].split(b"\0")[0].decode("utf8"); q += pad(nlen)
        kind, dt, esize, _ = self._datatype(q); q += pad(dlen)
        shape = self._dataspace(q); q += pad(slen)
        if shape is None:
            return name, None
        n = int(np.prod(shape)) if shape else 1
        return name, self._decode(kind, dt, esize, shape, b[q:q + n * esize])

    # ------------------------------------------------------------------ datasets
    def _read_dataset(self, msgs):
        b = self.buf
        kind = dt = esize = shape = None
        layout = None
        filters = []
        for mtype, mflags, p, size in msgs:
            if mflags & 0x02:
                raise KerasH5Error("shared header messages are not supported")
            if mtype == 0x03:
                kind, dt, esize, _ = self._datatype(p)
            elif mtype == 0x01:
                shape = self._dataspace(p)
            elif mtype == 0x08:
                layout = p
            elif mtype == 0x0B:
                filters = self._filters(p)
        if kind != "num" or shape is None or layout is None:
            raise KerasH5Error("dataset is not a plain numeric array")
        n = int(np.prod(shape)) if shape else 1
        nbytes = n * esize
        ver = b[layout]
        if ver in (3, 4):                                          # v4 differs from v3 only for chunked / virtual storage
            cls = b[layout + 1]
            if ver == 4 and cls >= 2:
                raise KerasH5Error("chunk-indexed (layout v4) datasets are not supported; re-save with h5py's default libver")
            if cls == 0:
                sz = self._u(layout + 2, 2)
                raw = b[layout + 4:layout + 4 + sz]
            elif cls == 1:
                addr = self._addr(layout + 2)
                raw = b[addr:addr + nbytes] if addr is not None else bytes(nbytes)
            elif cls == 2:
                rank = b[layout + 2]
                btree = self._addr(layout + 3)
                cdims = [self._u(layout + 3 + self.so + 4 * i, 4) for i in range(rank)]
                raw = self._read_chunks(btree, shape, cdims[:-1], esize, filters)
            else:
                raise KerasH5Error("unsupported data layout class")
        elif ver in (1, 2):
            rank, cls = b[layout + 1], b[layout + 2]
            if cls != 1:
                raise KerasH5Error("only contiguous storage is supported for layout message version 1/2")
            addr = self._addr(layout + 8)
            raw = b[addr:addr + nbytes]
        else:
            raise KerasH5Error(f"unsupported data layout message version {ver} (re-save with h5py's default libver)")
        if len(raw) < nbytes:
            raise KerasH5Error("truncated dataset")
        return self._decode("num", dt, esize, shape, raw[:nbytes])

    def _filters(self, p):
        b = self.buf
        ver, nf = b[p], b[p + 1]
        q = p + (8 if ver == 1 else 2)
        out = []
        for _ in range(nf):
            fid = self._u(q, 2)
            if ver == 1 or fid >= 256:
                nlen = self._u(q + 2, 2); q += 4
            else:
                nlen = 0; q += 2
            ncd = self._u(q + 2, 2); q += 4
            q += (nlen + 7) & ~7 if ver == 1 else nlen
            cd = [self._u(q + 4 * i, 4) for i in range(ncd)]
            q += 4 * ncd
            if ver == 1 and ncd & 1:
                q += 4
            out.append((fid, cd))
        return out

    def _read_chunks(self, btree, shape, cdims, esize, filters):
        b = self.buf
        rank = len(shape)
        full = np.zeros(tuple(shape) + (esize,), dtype=np.uint8)
        if btree is None:
            return full.tobytes()

        def walk(node):
            if b[node:node + 4] != b"TREE" or b[node + 4] != 1:
                raise KerasH5Error("bad chunk B-tree node")
            level, used = b[node + 5], self._u(node + 6, 2)
            keysz = 8 + 8 * (rank + 1)
            p = node + 8 + 2 * self.so
            for i in range(used):
                k = p + i * (keysz + self.so)
                csize, mask = self._u(k, 4), self._u(k + 4, 4)
                offs = [self._u(k + 8 + 8 * d, 8) for d in range(rank)]
                child = self._addr(k + keysz)
                if level > 0:
                    walk(child)
                    continue
                raw = b[child:child + csize]
                for j, (fid, cd) in reversed(list(enumerate(filters))):
                    if mask & (1 << j):
                        continue
                    if fid == 1:
                        raw = zlib.decompress(raw)
                    elif fid == 2:
                        ne = len(raw) // esize
                        raw = np.frombuffer(raw, np.uint8)[:ne * esize].reshape(esize, ne).T.tobytes()
                    else:
                        raise KerasH5Error(f"unsupported HDF5 filter {fid}")
                chunk = np.frombuffer(raw, np.uint8, count=int(np.prod(cdims)) * esize).reshape(tuple(cdims) + (esize,))
                sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shape))
                csl = tuple(slice(0, s.stop - s.start) for s in sl)
                full[sl] = chunk[csl]
        walk(btree)
        return full.tobytes()


class _LazyAttrs(dict):
    """Attributes of a group, decoded on first access (``model.h5`` carries a large JSON model_config nobody needs)."""

    def __init__(self, f, msgs):
        super().__init__()
        self._f, self._msgs, self._done = f, msgs, False

    def _load(self):
        if not self._done:
            self._done = True
            for mtype, mflags, p, size in self._msgs:
                if mtype == 0x0C:
                    if mflags & 0x02:
                        raise KerasH5Error("shared attribute messages are not supported")
                    name, val = self._f._attribute(p)
                    dict.__setitem__(self, name, val)

    def __getitem__(self, k):
        self._load(); return dict.__getitem__(self, k)

    def __contains__(self, k):
        self._load(); return dict.__contains__(self, k)

    def keys(self):
        self._load(); return dict.keys(self)


def _names(attrs, key):
    """Keras splits long name lists into <key>0, <key>1, ... (saving.py save_attributes_to_hdf5_group)."""
    if key in attrs:
        vals = list(np.asarray(attrs[key]).reshape(-1))
    else:
        vals, i = [], 0
        while f"{key}{i}" in attrs:
            vals += list(np.asarray(attrs[f"{key}{i}"]).reshape(-1)); i += 1
    return [v.decode("utf8") if isinstance(v, bytes) else str(v) for v in vals]


def read_keras_weights(path):
    """Arrays of a Keras HDF5 model / weights file in ``model.get_weights()`` order (layer order of ``layer_names``,
    per layer the order of ``weight_names``).  Returns (list of numpy arrays, list of their Keras names)."""
    f = H5File(path)
    g = f.root
    if "model_weights" in g:                                      # written by model.save (net.py:418-427)
        g = g["model_weights"]
    if "layer_names" not in g.attrs:
        raise KerasH5Error("no 'layer_names' attribute: not a Keras weights file")
    arrays, names = [], []
    for layer in _names(g.attrs, "layer_names"):
        lg = g[layer]
        for wname in _names(lg.attrs, "weight_names"):
            node = lg
            for part in wname.split("/"):
                if not isinstance(node, dict) or part not in node:
                    raise KerasH5Error(f"weight {wname} of layer {layer} not found in the file")
                node = node[part]
            if not isinstance(node, _Dataset):
                raise KerasH5Error(f"{layer}/{wname} is not a dataset")
            arrays.append(node.read())
            names.append(wname)
    return arrays, names
