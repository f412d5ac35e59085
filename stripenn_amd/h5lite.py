"""A small read-only HDF5 reader for cooler files (`.cool`, `.mcool`), numpy only.

`stripenn compute file.mcool::resolutions/5000` needs cooler's tables -- chromosome names / lengths, the bin table's
balancing columns, the index arrays and the three pixel columns (stripenn.py:80-118 opens them through
`cooler.Cooler`).  cooler and h5py are not always installed (they are absent from the interpreter this package is
built and tested under), and the part of HDF5 a cooler file uses is small: old-style groups (symbol tables), object
headers of version 1 (and 2), contiguous / compact / chunked datasets indexed by a version-1 B-tree, the deflate,
shuffle and fletcher32 filters, fixed-point / floating-point / fixed-length string (and enumerated) types, scalar
and simple attributes.  That subset is read here, straight from the "HDF5 File Format Specification" (version 3.0):
what the HDF5 library writes with its default (`libver='earliest'`) settings, which is how cooler, `cooler zoomify`
and hic2cool create their files.  Anything else -- version-4 data layouts, dense (fractal-heap) groups, other filters,
variable-length data -- raises `H5LiteUnsupported` with the name of the construct, so that the caller can say what to
install instead of returning wrong numbers.

The interface is the part of h5py's that `stripenn_amd.pixels.CoolTable` uses: `File(path)[...]`, `Group.keys()`,
`Group.attrs[...]`, `Dataset.shape / dtype / chunks / compression / shuffle / fletcher32`, 1-d slicing, and
`Dataset.id.read_direct_chunk((row,))` -> (filter mask, raw bytes) for readers that inflate chunks themselves.
Reads are positional (`os.pread`): datasets may be read from several threads at once.
"""
import os
import struct
import zlib

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'


class H5LiteError(IOError):
    pass


class H5LiteUnsupported(H5LiteError):
    """a construct of the file format this reader does not implement"""


class _Reader:
    def __init__(self, path):
        self.fd = os.open(path, os.O_RDONLY)
        self.size = os.fstat(self.fd).st_size
        self.base = 0
        self.O = self.L = 8

    def read(self, addr, n):
        if addr < 0 or addr + n > self.size:
            raise H5LiteError('read of %d bytes at %d beyond the end of the file (%d bytes)' % (n, addr, self.size))
        b = os.pread(self.fd, n, addr)
        if len(b) != n:
            raise H5LiteError('short read at %d' % addr)
        return b

    def close(self):
        if self.fd is not None:
            os.close(self.fd)
            self.fd = None

    def uint(self, buf, off, n):
        return int.from_bytes(buf[off:off + n], 'little')

    def undefined(self, addr, n=None):
        return addr == (1 << (8 * (n or self.O))) - 1


def _pad8(n):
    return (n + 7) & ~7


# ------------------------------------------------------------------------------------------------ messages
class _Datatype:
    """what a datatype message says, as a numpy dtype (III.A.2.4.d)"""

    def __init__(self, buf, off=0):
        cv = buf[off]
        self.cls, self.version = cv & 15, cv >> 4
        bits = buf[off + 1] | buf[off + 2] << 8 | buf[off + 3] << 16
        self.size = struct.unpack_from('<I', buf, off + 4)[0]
        self.vlen = False
        bo = '>' if bits & 1 else '<'
        if self.cls == 0:                                      # fixed point
            self.dtype = np.dtype('%s%s%d' % (bo, 'i' if bits & 8 else 'u', self.size))
            self.nbytes = 8 + 4
        elif self.cls == 1:                                    # floating point (IEEE layouts only)
            if self.size not in (2, 4, 8):
                raise H5LiteUnsupported('floating-point type of %d bytes' % self.size)
            self.dtype = np.dtype('%sf%d' % (bo, self.size))
            self.nbytes = 8 + 12
        elif self.cls == 3:                                    # fixed-length string
            self.dtype = np.dtype('S%d' % self.size)
            self.nbytes = 8
        elif self.cls == 8:                                    # enumeration: the base type's values (bins/chrom)
            base = _Datatype(buf, off + 8)
            self.dtype = base.dtype
            n = bits & 0xFFFF
            p = off + 8 + base.nbytes
            for _ in range(n):                                 # member names: null-terminated, padded to 8 (versions 1, 2) or not (3)
                e = buf.index(b'\0', p)
                p = p + _pad8(e - p + 1) if self.version < 3 else e + 1
            self.nbytes = p + n * base.size - off
        elif self.cls == 9:                                    # variable length: data lives in the global heap
            self.vlen = True
            self.dtype = None
            self.nbytes = 8 + _Datatype(buf, off + 8).nbytes
        else:
            raise H5LiteUnsupported('datatype class %d' % self.cls)


def _dataspace(buf):
    """shape from a dataspace message (III.A.2.4.b); () for a scalar, None for a null dataspace"""
    ver, rank, flags = buf[0], buf[1], buf[2]
    if ver == 1:
        off = 8
    elif ver == 2:
        if buf[3] == 2:
            return None
        off = 4
    else:
        raise H5LiteUnsupported('dataspace message version %d' % ver)
    return tuple(struct.unpack_from('<%dQ' % rank, buf, off)) if rank else ()


def _filters(buf):
    """[(filter id, client data)] from a filter pipeline message (III.A.2.4.l)"""
    ver, n = buf[0], buf[1]
    out, p = [], 8 if ver == 1 else 2
    for _ in range(n):
        fid = struct.unpack_from('<H', buf, p)[0]
        p += 2
        nlen = 0
        if ver == 1 or fid >= 256:
            nlen = struct.unpack_from('<H', buf, p)[0]
            p += 2
        _flags, ncd = struct.unpack_from('<HH', buf, p)
        p += 4
        p += _pad8(nlen) if ver == 1 else nlen
        cd = struct.unpack_from('<%dI' % ncd, buf, p)
        p += 4 * ncd
        if ver == 1 and ncd % 2:
            p += 4
        out.append((fid, cd))
    return out


class _Object:
    """the messages of one object header (III.A.2.1)"""

    def __init__(self, rd, addr):
        self.rd, self.addr, self.msgs = rd, addr, []
        head = rd.read(addr, 16)
        if head[:4] == b'OHDR':
            self._v2(addr)
        elif head[0] == 1:
            nmsg = struct.unpack_from('<H', head, 2)[0]
            size = struct.unpack_from('<I', head, 8)[0]
            self._v1_block(addr + 16, size, nmsg)
        else:
            raise H5LiteError('no object header at %d' % addr)

    def _v1_block(self, addr, size, nmsg):
        rd = self.rd
        blocks, seen = [(addr, size)], set()
        while blocks and len(self.msgs) < nmsg:
            a, n = blocks.pop(0)
            if a in seen:
                raise H5LiteError('object header continuation blocks form a cycle at %d' % a)
            seen.add(a)
            buf = rd.read(a, n)
            p = 0
            while p + 8 <= n and len(self.msgs) < nmsg:
                mtype, msize, mflags = struct.unpack_from('<HHB', buf, p)
                if mflags & 0x02 and mtype in (0x03, 0x0B, 0x0C):
                    raise H5LiteUnsupported('shared (committed) datatype / filter / attribute message in the object header at %d' % self.addr)
                body = buf[p + 8:p + 8 + msize]
                p += 8 + msize
                if mtype == 0x10:                               # continuation
                    blocks.append((rd.uint(body, 0, rd.O) + rd.base, rd.uint(body, rd.O, rd.L)))
                self.msgs.append((mtype, body))

    def _v2(self, addr):
        rd = self.rd
        head = rd.read(addr, min(64, rd.size - addr))          # (a header within 64 bytes of the end of the file)
        flags = head[5]
        p = 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        w = 1 << (flags & 3)
        size0 = rd.uint(head, p, w)
        p += w
        track = bool(flags & 0x04)
        blocks, seen = [(addr + p, size0)], set()
        while blocks:
            a, n = blocks.pop(0)
            if a in seen:
                raise H5LiteError('object header continuation blocks form a cycle at %d' % a)
            seen.add(a)
            buf = rd.read(a, n)
            q = 0
            while q + 4 <= n:
                mtype = buf[q]
                msize = struct.unpack_from('<H', buf, q + 1)[0]
                q += 4 + (2 if track else 0)
                body = buf[q:q + msize]
                q += msize
                if mtype == 0x10:
                    ca, cn = rd.uint(body, 0, rd.O) + rd.base, rd.uint(body, rd.O, rd.L)
                    if rd.read(ca, 4) != b'OCHK':
                        raise H5LiteError('object header continuation without signature at %d' % ca)
                    blocks.append((ca + 4, cn - 8))             # (signature in front, checksum behind)
                self.msgs.append((mtype, body))

    def get(self, mtype):
        return [b for t, b in self.msgs if t == mtype]


# ------------------------------------------------------------------------------------------------ objects
class AttributeManager:
    def __init__(self, obj):
        self._obj, self._d = obj, None

    def _load(self):
        if self._d is not None:
            return
        rd = self._obj.rd
        self._d = {}
        for body in self._obj.get(0x0C):
            ver = body[0]
            nsz, tsz, ssz = struct.unpack_from('<HHH', body, 2)
            p = 8 + (1 if ver == 3 else 0)
            pad = _pad8 if ver == 1 else (lambda n: n)
            name = body[p:p + nsz].split(b'\0')[0].decode('utf-8', 'replace')
            p += pad(nsz)
            try:
                dt = _Datatype(body, p)
            except H5LiteUnsupported:
                self._d[name] = None
                continue
            p += pad(tsz)
            shape = _dataspace(body[p:p + ssz])
            p += pad(ssz)
            if dt.vlen or shape is None:                        # (variable-length strings -- cooler's 'format', 'creation-date', ... -- are not needed)
                self._d[name] = None
                continue
            n = int(np.prod(shape)) if shape else 1
            a = np.frombuffer(body, dt.dtype, n, p)
            self._d[name] = a.reshape(shape) if shape else a[0]
        if any(self._obj.get(0x15)) and any(not rd.undefined(rd.uint(b, 2 + (2 if b[1] & 1 else 0), rd.O)) for b in self._obj.get(0x15)):
            raise H5LiteUnsupported('attributes in dense storage (fractal heap)')

    def __getitem__(self, k):
        self._load()
        if self._d.get(k) is None:
            if k in self._d:
                raise H5LiteUnsupported('attribute %r has a variable-length or unsupported type' % k)
            raise KeyError(k)
        return self._d[k]

    def __contains__(self, k):
        self._load()
        return k in self._d

    def keys(self):
        self._load()
        return list(self._d)

    def get(self, k, default=None):
        self._load()
        v = self._d.get(k)
        return default if v is None else v


class Group:
    def __init__(self, rd, obj, name='/'):
        self.rd, self._obj, self.name = rd, obj, name
        self.attrs = AttributeManager(obj)
        self._links = None

    def _load(self):
        if self._links is not None:
            return
        rd, links = self.rd, {}
        for body in self._obj.get(0x11):                        # symbol table: B-tree of symbol nodes + local heap of names
            btree, heap = rd.uint(body, 0, rd.O) + rd.base, rd.uint(body, rd.O, rd.O) + rd.base
            h = rd.read(heap, 8 + 2 * rd.L + rd.O)
            if h[:4] != b'HEAP':
                raise H5LiteError('no local heap at %d' % heap)
            dsize = rd.uint(h, 8, rd.L)
            data = rd.read(rd.uint(h, 8 + 2 * rd.L, rd.O) + rd.base, dsize)
            self._walk_group_tree(btree, data, links)
        for body in self._obj.get(0x06):                        # link messages (compact new-style group)
            flags = body[1]
            p = 2
            ltype = 0
            if flags & 0x08:
                ltype = body[p]
                p += 1
            if flags & 0x04:
                p += 8
            if flags & 0x10:
                p += 1
            w = 1 << (flags & 3)
            n = rd.uint(body, p, w)
            p += w
            nm = body[p:p + n].decode('utf-8', 'replace')
            p += n
            if ltype == 0:
                links[nm] = rd.uint(body, p, rd.O) + rd.base
        for body in self._obj.get(0x02):                        # link info: dense storage when a fractal heap is named
            p = 2 + (8 if body[1] & 1 else 0)
            if not rd.undefined(rd.uint(body, p, rd.O)):
                raise H5LiteUnsupported('group %r keeps its links in a fractal heap (file written with libver="latest")' % self.name)
        self._links = links

    def _walk_group_tree(self, addr, heap, links, depth=0):
        rd = self.rd
        if depth > 32:                                          # (a B-tree this deep does not exist; a cyclic file does)
            raise H5LiteError('group B-tree deeper than 32 levels at %d (corrupt or cyclic file)' % addr)
        head = rd.read(addr, 8 + 2 * rd.O)
        if head[:4] != b'TREE' or head[4] != 0:
            raise H5LiteError('no group B-tree node at %d' % addr)
        level, n = head[5], struct.unpack_from('<H', head, 6)[0]
        body = rd.read(addr + 8 + 2 * rd.O, n * (rd.O + rd.L) + rd.L)
        p = rd.L                                               # key 0
        for _ in range(n):
            child = rd.uint(body, p, rd.O) + rd.base
            p += rd.O + rd.L
            if level:
                self._walk_group_tree(child, heap, links, depth + 1)
                continue
            s = rd.read(child, 8)
            if s[:4] != b'SNOD':
                raise H5LiteError('no symbol node at %d' % child)
            ns = struct.unpack_from('<H', s, 6)[0]
            esz = 2 * rd.O + 24
            ent = rd.read(child + 8, ns * esz)
            for k in range(ns):
                no = rd.uint(ent, k * esz, rd.O)
                oh = rd.uint(ent, k * esz + rd.O, rd.O) + rd.base
                links[heap[no:heap.index(b'\0', no)].decode('utf-8', 'replace')] = oh

    def keys(self):
        self._load()
        return sorted(self._links)

    def __contains__(self, k):
        try:
            self[k]
            return True
        except KeyError:
            return False

    def __iter__(self):
        return iter(self.keys())

    def __getitem__(self, path):
        node = self
        for part in [p for p in str(path).split('/') if p]:
            if not isinstance(node, Group):
                raise KeyError(path)
            node._load()
            if part not in node._links:
                raise KeyError('%s (no %r in %s)' % (path, part, node.name))
            obj = _Object(self.rd, node._links[part])
            nm = node.name.rstrip('/') + '/' + part
            node = Dataset(self.rd, obj, nm) if obj.get(0x08) else Group(self.rd, obj, nm)
        return node


class _DatasetID:
    """the two calls of h5py's low-level dataset id that chunk-wise readers use"""

    def __init__(self, ds):
        self._ds = ds

    def read_direct_chunk(self, offsets):
        return self._ds._raw_chunk(int(offsets[0]))

    def get_create_plist(self):
        ds = self._ds

        class _P:
            def get_nfilters(self_inner):
                return len(ds._filters)
        return _P()


class Dataset:
    def __init__(self, rd, obj, name):
        self.rd, self._obj, self.name = rd, obj, name
        self.attrs = AttributeManager(obj)
        self._dt = _Datatype(obj.get(0x03)[0])
        if self._dt.vlen:
            raise H5LiteUnsupported('dataset %s holds variable-length data' % name)
        self.dtype = self._dt.dtype
        self.shape = _dataspace(obj.get(0x01)[0]) or ()
        self.ndim = len(self.shape)
        self._filters = _filters(obj.get(0x0B)[0]) if obj.get(0x0B) else []
        for fid, _cd in self._filters:
            if fid not in (1, 2, 3):
                raise H5LiteUnsupported('dataset %s: filter %d (only deflate, shuffle and fletcher32 are read)' % (name, fid))
        ids = [f for f, _ in self._filters]
        self.compression = 'gzip' if 1 in ids else None
        self.compression_opts = next((cd[0] for f, cd in self._filters if f == 1 and cd), None)
        self.shuffle, self.fletcher32, self.scaleoffset = 2 in ids, 3 in ids, None
        lay = obj.get(0x08)[0]
        if lay[0] != 3:
            raise H5LiteUnsupported('dataset %s: data layout message version %d (file written with libver="latest"?)' % (name, lay[0]))
        self._class = lay[1]
        self.chunks, self._index = None, None
        if self._class == 0:
            n = struct.unpack_from('<H', lay, 2)[0]
            self._compact = bytes(lay[4:4 + n])
        elif self._class == 1:
            self._addr = rd.uint(lay, 2, rd.O)
        elif self._class == 2:
            nd = lay[2]
            self._btree = rd.uint(lay, 3, rd.O)
            dims = struct.unpack_from('<%dI' % nd, lay, 3 + rd.O)
            self.chunks = tuple(dims[:-1])
            if len(self.chunks) != self.ndim:
                raise H5LiteError('dataset %s: chunk rank differs from the dataspace rank' % name)
        else:
            raise H5LiteUnsupported('dataset %s: layout class %d' % (name, self._class))
        self.id = _DatasetID(self)

    def __len__(self):
        return self.shape[0]

    # ---- chunk index (version-1 B-tree, node type 1): first-dimension offset -> (address, stored size, filter mask)
    def _chunk_index(self):
        if self._index is None:
            idx = {}
            if not self.rd.undefined(self._btree):
                self._walk_chunk_tree(self._btree + self.rd.base, idx)
            self._index = idx
        return self._index

    def _walk_chunk_tree(self, addr, idx, depth=0):
        rd = self.rd
        if depth > 32:
            raise H5LiteError('chunk B-tree deeper than 32 levels at %d (corrupt or cyclic file)' % addr)
        head = rd.read(addr, 8 + 2 * rd.O)
        if head[:4] != b'TREE' or head[4] != 1:
            raise H5LiteError('no chunk B-tree node at %d' % addr)
        level, n = head[5], struct.unpack_from('<H', head, 6)[0]
        ksz = 8 + 8 * (self.ndim + 1)
        body = rd.read(addr + 8 + 2 * rd.O, n * (ksz + rd.O) + ksz)
        for k in range(n):
            p = k * (ksz + rd.O)
            csize, mask = struct.unpack_from('<II', body, p)
            offs = struct.unpack_from('<%dQ' % self.ndim, body, p + 8)
            child = rd.uint(body, p + ksz, rd.O) + rd.base
            if level:
                self._walk_chunk_tree(child, idx, depth + 1)
            else:
                idx[offs] = (child, csize, mask)

    def _raw_chunk(self, row):
        """(filter mask, stored bytes) of the chunk that starts at first-dimension offset `row` (1-d datasets)"""
        if self._class != 2 or self.ndim != 1:
            raise H5LiteError('read_direct_chunk: %s is not a chunked 1-d dataset' % self.name)
        e = self._chunk_index().get((row,))
        if e is None:
            raise H5LiteError('dataset %s has no chunk at row %d' % (self.name, row))
        return e[2], self.rd.read(e[0], e[1])

    def _decode(self, raw, mask, nelem):
        buf = raw
        for pos in range(len(self._filters) - 1, -1, -1):       # the pipeline backwards; bit k of the mask: filter k was skipped
            fid = self._filters[pos][0]
            if mask >> pos & 1:
                continue
            if fid == 3:
                buf = buf[:-4]                                  # fletcher32: the checksum behind the data (not verified)
            elif fid == 1:
                buf = zlib.decompress(buf)
            elif fid == 2:
                isz = self.dtype.itemsize
                if isz > 1:
                    n = len(buf) // isz
                    a = np.frombuffer(buf, np.uint8, n * isz).reshape(isz, n).T
                    buf = np.ascontiguousarray(a).tobytes() + bytes(buf[n * isz:])
        a = np.frombuffer(buf, self.dtype, nelem)
        return a

    def _read_rows(self, lo, hi):
        """rows [lo, hi) of the first dimension as a new array"""
        n0 = self.shape[0] if self.shape else 1
        inner = int(np.prod(self.shape[1:])) if self.ndim > 1 else 1
        isz = self.dtype.itemsize
        out = np.zeros((hi - lo,) + tuple(self.shape[1:]), self.dtype)
        if hi <= lo:
            return out
        if self._class == 0:
            out[...] = np.frombuffer(self._compact, self.dtype, n0 * inner).reshape((n0,) + tuple(self.shape[1:]))[lo:hi]
        elif self._class == 1:
            if not self.rd.undefined(self._addr):
                raw = self.rd.read(self._addr + self.rd.base + lo * inner * isz, (hi - lo) * inner * isz)
                out[...] = np.frombuffer(raw, self.dtype).reshape(out.shape)
        else:
            if self.ndim != 1:
                if any(c != s for c, s in zip(self.chunks[1:], self.shape[1:])):
                    raise H5LiteUnsupported('dataset %s: chunked in more than its first dimension' % self.name)
            c = self.chunks[0]
            cinner = int(np.prod(self.chunks[1:])) if self.ndim > 1 else 1
            idx = self._chunk_index()
            for c0 in range(lo - lo % c, hi, c):
                e = idx.get((c0,) + (0,) * (self.ndim - 1))
                if e is None:
                    continue                                    # never written: the fill value (0)
                a = self._decode(self.rd.read(e[0], e[1]), e[2], c * cinner).reshape((c,) + tuple(self.chunks[1:]))
                s0, s1 = max(lo, c0), min(hi, c0 + c)
                out[s0 - lo:s1 - lo] = a[s0 - c0:s1 - c0]
        return out.astype(self.dtype.newbyteorder('='), copy=False) if not self.dtype.isnative else out

    def __getitem__(self, key):
        if key is Ellipsis or (isinstance(key, tuple) and len(key) == 0):
            a = self._read_rows(0, self.shape[0]) if self.shape else self._read_rows(0, 1)[0]
            return a
        if not self.shape:
            raise IndexError('scalar dataset %s: use [()]' % self.name)
        rest = ()
        if isinstance(key, tuple):
            key, rest = key[0], key[1:]
        n = self.shape[0]
        if isinstance(key, slice):
            lo, hi, step = key.indices(n)
            if step != 1:
                a = self._read_rows(0, n)[key]
            else:
                a = self._read_rows(lo, max(lo, hi))
        elif isinstance(key, (int, np.integer)):
            k = int(key) + (n if key < 0 else 0)
            if not 0 <= k < n:
                raise IndexError('index %d out of range for %s' % (key, self.name))
            a = self._read_rows(k, k + 1)[0]
        else:                                                  # index arrays, boolean masks: through the whole column
            a = self._read_rows(0, n)[key]
        return a[(slice(None),) + tuple(rest)] if rest and isinstance(key, slice) else (a[tuple(rest)] if rest else a)


class File(Group):
    """`h5lite.File(path)` -- read-only; `with` closes it"""

    def __init__(self, path, mode='r'):
        if mode != 'r':
            raise ValueError('h5lite reads only')
        rd = _Reader(path)
        try:
            off = 0
            while True:                                         # the superblock sits at 0, 512, 1024, ...
                if off + 8 > rd.size:
                    raise H5LiteError('%s is not an HDF5 file' % path)
                if rd.read(off, 8) == SIGNATURE:
                    break
                off = 512 if off == 0 else off * 2
            sb = rd.read(off, min(128, rd.size - off))
            ver = sb[8]
            if ver in (0, 1):
                rd.O, rd.L = sb[13], sb[14]
                p = 24 + (4 if ver == 1 else 0)
                rd.base = rd.uint(sb, p, rd.O)
                p += 4 * rd.O                                   # base, free-space info, end of file, driver info
                root = rd.uint(sb, p + rd.O, rd.O)              # root symbol table entry: name offset, object header address
            elif ver in (2, 3):
                rd.O, rd.L = sb[9], sb[10]
                rd.base = rd.uint(sb, 12, rd.O)
                root = rd.uint(sb, 12 + 3 * rd.O, rd.O)
            else:
                raise H5LiteUnsupported('superblock version %d' % ver)
            if off and not rd.base:
                rd.base = off
            Group.__init__(self, rd, _Object(rd, root + rd.base), '/')
        except BaseException:
            rd.close()
            raise
        self.filename = path

    def close(self):
        self.rd.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
