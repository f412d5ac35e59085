"""Contact matrices as cooler's own tables -- the data format on the input side of the hot path.

A .cool file is three tables (cooler schema v3): ``bins`` (chrom, start, end, one column per balancing
vector), ``pixels`` (bin1_id <= bin2_id, count; sorted by (bin1_id, bin2_id)) and ``indexes``
(chrom_offset).  ``cooler.Cooler(...).matrix(balance=name).fetch(...)`` -- the only way the reference
reads it (stripenn.py:80-118, getStripe.py .fetch sites) -- turns a rectangle of that table into a dense
array: the count block times np.outer(bias1, bias2), i.e. value = count * (b[bin1] * b[bin2]) with b = w for a
multiplicative column ("weight") and b = 1 / w for the divisive ones cooler knows by name (KR, VC, SQRT_VC:
the columns hic2cool writes; `--norm KR` is the reference CLI's default), mirrored below the diagonal; a cell
without a stored pixel is the count 0 times the same product, i.e. NaN along the whole row and column of a bin
whose weight is NaN and 0 elsewhere.  ``PixelTable`` holds the same arrays; ``PixelSelector.fetch`` is that dense read
on the host, and ``chrom_pixels`` hands the cis pixels of one chromosome to the HIP band packer
(``stp_band_pack``), which builds the resident diagonal band without any dense intermediate.

cooler is absent from the build image (parity of this reader against cooler itself is unpinned); the
arithmetic above restates cooler's `matrix()` dense branch (`arr * np.outer(bias1, bias2)`, `bias = 1 / bias`
under `divisive_weights`, which defaults to `balance in {"KR", "VC", "SQRT_VC"}`).  Tables travel as .npz (``save`` / ``load``); a
.cool / .mcool group is read through h5py when it is importable, else through stripenn_amd.h5lite.
"""
import os

import numpy as np

DIVISIVE_WEIGHTS = ('KR', 'VC', 'SQRT_VC')       # cooler's _4DN_DIVISIVE_WEIGHTS


def _counts(count):
    """pixels/count as one of the two column types the C ABI takes: int32 (cooler's default; wider integer columns
    and integer-valued float columns that fit are narrowed) or float64 (coolers written with --count-as-float,
    merged or scaled coolers: kept as they are, never truncated)."""
    c = np.asarray(count)
    if c.dtype == np.int32:
        return np.ascontiguousarray(c)
    if c.dtype.kind == 'f':
        if c.size == 0 or (np.all(np.isfinite(c) & (c == np.floor(c))) and c.min() >= -2**31 and c.max() <= 2**31 - 1):
            return np.ascontiguousarray(c, dtype=np.int32)
        return np.ascontiguousarray(c, dtype=np.float64)
    if c.size and (c.min() < -2**31 or c.max() > 2**31 - 1):
        return np.ascontiguousarray(c, dtype=np.float64)            # exact below 2^53
    return np.ascontiguousarray(c, dtype=np.int32)


def open_hdf5(path, backend=None):
    """The file through h5py when it is importable, else through the package's own reader of the HDF5 subset cooler
    files use (stripenn_amd.h5lite); `backend` = 'h5py' / 'h5lite' forces one (tests)."""
    if backend in (None, 'h5py'):
        try:
            import h5py
            return h5py.File(path, 'r')
        except ImportError:
            if backend == 'h5py':
                raise
    from . import h5lite
    return h5lite.File(path)


class PixelTable:
    def __init__(self, chromnames, chromsizes, binsize, chrom_offset, bin1_id, bin2_id, count, weights=None):
        self.chromnames = [str(c) for c in chromnames]
        self.chromsizes = np.asarray(chromsizes, dtype=np.int64)
        self.binsize = int(binsize)
        self.chrom_offset = np.asarray(chrom_offset, dtype=np.int64)          # len(chromnames) + 1
        self.bin1_id = np.ascontiguousarray(bin1_id, dtype=np.int64)
        # bin2_id keeps a 32-bit type when it arrives as one (a bin id fits; the column then crosses PCIe at half the bytes)
        b2 = np.asarray(bin2_id)
        self.bin2_id = np.ascontiguousarray(b2, dtype=np.int32 if b2.dtype == np.int32 else np.int64)
        self.count = _counts(count)
        self.weights = {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in (weights or {}).items()}
        if len(self.chrom_offset) != len(self.chromnames) + 1:
            raise ValueError('chrom_offset must have one entry per chromosome plus one')
        if not (len(self.bin1_id) == len(self.bin2_id) == len(self.count)):
            raise ValueError('pixel columns differ in length')
        self._trans = {}              # chromosome -> does its row range hold trans pixels (asked once per chromosome)
        if len(self.bin1_id) and (np.any(np.diff(self.bin1_id) < 0) or np.any(self.bin1_id > self.bin2_id)):
            raise ValueError('pixels must be sorted by bin1_id and upper-triangular (bin1_id <= bin2_id)')
        # cooler's indexes/bin1_offset: first pixel of every bin (bin1_id is sorted, so the column is this index expanded).
        # The band packer takes the index instead of the column (stp_band_pack_csr): 8 bytes per BIN instead of per pixel.
        self.bin1_offset = np.searchsorted(self.bin1_id, np.arange(int(self.chrom_offset[-1]) + 1), side='left').astype(np.int64)

    # ------------------------------------------------------------------ geometry
    def chrom_index(self, chrom):
        return self.chromnames.index(str(chrom))

    def chrom_bins(self, chrom):
        k = self.chrom_index(chrom)
        return int(self.chrom_offset[k]), int(self.chrom_offset[k + 1])

    def weight(self, balance, divisive=None):
        """The multiplicative bias vector of a balancing column.  balance: False / None / 'NONE' -> None (raw
        counts); True -> 'weight'; a name -> that column.  divisive: None -> cooler's rule (KR, VC, SQRT_VC are
        divisive: bias = 1 / w); True / False overrides it."""
        if balance in (False, None, 'NONE'):
            return None
        name = 'weight' if balance is True else str(balance)
        if name not in self.weights:
            raise ValueError('no balancing column %r in the pixel table' % name)
        if divisive is None:
            divisive = name in DIVISIVE_WEIGHTS
        w = self.weights[name]
        if divisive:
            with np.errstate(divide='ignore', invalid='ignore'):
                return 1.0 / w
        return w

    def rows_slice(self, g0, g1):
        """Index range of the pixels whose bin1_id lies in [g0, g1) (global bin ids)."""
        nb = len(self.bin1_offset) - 1
        if 0 <= g0 <= nb and 0 <= g1 <= nb:                   # the CSR index answers it
            return int(self.bin1_offset[g0]), int(self.bin1_offset[g1])
        a, b = np.searchsorted(self.bin1_id, [g0, g1], side='left')
        return int(a), int(b)

    def count_nonnegative(self):
        # (one pass over the whole column -- 25 ms for a genome's 266 M pixels -- so asked once per table, not once per matrix view)
        if getattr(self, '_count_nonneg', None) is None:
            self._count_nonneg = bool(len(self.count) == 0 or self.count.min() >= 0)
        return self._count_nonneg

    def prefetch(self, chrom):
        """(in-memory table: nothing to read ahead)"""

    def chrom_pixels(self, chrom):
        """cis pixels of one chromosome (views, global bin ids) + its first bin and bin count."""
        lo, hi = self.chrom_bins(chrom)
        a, b = self.rows_slice(lo, hi)
        b1, b2 = self.bin1_id[a:b], self.bin2_id[a:b]
        # trans pixels?  Pixels are sorted by (bin1, bin2), so the largest bin2 of a row is its LAST pixel: looking at
        # the row ends (one look-up per bin) answers it without a pass over the whole column
        if len(b2):
            trans = self._trans.get(chrom)
            if trans is None:
                ends = self.bin1_offset[lo + 1:hi + 1] - a - 1       # (the CSR index: no search per bin -- 0.1 s of a genome's first run)
                ends = ends[ends >= 0]
                trans = self._trans[chrom] = bool(b2[ends].max() >= hi)
            if trans:                                        # drop trans pixels
                keep = b2 < hi
                return b1[keep], b2[keep], self.count[a:b][keep], lo, hi - lo
        return b1, b2, self.count[a:b], lo, hi - lo

    def chrom_offsets(self, chrom):
        """CSR index of chrom_pixels(chrom)'s rows (position of the first pixel of every bin of the chromosome, + the
        pixel count), or None when the chromosome's rows hold trans pixels (the filtered columns have no stored index)."""
        lo, hi = self.chrom_bins(chrom)
        trans = self._trans.get(chrom)                  # (the key chrom_pixels stores its verdict under)
        if trans is None:                               # not asked yet: chrom_pixels establishes the verdict (one look-up per bin)
            self.chrom_pixels(chrom)
            trans = self._trans.get(chrom, False)       # (a chromosome without pixels leaves no verdict: nothing was dropped)
        if trans:                                       # at least one pixel was dropped: the stored index does not describe the kept rows
            return None
        return self.bin1_offset[lo:hi + 1] - self.bin1_offset[lo]

    # ------------------------------------------------------------------ I/O
    def save(self, path, compressed=False):
        """.npz; uncompressed by default (loading a deflated 14 M-pixel table costs more than the whole GPU run)."""
        d = dict(chromnames=np.array(self.chromnames), chromsizes=self.chromsizes, binsize=self.binsize,
                 chrom_offset=self.chrom_offset, bin1_id=self.bin1_id, bin2_id=self.bin2_id, count=self.count)
        for k, v in self.weights.items():
            d['weight__' + k] = v
        (np.savez_compressed if compressed else np.savez)(path, **d)

    @classmethod
    def load(cls, path):
        z = np.load(path, allow_pickle=False)
        w = {k[len('weight__'):]: z[k] for k in z.files if k.startswith('weight__')}
        return cls([str(c) for c in z['chromnames']], z['chromsizes'], int(z['binsize']), z['chrom_offset'],
                   z['bin1_id'], z['bin2_id'], z['count'], w)

    @classmethod
    def from_cool(cls, path, group=None):
        """Read the tables of a .cool file / an .mcool resolution group (h5py, or the built-in reader: no cooler needed)."""
        with open_hdf5(path) as f:
            g = f[group] if group else f
            names = [c.decode() if isinstance(c, bytes) else str(c) for c in g['chroms/name'][:]]
            sizes = g['chroms/length'][:]
            binsize = int(g.attrs['bin-size'])
            w = {k: g['bins'][k][:] for k in g['bins'].keys() if k not in ('chrom', 'start', 'end')}
            co = g['indexes/chrom_offset'][:]
            b2 = g['pixels/bin2_id'][:]
            if int(co[-1]) < 2**31:
                b2 = b2.astype(np.int32)                     # (as the lazy reader hands it over: a bin id fits 32 bits)
            return cls(names, sizes, binsize, co, g['pixels/bin1_id'][:], b2, g['pixels/count'][:], w)

    @classmethod
    def from_synth(cls, names, chroms, resol, hw_limit=None):
        """Pixel table of synthetic chromosomes (stripenn_amd.synth.SynthChrom): the raw counts of the upper
        triangle, a 'weight' column (NaN on the masked bins) when the chromosome is balanced."""
        from . import synth
        b1, b2, cn, ws, off, sizes = [], [], [], [], [0], []
        for nm in names:
            ch = chroms[nm]
            lo = off[-1]
            lim = synth.BAND_LIMIT if hw_limit is None else int(hw_limit)
            for r0 in range(0, ch.nbins, 2048):
                r1 = min(r0 + 2048, ch.nbins)
                c1 = min(r1 + lim, ch.nbins)
                cnt = ch.counts(r0, r1, r0, c1)
                ii, jj = np.nonzero(cnt)
                keep = (jj + r0) >= (ii + r0)
                ii, jj = ii[keep], jj[keep]
                b1.append(ii + r0 + lo); b2.append(jj + r0 + lo); cn.append(cnt[ii, jj].astype(np.int32))
            w = ch.w.astype(np.float64).copy()
            w[ch.nan_bins] = np.nan
            ws.append(w)
            off.append(lo + ch.nbins)
            sizes.append(ch.nbins * int(resol))
        weights = {'weight': np.concatenate(ws)} if any(chroms[n].balanced for n in names) else {}
        return cls(names, sizes, resol, off, np.concatenate(b1), np.concatenate(b2), np.concatenate(cn), weights)


class _ChunkReader:
    """One chunked, deflate-compressed 1-d HDF5 column decoded outside HDF5's filter pipeline (CoolTable._read_piece):
    raw chunk -> zlib inflate -> byte unshuffle -> the wanted rows copied into the caller's array."""

    def __init__(self, ds, shuffle):
        self.ds, self.id, self.shuffle = ds, ds.id, shuffle
        self.clen, self.dtype, self.n = int(ds.chunks[0]), np.dtype(ds.dtype), int(ds.shape[0])

    @classmethod
    def of(cls, ds):
        """a reader for `ds`, or None when its layout is not chunked 1-d deflate (+ shuffle) or h5py is too old"""
        try:
            if ds.ndim == 1 and ds.chunks and ds.compression == 'gzip' and not ds.fletcher32 and ds.scaleoffset is None \
                    and hasattr(ds.id, 'read_direct_chunk') and ds.dtype.kind in 'iuf' and ds.dtype.isnative:
                if ds.id.get_create_plist().get_nfilters() == (2 if ds.shuffle else 1):
                    return cls(ds, bool(ds.shuffle))
        except Exception:      # noqa: BLE001 -- anything unexpected about the dataset: h5py's ordinary read
            pass
        return None

    def into(self, c0, x, y, out):
        """rows [max(c0, x), min(c0 + clen, y)) of the chunk that starts at row c0 into out[. - x]; returns 1 when the
        chunk was decoded here, 0 when it went through h5py (a chunk stored without its filters)"""
        import zlib
        lo, hi = max(c0, x), min(c0 + self.clen, y, self.n)
        if hi <= lo:
            return 0
        mask, raw = self.id.read_direct_chunk((c0,))
        if mask != 0:                                        # some filter was skipped for this chunk by the writer
            out[lo - x:hi - x] = self.ds[lo:hi]
            return 0
        buf = zlib.decompress(raw)
        isz = self.dtype.itemsize
        if len(buf) != self.clen * isz:
            raise IOError('pixel column chunk at row %d inflates to %d bytes, %d expected' % (c0, len(buf), self.clen * isz))
        if self.shuffle and isz > 1:                         # byte plane k of all elements, then plane k + 1, ...
            planes = np.frombuffer(buf, np.uint8).reshape(isz, self.clen)
            vals = np.empty((hi - lo, isz), np.uint8)
            vals[:] = planes[:, lo - c0:hi - c0].T
            out[lo - x:hi - x] = vals.view(self.dtype).ravel()
        else:
            out[lo - x:hi - x] = np.frombuffer(buf, self.dtype)[lo - c0:hi - c0]
        return 1


class CoolTable:
    """cooler's tables read LAZILY from the .cool file / .mcool resolution group (h5py or h5lite): what replaces
    `cooler.Cooler(cool)` + `matrix(balance=norm)` (stripenn.py:80, 118) when cooler is absent, at real file sizes.

    Only the small tables are read at once (chromosome names / lengths, `indexes/chrom_offset`, `indexes/bin1_offset`,
    the balancing columns of `bins`).  The pixel columns stay in the file: `rows_slice` is cooler's own index look-up
    (`bin1_offset`, no search over a column), `bin1_id` / `bin2_id` / `count` are sliceable views that read what is
    asked for, and `chrom_pixels` reads ONE chromosome's rows in pieces of `chunk` pixels, keeping the cis pixels only
    (trans pixels are dropped piece by piece, never accumulated).  `prefetch(chrom)` starts that read on a host thread
    (h5py releases the GIL inside HDF5's read / inflate), so the next chromosome's columns arrive while the device
    packs and searches the current one: the host holds at most two chromosomes' cis columns.
    Same interface as PixelTable as far as PixelSelector and the facade use it."""

    def __init__(self, path, group=None, chunk=1 << 20, backend=None):
        self._h5 = open_hdf5(path, backend)
        g = self._h5[group] if group else self._h5
        self._g = g
        self.chromnames = [c.decode() if isinstance(c, bytes) else str(c) for c in g['chroms/name'][:]]
        self.chromsizes = np.asarray(g['chroms/length'][:], dtype=np.int64)
        self.binsize = int(g.attrs['bin-size'])
        self.chrom_offset = np.asarray(g['indexes/chrom_offset'][:], dtype=np.int64)
        self.bin1_offset = np.asarray(g['indexes/bin1_offset'][:], dtype=np.int64)
        self.weights = {k: np.ascontiguousarray(g['bins'][k][:], dtype=np.float64)
                        for k in g['bins'].keys() if k not in ('chrom', 'start', 'end')}
        self.bin1_id, self.bin2_id, self.count = g['pixels/bin1_id'], g['pixels/bin2_id'], g['pixels/count']
        for d in (self.bin1_id, self.bin2_id, self.count):     # (the built-in reader indexes a column's chunks on first use:
            if hasattr(d, '_chunk_index') and getattr(d, 'chunks', None):      #  done here, before reader threads exist)
                d._chunk_index()
        self.chunk = int(chunk)
        self.max_read = 0            # largest single read of a pixel column, in pixels (tests)
        self.threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)))
        self._pool = None            # inflate workers (created with the first piece read, under _lazy_lock: the read-ahead
        self._readers = None         #  thread and the caller's thread may both arrive here first)
        import threading
        self._lazy_lock = threading.Lock()
        self.direct_reads = 0        # HDF5 chunks decoded by _ChunkReader rather than by HDF5's own filter pipeline (tests)
        self._ahead = {}             # chrom -> (thread, result box)
        self._nonneg = None
        if len(self.bin1_offset) != int(self.chrom_offset[-1]) + 1:
            raise ValueError('indexes/bin1_offset must have one entry per bin plus one')

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        self._h5.close()

    chrom_index = PixelTable.chrom_index
    chrom_bins = PixelTable.chrom_bins
    weight = PixelTable.weight

    def rows_slice(self, g0, g1):
        n = len(self.bin1_offset) - 1
        return int(self.bin1_offset[min(max(int(g0), 0), n)]), int(self.bin1_offset[min(max(int(g1), 0), n)])

    def count_nonnegative(self):
        """no negative count anywhere (decides whether a zero row sum means an empty row): one pass over pixels/count
        in pieces, cached."""
        if self._nonneg is None:
            ok, n = True, self.count.shape[0]
            for a in range(0, n, self.chunk):
                if np.asarray(self.count[a:min(a + self.chunk, n)]).min(initial=0) < 0:
                    ok = False
                    break
            self._nonneg = ok
        return self._nonneg

    def _read_cis(self, chrom):
        """One chromosome's cis pixels.  The three columns are filled piece by piece into arrays allocated once at the row
        count the index gives (cis + trans rows of the chromosome's bins; trimmed views are returned), so the reader holds
        one set of columns per chromosome -- with the read-ahead, two chromosomes' -- and never a second copy."""
        lo, hi = self.chrom_bins(chrom)
        a, b = self.rows_slice(lo, hi)
        n = 0
        # (bin2_id is narrowed to 32 bits while the pieces are copied: half the host memory of the column and half its PCIe bytes)
        o1 = np.empty(b - a, np.int64); o2 = np.empty(b - a, np.int32 if int(self.chrom_offset[-1]) < 2**31 else np.int64)
        cdt = np.dtype(self.count.dtype)                     # (wider integer columns keep their type: _counts decides)
        oc = np.empty(b - a, np.float64 if cdt.kind == 'f' else (np.int32 if cdt.itemsize <= 4 and cdt.kind == 'i' else cdt))
        for x in range(a, b, self.chunk):
            y = min(x + self.chunk, b)
            self.max_read = max(self.max_read, y - x)
            p1, p2, pc = self._read_piece(x, y)
            keep = p2 < hi                                   # cis pixels of this piece (pixels are sorted by bin1 only)
            if keep.all():
                m = y - x
                o2[n:n + m] = p2; o1[n:n + m] = p1; oc[n:n + m] = pc
            else:
                m = int(keep.sum())
                o2[n:n + m] = p2[keep]; o1[n:n + m] = p1[keep]; oc[n:n + m] = pc[keep]
            n += m
        return o1[:n], o2[:n], _counts(oc[:n]), lo, hi - lo

    def chrom_offsets(self, chrom):
        """None: the cis columns of a file are filtered copies (trans pixels dropped), the stored index does not describe
        them; PixelSelector derives the index of what it got from the sorted bin1 column."""
        return None

    def _read_piece(self, x, y):
        """Rows [x, y) of the three pixel columns.  HDF5 inflates one chunk at a time under the library's global lock
        (10 Mpixel / s on one core for gzip + shuffle columns, whatever the number of reading threads), so for the
        usual cooler layout -- chunked 1-d columns, deflate with or without byte shuffle -- the COMPRESSED chunks are
        fetched as they are (`read_direct_chunk`) and inflated / unshuffled on `self.threads` host threads (zlib
        releases the GIL).  Any other filter pipeline, or a chunk the writer stored unfiltered, goes through h5py's
        ordinary read."""
        cols = (self.bin1_id, self.bin2_id, self.count)
        with self._lazy_lock:
            if self._readers is None:
                self._readers = [_ChunkReader.of(d) for d in cols]
            readers = self._readers
            if self.threads >= 2 and self._pool is None and not any(r is None for r in readers):
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(self.threads)
        if self.threads < 2 or any(r is None for r in readers):
            return tuple(np.asarray(d[x:y]) for d in cols)
        outs = [np.empty(y - x, d.dtype) for d in cols]
        jobs = []
        for r, o in zip(readers, outs):
            for c0 in range(x - x % r.clen, y, r.clen):
                jobs.append(self._pool.submit(r.into, c0, x, y, o))
        n = sum(j.result() for j in jobs)
        with self._lazy_lock:
            self.direct_reads += n
        return tuple(outs)

    def prefetch(self, chrom):
        """Start reading the chromosome's cis pixels on a host thread (at most one read ahead is kept)."""
        import threading
        chrom = str(chrom)
        if chrom in self._ahead or chrom not in self.chromnames:
            return
        for k in list(self._ahead):                          # an unused older read-ahead is dropped
            self._ahead.pop(k)[0].join()
        box = {}

        def work():
            try:
                box['r'] = self._read_cis(chrom)
            except BaseException as e:      # noqa: BLE001 -- re-raised by chrom_pixels on the caller's thread
                box['e'] = e
        th = threading.Thread(target=work, daemon=True)
        th.start()
        self._ahead[chrom] = (th, box)

    def chrom_pixels(self, chrom):
        chrom = str(chrom)
        if chrom in self._ahead:
            th, box = self._ahead.pop(chrom)
            th.join()
            if 'e' in box:
                raise box['e']
            return box['r']
        return self._read_cis(chrom)


def pixel_values(count, weight, bin1, bin2):
    """cooler's balanced value of each stored pixel: count * (b[bin1] * b[bin2]) (`arr * np.outer(bias1, bias2)`),
    b being the multiplicative bias (PixelTable.weight); raw counts when weight is None."""
    v = count.astype(np.float64)
    if weight is not None:
        v = v * (weight[bin1] * weight[bin2])
    return v


class PixelSelector:
    """``cooler.Cooler(...).matrix(balance=...)`` over a PixelTable: ``fetch(region[, region2])`` with cooler's
    extent rule (0-based half-open bp intervals; bins lo = start // binsize, hi = ceil(end / binsize))."""

    def __init__(self, table, balance=True, divisive=None):
        self.table = table
        self.balance = balance
        self.w = table.weight(balance, divisive)         # multiplicative bias (1 / w for KR, VC, SQRT_VC)
        self.resol = table.binsize
        self.nfetch = 0
        self._nonneg = None          # no negative count / weight anywhere: row sums are zero iff all pixels are
        self._cheap = None
        self._wpos = None

    def _extent(self, region):
        region = str(region)
        t = self.table
        if ':' not in region:
            lo, hi = t.chrom_bins(region)
            return region, 0, hi - lo
        name, rng = region.rsplit(':', 1)
        s, e = rng.replace(',', '').split('-')
        s, e = int(s), int(e)
        lo, hi = t.chrom_bins(name)
        if s < 0 or e > (hi - lo) * self.resol or s > e:
            raise ValueError('Genomic region out of bounds: %s' % region)
        return name, s // self.resol, -(-e // self.resol)

    def chrom_pixels(self, chrom):
        b1, b2, cn, lo, n = self.table.chrom_pixels(chrom)
        px = dict(bin1=b1, bin2=b2, count=cn, weight=self.w, lo=lo, nrows=n)
        off = self.table.chrom_offsets(chrom) if hasattr(self.table, 'chrom_offsets') else None
        if off is None and len(b1):      # (filtered columns: the index of what is left, one search per bin in the sorted column)
            off = np.searchsorted(b1, np.arange(lo, lo + n + 1), side='left').astype(np.int64)
        if off is not None:
            px['off'] = off              # the CSR form of bin1 (stp_band_pack_csr takes it instead of the column)
        return px

    def prefetch(self, chrom):
        """Hint: `chrom_pixels(chrom)` comes next (a lazily read table starts the read on a host thread)."""
        self.table.prefetch(chrom)

    def fetch(self, region, region2=None):
        self.nfetch += 1
        n1, r0, r1 = self._extent(region)
        n2, c0, c1 = (n1, r0, r1) if region2 is None else self._extent(region2)
        if n1 != n2:
            raise ValueError('trans fetch is not on the stripenn path')
        lo, _ = self.table.chrom_bins(n1)
        return self._dense(r0 + lo, r1 + lo, c0 + lo, c1 + lo)

    def nonnegative(self):
        """No negative count / bias anywhere: a row's sum is zero iff it holds no positive pixel."""
        if self._nonneg is None:
            t = self.table
            with np.errstate(invalid='ignore'):
                self._nonneg = bool(t.count_nonnegative() and (self.w is None or not np.any(self.w < 0)))
        return self._nonneg

    def _positive(self, a, b):
        """value > 0 for the stored pixels [a, b) (NaN from a masked bin compares False), formed for the slice a
        query needs: no whole-table array ever exists on the host (a lazily read .cool table has none to offer)."""
        t = self.table
        w = self.w
        cnt = np.asarray(t.count[a:b])
        if w is None:
            return cnt > 0
        if self._cheap is None:
            # (count * w1) * w2 > 0  <=>  count > 0 and w1 > 0 and w2 > 0, unless the product underflows to 0:
            # impossible while every positive weight is >= 1e-100 (counts are >= 1); otherwise form the products
            self._cheap = not np.any((w > 0) & (w < 1e-100))
            self._wpos = w > 0                                       # NaN (masked bin) -> False
        b1, b2 = np.asarray(t.bin1_id[a:b]), np.asarray(t.bin2_id[a:b])
        if self._cheap:
            return (cnt > 0) & self._wpos[b1] & self._wpos[b2]
        return pixel_values(cnt, w, b1, b2) > 0

    def row_nonzero(self, region, region2=None):
        """Which rows of `fetch(region, region2)` have a non-zero sum after NaN -> 0 (what nulldist's pools
        ask of the unit matrix, getStripe.py:329-331), straight from the pixel table.  Returns None when the
        table holds negative values (then only the dense sum can tell)."""
        n1, r0, r1 = self._extent(region)
        n2, c0, c1 = (n1, r0, r1) if region2 is None else self._extent(region2)
        if n1 != n2:
            raise ValueError('trans fetch is not on the stripenn path')
        t = self.table
        if not self.nonnegative():
            return None
        lo, _ = t.chrom_bins(n1)
        R0, R1, C0, C1 = r0 + lo, r1 + lo, c0 + lo, c1 + lo
        hit = np.zeros(r1 - r0, dtype=bool)
        a, b = t.rows_slice(R0, R1)                 # stored pixels: bin1 in the rows, bin2 in the columns
        b2 = np.asarray(t.bin2_id[a:b])
        ok = self._positive(a, b) & (b2 < C1)
        if C0 > R0:                                  # (bin2 >= bin1 >= R0 covers the lower bound otherwise)
            ok &= b2 >= C0
        hit[np.asarray(t.bin1_id[a:b])[ok] - R0] = True
        a, b = t.rows_slice(C0, min(C1, R1))        # mirror images: bin2 in the rows, bin1 (<= bin2 < R1) in the columns
        b2 = np.asarray(t.bin2_id[a:b])
        ok = self._positive(a, b) & (b2 >= R0) & (b2 < R1)
        hit[b2[ok] - R0] = True
        return hit

    def __getitem__(self, key):
        """cooler's `matrix[r0:r1, c0:c1]` with GLOBAL bin indices (getStripe.py:107-158, the `-s` quantile)."""
        rs, cs = key
        n = int(self.table.chrom_offset[-1])
        r0, r1, _ = rs.indices(n)
        c0, c1, _ = cs.indices(n)
        return self._dense(r0, max(r1, r0), c0, max(c1, c0))

    def _dense(self, R0, R1, C0, C1):
        """Dense block of global bins rows [R0, R1) x cols [C0, C1)."""
        t = self.table
        out = np.zeros((R1 - R0, C1 - C0), dtype=np.float64)
        if self.w is not None:
            # cooler multiplies the dense count block by np.outer(bias1, bias2): a cell without a stored pixel is
            # 0 * (b1 * b2) -- NaN along the whole row / column of an unbalanced (NaN-weight) bin, 0 elsewhere
            with np.errstate(invalid='ignore', over='ignore'):
                out = 0.0 * (self.w[R0:R1][:, None] * self.w[C0:C1][None, :])
        # stored pixels (bin1 in rows, bin2 in cols), then their mirror images (bin2 in rows, bin1 in cols)
        for (A0, A1, B0, B1, mirror) in ((R0, R1, C0, C1, False), (C0, C1, R0, R1, True)):
            a, b = t.rows_slice(A0, A1)
            b1 = np.asarray(t.bin1_id[a:b]); b2 = np.asarray(t.bin2_id[a:b])
            keep = (b2 >= B0) & (b2 < B1)
            b1, b2 = b1[keep], b2[keep]
            v = pixel_values(np.asarray(t.count[a:b])[keep], self.w, b1, b2)
            if mirror:
                out[b2 - R0, b1 - C0] = v
            else:
                out[b1 - R0, b2 - C0] = v
        return out
