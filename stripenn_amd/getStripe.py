"""Drop-in counterpart of the reference's stripe engine (src/stripenn/getStripe.py:17-1232).

Same class name, constructor signature, method names, argument meaning and return shapes as the
reference, so `stripenn.compute` / `score.getScore`-style drivers run unchanged; underneath, every
arithmetic loop of the hot path runs in hand-written HIP kernels on an MI355X through the C ABI of
libstripenn_hip.so (include/stripenn_hip.h).  What stays on the host, as in the north star:
cooler-style `fetch` I/O, Python's `random.Random` sampling, pandas table assembly, RemoveRedundant.

Data flow: each chromosome is fetched ONCE into a dense diagonal band (halfwidth 512 bins) that
stays resident in HBM; frames, background windows, p-values and Stripiness all read that band
instead of re-fetching 400x400 dense blocks per frame / per stripe / per maxpixel as the
reference does (getStripe.py:808 inside stripenn.py:134, :560, :684-696).
"""
import math
import random
import time

import numpy as np
import pandas as pd

from .backend import (HipBackend, NULL_SAMPLE_DTYPE, PV_STRIPE_DTYPE, RECT_DTYPE, SCORE_STRIPE_DTYPE)

HALFWIDTH = 512
EXTRACT_COLUMNS = ['chr', 'pos1', 'pos2', 'chr2', 'pos3', 'pos4', 'length', 'width', 'total', 'Mean', 'maxpixel', 'num',
                   'start', 'end', 'x', 'y', 'h', 'w', 'medpixel']


def nantozero(nparray):
    """getStripe.py:1229-1232"""
    nparray[np.isnan(nparray)] = 0
    return nparray


def _extent(start, end, resol):
    """cooler's bin extent of a 'chr:start-end' region (0-based half-open bp): [lo, hi)."""
    start, end = int(start), int(end)
    return start // resol, -(-end // resol)


def quantile_linear(order_stats, n, q):
    """numpy.quantile(a, q) (method 'linear', numpy/lib/function_base.py `_quantile`) given a callable
    that returns exact order statistics a_sorted[ranks]: virtual index (n-1)*q, neighbours floor / +1
    (both the last element once the index reaches n-1, the first below 0), gamma = index - floor, and
    numpy's `_lerp` (a + (b-a)*t, replaced by b - (b-a)*(1-t) where t >= 0.5)."""
    q = np.asanyarray(q, dtype=np.float64)
    scalar = (q.ndim == 0)
    qa = np.atleast_1d(q)
    if n <= 0:
        raise IndexError('index -1 is out of bounds for axis 0 with size 0')      # np.quantile of an empty array
    virtual = (n - 1) * qa
    prev = np.floor(virtual).astype(np.intp)
    nxt = prev + 1
    above = virtual >= n - 1
    prev[above] = -1
    nxt[above] = -1
    below = virtual < 0
    prev[below] = 0
    nxt[below] = 0
    gamma = np.asanyarray(virtual - prev)
    ranks = np.unique(np.concatenate((prev % n, nxt % n)))
    vals = dict(zip(ranks.tolist(), np.asarray(order_stats(ranks), dtype=np.float64).tolist()))
    a = np.array([vals[int(i % n)] for i in prev], dtype=np.float64)
    b = np.array([vals[int(i % n)] for i in nxt], dtype=np.float64)
    diff = np.subtract(b, a)
    lerp = np.asanyarray(np.add(a, diff * gamma))
    np.subtract(b, diff * (1 - gamma), out=lerp, where=gamma >= 0.5)
    return lerp[0] if scalar else lerp


class getStripe:
    def __init__(self, unbalLib, resol, minH, maxW, canny, all_chromnames, chromnames, all_chromsizes, chromsizes, core,
                 bfilter, seed, backend=None, device=0, halfwidth=None, frame_span=None):
        """Reference signature (getStripe.py:18) + optional keywords (not in the reference): backend / device;
        halfwidth of the resident band (default: 512, or more when 448 + 2 * (50000 / resol) needs it, i.e. below
        1 kb); frame_span = {chromosome: (first frame, one past the last)} restricts extract() to those frames
        of a chromosome (multi-GPU driver, stripenn_amd/shard.py) -- only the band rows they need are built."""
        self.unbalLib = unbalLib
        self.resol = int(resol)
        self.minH = minH
        self.maxW = maxW
        self.canny = canny
        self.all_chromnames = list(all_chromnames)
        self.all_chromsizes = np.asarray(all_chromsizes)
        self.chromnames = list(chromnames)
        self.chromsizes = np.asarray(chromsizes)
        self.core = core
        self.bfilter = bfilter
        self.seed = seed
        self.prng = random.Random(seed)
        self.chromnames2sizes = {}
        for i in range(len(self.all_chromnames)):
            self.chromnames2sizes[self.all_chromnames[i]] = self.all_chromsizes[i]
        if halfwidth is None:
            need = 448 + 2 * int(50000 / self.resol)              # frame 400 + flank + windows (include/stripenn_hip.h)
            halfwidth = max(HALFWIDTH, -(-need // 64) * 64)
        self.halfwidth = int(halfwidth)
        self.frame_span = {str(k): (int(v[0]), int(v[1])) for k, v in (frame_span or {}).items()}
        self.backend = backend if backend is not None else HipBackend(device)   # raises without a GPU
        self._bands = {}
        self._partial = set()           # chromosomes whose resident band holds only the rows of their frame span
        self._near = {}
        self._frames = {}
        self._search_cache = {}
        self.timing = {}
        # eager_search (set by the single-process `compute` driver): the StripeSearch of a chromosome is enqueued as soon
        # as its maxpixel quantiles are known, so the device searches chromosome i while the host uploads and packs the
        # pixel columns of chromosome i + 1 (the packer and the select run on a stream of their own)
        self.eager_search = False

    # ------------------------------------------------------------------ band management
    def _nbins(self, chrom):
        return int(math.ceil(int(self.chromnames2sizes[str(chrom)]) / self.resol))

    def _band(self, chrom, whole=True, select=None):
        """Resident band of one chromosome; built from row strips fetched through the selector, or packed on the
        device from the pixel table.  whole=False (extract and the per-stripe kernels) accepts a band that holds
        only the rows of the chromosome's frame span; the whole-chromosome steps (expected values, background)
        ask for all rows."""
        chrom = str(chrom)
        if chrom in self._bands and not (whole and chrom in self._partial):
            return self._bands[chrom]
        if chrom in self._bands:                                   # a partial band has to become a whole one
            self.release(chrom)
        t0 = time.time()
        nb = self._nbins(chrom)
        size = int(self.chromnames2sizes[chrom])
        hw = self.halfwidth
        # rows the frames of this chromosome's span can touch (+ one halfwidth of margin), or all of them
        ra, rb = 0, nb
        if not whole and chrom in self.frame_span:
            fa, fb = self.frame_span[chrom]
            ra, rb = max(0, fa * 200 - 100 - hw), min(nb, fb * 200 + 100 + hw)
            if (ra, rb) != (0, nb):
                self._partial.add(chrom)
        if hasattr(self.unbalLib, 'chrom_pixels') and hasattr(self.backend, 'pack_chrom'):
            # the source IS cooler's pixel table: the device builds the band from it, no dense fetch at all
            px = self.unbalLib.chrom_pixels(chrom)
            if (ra, rb) != (0, nb):                                # stored pixels (i <= j) that land in rows [ra, rb)
                a, b = np.searchsorted(px['bin1'], [px['lo'] + ra - hw, px['lo'] + rb], side='left')
                px = dict(px, bin1=px['bin1'][a:b], bin2=px['bin2'][a:b], count=px['count'][a:b])
                px.pop('off', None)                                # (the CSR index describes the whole chromosome's pixels)
                select = None                                      # (a partial band does not see every pixel)
            self._bands[chrom] = self.backend.pack_chrom(px, hw, select) if select is not None else self.backend.pack_chrom(px, hw)
            self.timing['band_build_s'] = self.timing.get('band_build_s', 0.0) + time.time() - t0
            return self._bands[chrom]
        band = np.zeros((nb, 2 * hw), dtype=np.float64)
        strip = 2048
        dd = np.arange(-hw, hw)[None, :]
        for r0 in range(ra, rb, strip):
            r1 = min(r0 + strip, rb)
            c0, c1 = max(r0 - hw, 0), min(r1 + hw, nb)
            rows = '%s:%d-%d' % (chrom, r0 * self.resol + 1, min(r1 * self.resol, size))
            cols = '%s:%d-%d' % (chrom, c0 * self.resol + 1, min(c1 * self.resol, size))
            blk = np.asarray(self.unbalLib.fetch(rows, cols), dtype=np.float64)
            rr = np.arange(r0, r1)[:, None]
            cc = rr + dd
            ok = (cc >= 0) & (cc < nb)
            sub = blk[rr - r0, np.clip(cc - c0, 0, c1 - c0 - 1)]
            band[r0:r1] = np.where(ok, sub, 0.0)
        self._bands[chrom] = self.backend.open_chrom(band)
        self.timing['band_build_s'] = self.timing.get('band_build_s', 0.0) + time.time() - t0
        return self._bands[chrom]

    def release(self, chrom=None):
        """Free device memory of one / all chromosomes (not in the reference)."""
        for c in ([str(chrom)] if chrom is not None else list(self._bands)):
            for key in [k for k in self._search_cache if k[0] == c]:      # a search still in flight reads these buffers
                pend = self._search_cache[key]
                if hasattr(pend, 'wait'):
                    self._search_cache[key] = pend.wait()
            fr = self._frames.pop(c, None)
            if fr is not None and hasattr(fr[0], 'close'):
                fr[0].close()
            self._partial.discard(c)
            self._near.pop(c, None)
            b = self._bands.pop(c, None)
            if b is not None:
                self.backend.close_chrom(b)

    # ------------------------------------------------------------------ quantiles (host)
    def getQuantile_original(self, coolinfo, ChrList, quantile):
        """getStripe.py:160-176: `np.quantile(mat[mat > 0], quantile)` over the whole chromosome.
        The chromosome is streamed to the device in row strips (the reference materialises it densely:
        12 GB for chr1 at 5 kb); the exact order statistics come from a radix select on the GPU
        (stp_select_*), numpy's 'linear' interpolation between them is applied here."""
        res = {}
        chrom_names = list(coolinfo.chromsizes.keys())
        chridx = sorted(c for c in range(len(chrom_names)) if chrom_names[c] in ChrList)
        for pos, ci in enumerate(chridx):
            CHROM = chrom_names[ci]
            nb = self._nbins(CHROM)
            size = int(self.chromnames2sizes[str(CHROM)])
            if hasattr(self.unbalLib, 'prefetch') and pos + 1 < len(chridx):
                # a table read lazily from a .cool file fetches the NEXT chromosome's pixel columns on a host thread
                # while this one is packed, selected and (later) searched
                nxt = str(chrom_names[chridx[pos + 1]])
                if nxt not in self._bands:
                    self.unbalLib.prefetch(nxt)
            sel = self.backend.select_open()
            try:
                if hasattr(self.unbalLib, 'chrom_pixels'):
                    # pixel-table source: the dense symmetric matrix holds every off-diagonal pixel twice.  When the
                    # chromosome's band is still to be built, one pass over the table columns (one trip over PCIe)
                    # packs the band AND appends the values to the select.
                    if str(CHROM) not in self._bands and hasattr(self.backend, 'pack_chrom'):
                        self._band(CHROM, whole=True, select=sel)
                    else:
                        px = self.unbalLib.chrom_pixels(CHROM)
                        self.backend.select_append_pixels(sel, px['bin1'], px['bin2'], px['count'], px['weight'])
                    nb = 0
                strip = max(1, int(16e6 // max(nb, 1)))                 # <= 128 MB of float64 per fetch
                for r0 in range(0, nb, strip):
                    r1 = min(r0 + strip, nb)
                    rows = '%s:%d-%d' % (CHROM, r0 * self.resol + 1, min(r1 * self.resol, size))
                    blk = np.asarray(self.unbalLib.fetch(rows, str(CHROM)), dtype=np.float64)
                    self.backend.select_append(sel, blk[blk > 0])
                n = self.backend.select_count(sel)
                res[CHROM] = quantile_linear(lambda ranks: self.backend.select_ranks(sel, ranks), n, quantile)
            finally:
                self.backend.select_close(sel)
            if self.eager_search and str(CHROM) in self._bands and hasattr(self.backend, 'stripe_search_begin'):
                mine = [str(c) for c in self.chromnames]
                if str(CHROM) in mine:
                    k = mine.index(str(CHROM))
                    self._search_begin(self.chromnames[k], k, res[CHROM])
        return res

    def getQuantile_slow(self, coolinfo, ChrList, quantile):
        """getStripe.py:107-158 (row strips through the selector's __getitem__)."""
        res = {}
        chrom_size = coolinfo.chromsizes
        chrom_cum = np.nancumsum(chrom_size)
        chrom_names = list(chrom_size.keys())
        nbin = coolinfo.binsize
        chridx = sorted(c for c in range(len(chrom_names)) if chrom_names[c] in ChrList)
        for ci in chridx:
            CHROM = chrom_names[ci]
            CHROMSIZE = chrom_size.iloc[ci] if hasattr(chrom_size, 'iloc') else chrom_size[ci]
            L = int(np.ceil(CHROMSIZE / nbin))
            r_start = 0 if ci == 0 else chrom_cum[ci - 1]
            r_end = chrom_cum[ci]
            r_start = int(np.ceil(r_start / nbin) + 1)
            r_end = int(np.ceil(r_end / nbin))
            w = int(np.floor(25000000 / L))
            parts = []
            for k in range(int(np.ceil(L / w))):
                w_start = k * w + r_start
                w_end = (k + 1) * w - 1
                if w_end >= np.floor(CHROMSIZE / nbin):
                    w_end = int(np.floor(CHROMSIZE / nbin))
                w_end = int(w_end + r_start)
                blk = self.unbalLib[int(w_start):w_end, r_start:r_end]
                parts.append(blk[blk > 0])
            hs = np.concatenate([np.empty(0)] + parts)
            res[CHROM] = np.quantile(hs, quantile)
        return res

    # ------------------------------------------------------------------ expected values
    def mpmean(self):
        """getStripe.py:178-235: mean contact per diagonal 0..399 (K: k_diag_sums)."""
        meantable = {}
        for chrom in self.chromnames:
            band = self._band(chrom)
            ps, pc = self.backend.diag_sums(band)
            means = []
            for j in range(400):
                pixelsum = sum(ps[:, j].tolist())        # frames added in order like :227-230
                countsum = int(pc[:, j].sum())
                means.append(pixelsum / countsum)
            meantable[chrom] = means
        return meantable

    # ------------------------------------------------------------------ background distribution
    def _unit_geometry(self, chrom):
        chrsize = int(self.chromnames2sizes[chrom])
        itera = int(min(chrsize / self.resol / 500, 25))
        unitsize = int(np.floor(chrsize / self.resol / itera)) if itera > 0 else 0
        return chrsize, itera, unitsize

    def _unit_regions(self, chrom, it, chrsize, unitsize):
        """getStripe.py:313-325 -> bin extents (row0,row1,col0,col1) and region strings."""
        resol = self.resol
        start1 = int(unitsize * resol * it + 1)
        start0 = start1 - (400 * resol)
        end1 = int(unitsize * resol * (it + 1))
        end2 = int(unitsize * resol * (it + 1) + (400 * resol))
        if end2 > chrsize:
            end2 = chrsize - 1
        if end1 > chrsize - 400 * resol:
            end1 = chrsize - 400 * resol
        if start0 <= 1:
            start0 = 1
        start0 = int(start0)
        p1 = '%s:%d-%d' % (chrom, start1, end1)
        p2 = '%s:%d-%d' % (chrom, start0, end2)
        r0, r1 = _extent(start1, end1, resol)
        c0, c1 = _extent(start0, end2, resol)
        return p1, p2, r0, r1, c0, c1

    def nulldist(self):
        """getStripe.py:238-499.  Sample-size arithmetic, pools and random.Random draws on the host in
        the reference's order; the 2.4 M window means in k_null_windows.  PRNG rule: numcores == 1
        keeps one stream across chromosomes (joblib runs in-process); numcores > 1 restarts from the
        seed for every chromosome (loky pickles `self`), like the reference.
        The three phases below are also called one by one by the multi-GPU driver (shard.py)."""
        t0 = time.time()
        chromnames2 = self.null_candidates()
        n_available_col = [self.null_available_cols(c) for c in chromnames2]
        chromnames2, samplesize = self.null_samplesizes(chromnames2, n_available_col)
        parts = [self.null_tables(c, chromnames2, samplesize) for c in chromnames2]
        out = self.null_concat(parts)
        self.timing['nulldist_s'] = time.time() - t0
        return out

    def null_candidates(self):
        """:242-247 chromosomes whose size share gives at least one of the 1000 samples"""
        with np.errstate(divide='ignore', invalid='ignore'):
            samplesize = (self.all_chromsizes / np.sum(self.all_chromsizes)) * 1000
            samplesize = np.uint64(samplesize)
            notzero = np.where(samplesize != 0)
            return [self.all_chromnames[i] for i in notzero[0]]

    def null_available_cols(self, chrom):
        """:249-273 number of non-empty rows over the chromosome's units"""
        resol = self.resol
        chrom = str(chrom)
        chrsize, itera, unitsize = self._unit_geometry(chrom)
        poolsum = 0
        for it in range(itera):
            a = int(unitsize * resol * it + 1)
            b = int(unitsize * resol * (it + 1))
            if a > b:
                a, b = b, a
            pos = '%s:%d-%d' % (chrom, a, b)
            live = self._rows_nonzero(pos, pos, chrom)[0]
            poolsum += int(np.count_nonzero(live))
        return poolsum

    def _nearest(self, chrom):
        """(right, left) nearest-positive-pixel distances of a chromosome whose band was packed on the device from
        ALL its cis pixels (stp_band_nearest), or None: dense-fetch sources, partial bands, tables with negative
        values (there a zero row sum does not mean an empty row)."""
        if chrom in self._near:
            return self._near[chrom]
        near = None
        ok = getattr(self.unbalLib, 'nonnegative', None)
        if ok is not None and ok() and hasattr(self.backend, 'band_nearest'):
            band = self._band(chrom)
            if chrom not in self._partial:
                near = self.backend.band_nearest(band)
        self._near[chrom] = near
        return near

    def _rows_nonzero(self, p1, p2, chrom=None):
        """(rows whose sum over the fetched block is non-zero after NaN -> 0, the block or None).  Answered without
        a dense block when possible: from the nearest-positive-pixel table the device filled while packing the band
        (row i of rows x [c0, c1), c0 <= rows < c1, is non-empty iff i + right[i] < c1 or i - left[i] >= c0), else
        from the selector's own tables (PixelSelector.row_nonzero); the block is then fetched only by the caller
        that really needs its pixels."""
        if chrom is not None:
            near = self._nearest(chrom)
            if near is not None:
                (r0, r1), (c0, c1) = self._region_bins(p1), self._region_bins(p2)
                if c0 <= r0 and c1 >= r1:
                    i = np.arange(r0, r1, dtype=np.int64)
                    return (i + near[0][r0:r1] < c1) | (i - near[1][r0:r1] >= c0), None
        fast = getattr(self.unbalLib, 'row_nonzero', None)
        if fast is not None:
            live = fast(p1, p2)
            if live is not None:
                return live, None
        mat = nantozero(np.array(self.unbalLib.fetch(p1, p2), dtype=np.float64))
        return np.sum(mat, axis=1) != 0, mat

    def _region_bins(self, region):
        name, rng = str(region).rsplit(':', 1)
        a, b = rng.split('-')
        return _extent(int(a), int(b), self.resol)

    def null_samplesizes(self, chromnames2, n_available_col):
        """:278-283 (the sample-size array keeps its unfiltered indexing, as in the reference)"""
        with np.errstate(divide='ignore', invalid='ignore'):
            samplesize = (n_available_col / np.sum(n_available_col)) * 1000
            samplesize = np.uint64(samplesize)
            dif = 1000 - int(np.sum(samplesize))
            notzero = np.where(samplesize != 0)
            chromnames2 = [chromnames2[i] for i in notzero[0]]
            samplesize[0] = np.uint64(int(samplesize[0]) + dif)
        return chromnames2, samplesize

    def null_pools(self, chrom):
        """The sampling pools of one chromosome's units (:329-335): per unit the rows with a non-zero sum that keep
        the 20-row margin (`base`, what the top-up branch draws from, :436-438) and the pool of the unit loop
        (`pool`: unit 0 also drops x <= 410).  Plain lists: they travel between ranks in the multi-GPU driver."""
        chrom = str(chrom)
        chrsize, itera, unitsize = self._unit_geometry(chrom)
        out = []
        for it in range(itera):
            p1, p2, r0, r1, c0, c1 = self._unit_regions(chrom, it, chrsize, unitsize)
            live = self._rows_nonzero(p1, p2, chrom)[0]
            x = np.nonzero(live)[0]
            base = x[(x > 20) & (x < (unitsize - 20))]
            pool = base[(base > 410) & (base < (c1 - c0))] if it == 0 else base
            out.append((base.tolist(), pool.tolist()))
        return out

    def null_tables(self, chrom, chromnames2, samplesize, pools=None, windows=True):
        """main_null_calc (:285-479) for one chromosome -> four 400 x ss tables.  `pools` (null_pools) may come
        from another rank; windows=False only draws the samples (the PRNG stream of numcores == 1 runs on across
        chromosomes, so every rank replays the draws of the chromosomes it does not own) and returns None."""
        resol = self.resol
        bs = int(50000 / resol)
        tabs = [[], [], [], []]
        with np.errstate(divide='ignore', invalid='ignore'):
            chrom = str(chrom)
            prng = self.prng if self.core == 1 else random.Random(self.seed)
            c = chromnames2.index(chrom)
            ss = samplesize[c]                                     # (index into the unfiltered array, :295-298)
            chrsize, itera, unitsize = self._unit_geometry(chrom)
            if pools is None:
                pools = self.null_pools(chrom)
            band = self._band(chrom) if windows else None
            self._null_pending = []
            n_pool = []
            collected = 0
            sss = int(ss / itera)
            last_it = -1
            for it in range(itera):
                last_it = it
                pool = pools[it][1]
                n_pool.append(len(pool))
                if len(pool) == 0:
                    continue
                k = len(pool) if len(pool) < sss else sss
                randval = prng.choices(pool, k=k)
                collected += len(randval)
                if windows:
                    p1, p2, r0, r1, c0, c1 = self._unit_regions(chrom, it, chrsize, unitsize)
                    self._null_batch(band, tabs, randval, r0, r1, c0, c1, 400 if it > 0 else 0, bs, None, (p1, p2))
            depl = int(ss) - collected                             # :416-477
            if depl > 0:
                rich = int(np.argmax(n_pool))
                randval = prng.choices(pools[rich][0], k=depl)
                if windows:
                    p1, p2, r0, r1, c0, c1 = self._unit_regions(chrom, rich, chrsize, unitsize)
                    # the reference tests the loop variable `it` left over from the unit loop (:458)
                    self._null_batch(band, tabs, randval, r0, r1, c0, c1, 400 if last_it > 0 else 0, bs, None, (p1, p2))
            if not windows:
                return None
            self._null_flush(band, tabs, bs)
        return [np.column_stack([np.zeros((400, 0))] + t) for t in tabs]

    @staticmethod
    def null_concat(parts):
        """:483-496"""
        out = [np.column_stack([np.zeros((400, 0))] + [p[k] for p in parts]) for k in range(4)]
        return out[0], out[1], out[2], out[3]

    def _null_batch(self, band, tabs, randval, r0, r1, c0, c1, yoff, bs, mat, regions=None):
        """One batch of sampled rows.  The resident band serves every window unless a Python slice
        of the reference wraps around (negative start) or leaves the band; then the unit matrix the
        host has just fetched for the pool is handed to the kernel instead."""
        if len(randval) == 0:
            return 0
        up = bs // 2
        xmin, xmax = min(randval), max(randval)
        wraps = (xmin - up - bs < 0) or (xmin + yoff - 399 - up < 0)
        reach = 399 + up + (bs - up) + bs + abs((c0 + yoff) - r0)      # farthest |col - row| a window touches
        unit = None
        if wraps or reach >= self.halfwidth:
            if mat is None:                      # the pool came from the selector's own tables: fetch the block now
                mat = nantozero(np.array(self.unbalLib.fetch(*regions), dtype=np.float64))
            unit = mat
        samples = np.zeros(len(randval), dtype=NULL_SAMPLE_DTYPE)
        samples['row0'], samples['nrow'] = r0, r1 - r0
        samples['col0'], samples['ncol'] = c0, c1 - c0
        samples['x'] = np.asarray(randval, dtype=np.int32)
        samples['yoff'] = yoff
        if unit is None:
            # served by the resident band: nothing on the host depends on the result, so the batch only takes
            # its place in the tables now and all such batches of the chromosome go to the device in one launch
            for t in tabs:
                t.append(None)
            self._null_pending.append((len(tabs[0]) - 1, samples))
            return len(randval)
        lu, ru, ld, rd = self.backend.null_windows(band, samples, bs, unit)
        for t, a in zip(tabs, (lu, ru, ld, rd)):
            t.append(a)
        return len(randval)

    def _null_flush(self, band, tabs, bs):
        """Run the deferred band-served batches of one chromosome as one device call and drop the results
        into their places (column order = batch order, as the reference appends them)."""
        pend, self._null_pending = self._null_pending, []
        if not pend:
            return
        allsamp = np.concatenate([p[1] for p in pend])
        res = self.backend.null_windows(band, allsamp, bs, None)
        o = 0
        for pos, samp in pend:
            n = len(samp)
            for t, a in zip(tabs, res):
                t[pos] = a[:, o:o + n]
            o += n

    # ------------------------------------------------------------------ table columns as arrays
    @staticmethod
    def _icol(df, k):
        """A position column as int64 (the reference converts each value with int(float(v)))."""
        a = np.asarray(df[k])
        if a.dtype.kind in 'iu':
            return a.astype(np.int64)
        return np.trunc(np.asarray(a, dtype=np.float64)).astype(np.int64)

    @staticmethod
    def _tdiv(a, r):
        """int(a / r) of the reference: division truncated towards zero."""
        return np.where(a >= 0, a // r, -((-a) // r))

    @staticmethod
    def _groups(df):
        """[(chromosome, row indices in table order)], chromosomes in order of first appearance."""
        code, uniq = pd.factorize(np.asarray(df['chr'].astype(str)), sort=False)      # codes in order of first appearance
        return [(str(c), np.nonzero(code == k)[0]) for k, c in enumerate(uniq)]

    # ------------------------------------------------------------------ observed mean (score only)
    def getMean(self, df, mask='0'):
        """getStripe.py:501-534 (K: k_stripe_mean)."""
        n = len(df)
        listM = [0 for _ in range(n)]
        listS = [0 for _ in range(n)]
        if n == 0:
            return listM, listS
        r = self.resol
        xs, xe, ys, ye = (self._icol(df, k) for k in ('pos1', 'pos2', 'pos3', 'pos4'))
        rects = np.zeros(n, dtype=RECT_DTYPE)
        rects['row0'], rects['row1'] = ys // r, -(-ye // r)           # cooler's extent of 'chr:start-end'
        rects['col0'], rects['col1'] = xs // r, -(-xe // r)
        M, S = np.zeros(n), np.zeros(n)
        for chrom, idx in self._groups(df):
            M[idx], S[idx] = self.backend.stripe_mean(self._band(chrom, whole=False), rects[idx])
        return list(M), list(S)

    # ------------------------------------------------------------------ p-value
    def pvalue(self, bgleft_up, bgright_up, bgleft_down, bgright_down, df):
        """getStripe.py:536-606 (K: k_pvalue).  The direction of every stripe -- including the
        background rows a stripe INHERITS from its predecessor when it touches neither end of the
        diagonal (:584-597) -- is decided here exactly like the reference's loop, on whole columns."""
        bs = int(50000 / self.resol)
        resol = self.resol
        n = len(df)
        if n == 0:
            return []
        self.backend.set_background(bgleft_up, bgright_up, bgleft_down, bgright_down)
        groups = self._groups(df)
        pos1, pos2, pos3, pos4 = (self._icol(df, k) for k in ('pos1', 'pos2', 'pos3', 'pos4'))
        chrlen = np.zeros(n, dtype=np.int64)
        for chrom, idx in groups:
            chrlen[idx] = int(self.chromnames2sizes[chrom])
        leftmost = np.maximum(pos1 - bs * resol, 1)                    # :552-559
        rightmost = np.minimum(pos2 + bs * resol, chrlen)
        stripes = np.zeros(n, dtype=PV_STRIPE_DTYPE)
        stripes['col0'], stripes['col1'] = leftmost // resol, -(-rightmost // resol)
        stripes['row0'], stripes['row1'] = pos3 // resol, -(-pos4 // resol)
        x1, x2 = self._tdiv(pos1 - 1, resol), self._tdiv(pos2, resol)
        y1, y2 = self._tdiv(pos3 - 1, resol), self._tdiv(pos4, resol)
        h = stripes['row1'].astype(np.int64) - stripes['row0']
        stripes['upbase'] = y2 - y1
        mode = np.where(x1 == y1, 0, np.where(x2 == y2, 1, 2))
        stripes['mode'] = mode
        need = mode == 2
        if need.any():
            # `bleft` / `bright` keep the rows the last stripe that set them left behind (:584-597)
            src = np.where((mode != 2) & (h > 0), np.arange(n), -1)
            last = np.maximum.accumulate(src)
            if (last[need] < 0).any():
                raise UnboundLocalError("local variable 'bleft' referenced before assignment")  # as the reference
            j = last[need]
            d_up = (y2[j] - y1[j]) - (h[j] - 1) - 1
            prev_row = np.where(mode[j] == 0, np.minimum(h[j] - 1, 399), np.where(d_up >= 400, 399, d_up))
            stripes['fixed_row'][need] = prev_row % 400
            stripes['fixed_tab'][need] = np.where(mode[j] == 0, 1, 0)
        P = np.zeros(n)
        for chrom, idx in groups:
            P[idx] = self.backend.pvalue(self._band(chrom, whole=False), bs, stripes[idx])
        return P.tolist()

    # ------------------------------------------------------------------ Stripiness
    def scoringstripes(self, df, expecVal, mask='0'):
        """getStripe.py:608-788 + stats.py:184-199 (K: k_stripiness)."""
        bs = int(50000 / self.resol)
        resol = self.resol
        is_masking = mask != '0'
        if is_masking:
            m = mask.split(':')
            mask_chr = m[0]
            mask_start = int(m[1].split('-')[0])
            mask_end = int(m[1].split('-')[1])
            mask_x_start = int(mask_start / resol)
            mask_x_end = int(mask_end / resol)

        def mask_range(start_index, end_index, dim, what):
            """masking() (:620-639): relative index range to blank; IndexError like the reference
            when the range reaches `dim` (its L is end-start+1)."""
            rel0 = mask_x_start - start_index
            rel1 = rel0 + (mask_x_end - mask_x_start)
            L = end_index - start_index + 1
            lo, hi = max(rel0, 0), min(rel1, L - 1)
            if lo > hi:
                return 1, 0
            if hi >= dim:
                raise IndexError('index %d is out of bounds for axis %d with size %d' % (dim, what, dim))
            return lo, hi

        nrow = df.shape[0]
        G, CM, CT = np.zeros(nrow), np.zeros(nrow), np.zeros(nrow)
        if nrow == 0:
            return [], [], []
        XS, XE, YS, YE = (self._icol(df, k) for k in ('pos1', 'pos2', 'pos3', 'pos4'))
        for c, idx in self._groups(df):
            is_mask = is_masking and (mask_chr == c)
            chrom_idx = self.chromnames.index(str(c))
            chrom_bin_size = int(np.ceil(self.chromsizes[chrom_idx] / resol))
            exval = np.asarray(expecVal[str(c)], dtype=np.float64)
            xs, xe, ys, ye = XS[idx], XE[idx], YS[idx], YE[idx]
            xsi, xei = self._tdiv(xs, resol), self._tdiv(xe, resol)       # :668-673
            ysi, yei = self._tdiv(ys, resol), self._tdiv(ye, resol)
            leftmost = np.maximum(xsi - bs, 1)                            # :675-680
            rightmost = np.where(xei + bs >= chrom_bin_size, chrom_bin_size - 1, xei + bs)
            st = np.zeros(len(idx), dtype=SCORE_STRIPE_DTYPE)
            st['row0'], st['row1'] = ys // resol, -(-ye // resol)         # the three fetches (:682-696) as bin extents
            st['col0'][:, 0], st['col1'][:, 0] = xs // resol, -(-xe // resol)
            st['col0'][:, 1], st['col1'][:, 1] = leftmost, xsi
            st['col0'][:, 2], st['col1'][:, 2] = xei, rightmost
            st['ex0'][:, 0], st['ex0'][:, 1], st['ex0'][:, 2] = xsi, leftmost, xei + 1
            st['ey0'] = ysi
            st['mirror'] = np.where(xs == ys, 0, 1)
            # np.divide(obs, exp) needs equal shapes (:687,693,699); the reference raises otherwise
            ex_w = np.stack([xei - xsi, xsi - leftmost, rightmost - xei], axis=1)
            ex_h = yei - ysi
            ow = st['col1'].astype(np.int64) - st['col0']
            oh = st['row1'].astype(np.int64) - st['row0']
            bad = (oh[:, None] != ex_h[:, None]) | (ow != ex_w)
            if bad.any():
                k, b = np.argwhere(bad)[0]
                raise ValueError('operands could not be broadcast together with shapes (%d,%d) (%d,%d) '
                                 % (oh[k], ow[k, b], ex_h[k], ex_w[k, b]))
            st['mcol0'], st['mcol1'] = 1, 0
            st['mrow0'], st['mrow1'] = 1, 0
            if is_mask:
                hit = (mask_start > np.minimum(xs - 50000, ys)) & (mask_start < np.maximum(xe + 50000, ye))
                for k in np.nonzero(hit)[0].tolist():
                    s = st[k]
                    h = int(s['row1'] - s['row0'])
                    starts = (int(xsi[k]), int(leftmost[k]), int(xei[k]) + 1)
                    ends = (int(xei[k]), int(xsi[k]), int(rightmost[k]))
                    for b in range(3):
                        lo, hi = mask_range(starts[b], ends[b], int(s['col1'][b] - s['col0'][b]), 1)
                        s['mcol0'][b], s['mcol1'][b] = lo, hi
                    s['mrow0'], s['mrow1'] = mask_range(int(ysi[k]), int(yei[k]), h, 0)
            G[idx], CM[idx], CT[idx] = self.backend.stripiness(self._band(str(c), whole=False), exval, st)
        return G.tolist(), list(CM), list(CT)

    # ------------------------------------------------------------------ stripe search
    def _chrom_frames(self, chrom, chridx):
        """(frames handle, starts, ends, number of the first frame) of a chromosome -- all its frames, or those
        of its frame span."""
        if chrom in self._frames:
            return self._frames[chrom]
        rowsize = int(np.ceil(self.chromsizes[chridx] / self.resol))
        nframes = math.ceil(rowsize / 200)
        starts, ends = [], []
        for idx in range(nframes):                                       # :794-799
            start = idx * 200 - 100
            end = (idx + 1) * 200 + 99
            if end >= rowsize:
                end = rowsize - 1
            if idx == 0:
                start = 0
            starts.append(start)
            ends.append(end)
        fa, fb = self.frame_span.get(str(chrom), (0, nframes))
        fa, fb = max(0, fa), min(nframes, fb)
        starts, ends = starts[fa:fb], ends[fa:fb]
        fr = self.backend.frames(self._band(chrom, whole=False), np.array(starts, np.int32), np.array(ends, np.int32))
        self._frames[chrom] = (fr, starts, ends, fa)
        return self._frames[chrom]

    def _search(self, chrom, chridx, M_levels):
        """All maxpixel levels of one chromosome in ONE batched device pass (cached): the reference
        re-runs every frame per level (stripenn.py:134-138); the kernels share the band reads."""
        key = (chrom, tuple(float(m) for m in M_levels))
        if key not in self._search_cache:
            self._search_begin(chrom, chridx, M_levels)
        recs = self._search_cache[key]
        if hasattr(recs, 'wait'):                                        # still in flight: collect it now
            t0 = time.time()
            recs = self._search_cache[key] = recs.wait()
            self.timing['stripe_search_s'] = self.timing.get('stripe_search_s', 0.0) + time.time() - t0
        return recs

    def _search_begin(self, chrom, chridx, M_levels):
        """Enqueue the search of one chromosome (all levels) without waiting for it, where the backend can
        (stp_stripe_search_begin); extract() starts the searches of ALL its chromosomes before it collects the first,
        so the host builds one chromosome's table while the device searches the next ones."""
        key = (chrom, tuple(float(m) for m in M_levels))
        if key in self._search_cache:
            return
        fr = self._chrom_frames(chrom, chridx)[0]
        t0 = time.time()
        begin = getattr(self.backend, 'stripe_search_begin', None)
        args = (fr, np.asarray(M_levels, dtype=np.float64), self.canny, self.minH, self.maxW, int(self.bfilter))
        self._search_cache[key] = begin(*args) if begin is not None else self.backend.stripe_search(*args)
        self.timing['stripe_search_s'] = self.timing.get('stripe_search_s', 0.0) + time.time() - t0

    def extract(self, MP, index, perc, bgleft_up, bgright_up, bgleft_down, bgright_down):
        """getStripe.py:790-862: candidate stripes of one maxpixel level, all chromosomes."""
        parts = []
        for chridx in range(len(self.chromnames)):                       # (no-ops once the searches are cached)
            self._search_begin(self.chromnames[chridx], chridx, MP[self.chromnames[chridx]])
        for chridx in range(len(self.chromnames)):
            chrom = self.chromnames[chridx]
            print('Chromosome: ' + str(chrom) + " / Maximum pixel: " + str(round(perc * 100, 3)) + "%")
            recs = self._search(chrom, chridx, MP[chrom])
            recs = recs[recs['level'] == index]
            fr, starts, ends, f0 = self._chrom_frames(chrom, chridx)
            parts.append(self._chrom_rows(chrom, int(self.chromsizes[chridx]), starts, ends, f0, fr, recs, perc))
        if sum(len(p['x']) for p in parts) == 0:
            res = pd.DataFrame(columns=EXTRACT_COLUMNS)
        else:
            # (round 6) both filters run on the level's COLUMNS -- plain numpy arrays, the chromosome code known from the loop --
            # and the table is built once, from the rows that survive: as DataFrames the two `iloc` selections, `reset_index`
            # and `assign` copied the 19-column table four times per level (0.04 s of the genome's 0.12 s in extract)
            cols = {k: np.concatenate([p[k] for p in parts]) for k in EXTRACT_COLUMNS}
            code = np.concatenate([np.full(len(p['x']), ci, dtype=np.int64) for ci, p in enumerate(parts)])
            # StripeSearch ends with RemoveRedundant over the rows of ONE frame (getStripe.py:1112): same filter,
            # all frames in one device call, pairs restricted to equal frame numbers
            keep = self._redundant_keep(cols, code, 'size', same_frame_only=True)
            cols = {k: v[keep] for k, v in cols.items()}
            code = code[keep]
            keep = self._redundant_keep(cols, code, 'size', same_frame_only=False)       # :852
            res = pd.DataFrame({k: v[keep] for k, v in cols.items()}, columns=EXTRACT_COLUMNS)
        p = self.pvalue(bgleft_up, bgright_up, bgleft_down, bgright_down, res)
        res.insert(res.shape[1], 'pvalue', np.asarray(p, dtype=np.float64))
        return res

    def _chrom_rows(self, chrom, chromsize, starts, ends, f0, fr, recs, perc):
        """Columns of all frames of one chromosome as StripeSearch builds them (getStripe.py:1081-1110),
        vectorised: bp coordinates through the compaction map, frame order then record order.  `f0` is the
        number of the first frame in `starts` (the `num` column counts frames from the chromosome's start)."""
        n = len(recs)
        f = recs['frame'].astype(np.int64)
        st = np.asarray(starts, dtype=np.int64)[f] if n else np.zeros(0, np.int64)
        en = np.asarray(ends, dtype=np.int64)[f] if n else np.zeros(0, np.int64)
        x = recs['x'].astype(np.int64)
        y = recs['y'].astype(np.int64)
        w = recs['w'].astype(np.int64)
        h = recs['h'].astype(np.int64)
        nz = fr.nz

        def start_bp(idx):
            return (st + nz[f, idx]) * self.resol + 1

        def end_bp(idx):
            b = st + nz[f, idx]
            e = b * self.resol + self.resol
            return np.where((b == en) & (e >= chromsize), chromsize, e)     # only a frame's last bin is clipped (:804-806)
        pos1, pos2 = start_bp(x), end_bp(x + w - 1)
        pos3, pos4 = start_bp(y), end_bp(y + h - 1)
        total = recs['total']
        name = np.empty(n, dtype=object)
        name[:] = chrom
        label = np.empty(n, dtype=object)
        label[:] = str(perc * 100) + '%'
        with np.errstate(divide='ignore', invalid='ignore'):
            mean = total / h / w
        return {'chr': name, 'pos1': pos1, 'pos2': pos2, 'chr2': name, 'pos3': pos3, 'pos4': pos4,
                'length': pos4 - pos3 + 1, 'width': pos2 - pos1 + 1, 'total': total, 'Mean': mean,
                'maxpixel': label, 'num': f + f0, 'start': st, 'end': en, 'x': x, 'y': y,
                'h': h, 'w': w, 'medpixel': fr.medpixel[f].astype(np.float64) if n else np.zeros(0)}

    def StripeSearch(self, submat, num, start, end, M, perc, chr, framesize, start_array, end_array):
        """getStripe.py:864-1114 on one dense frame matrix, as search_frame calls it (`submat` after its zero-column
        removal, at most 400 x 400): the matrix is laid out as a one-frame diagonal band, uploaded, and sent through
        the same device chain as extract() -- with the frame's columns kept as they are (no second zero-column
        removal, no more-than-10-columns rule: those belong to search_frame, :812-821)."""
        submat = np.asarray(submat, dtype=np.float64)
        S = submat.shape[0]
        if submat.ndim != 2 or submat.shape[1] != S or S < 1 or S > 400:
            raise ValueError('StripeSearch: submat must be square and at most 400 x 400 (got %r)' % (submat.shape,))
        hw = self.halfwidth
        band = np.zeros((S, 2 * hw), dtype=np.float64)
        rr = np.arange(S)[:, None]
        cc = rr + np.arange(-hw, hw)[None, :]
        ok = (cc >= 0) & (cc < S)
        band[ok] = submat[np.broadcast_to(rr, cc.shape)[ok], cc[ok]]
        hb = self.backend.open_chrom(band)
        try:
            fr = self.backend.frames(hb, np.array([0], np.int32), np.array([S - 1], np.int32), keep_all=True)
            try:
                recs = self.backend.stripe_search(fr, np.array([float(M)]), self.canny, self.minH, self.maxW,
                                                  int(self.bfilter))
                medpixel = float(fr.medpixel[0])
            finally:
                fr.close()
        finally:
            self.backend.close_chrom(hb)
        sa, ea = np.asarray(start_array), np.asarray(end_array)
        n = len(recs)
        x, y = recs['x'].astype(np.int64), recs['y'].astype(np.int64)
        w, h = recs['w'].astype(np.int64), recs['h'].astype(np.int64)
        total = recs['total']
        with np.errstate(divide='ignore', invalid='ignore'):
            mean = total / h / w
        result = pd.DataFrame({'chr': [chr] * n, 'pos1': sa[x], 'pos2': ea[x + w - 1], 'chr2': [chr] * n, 'pos3': sa[y],
                               'pos4': ea[y + h - 1], 'length': ea[y + h - 1] - sa[y] + 1,
                               'width': ea[x + w - 1] - sa[x] + 1, 'total': total, 'Mean': mean,
                               'maxpixel': [str(perc * 100) + '%'] * n, 'num': [num] * n, 'start': [start] * n,
                               'end': [end] * n, 'x': x, 'y': y, 'h': h, 'w': w, 'medpixel': [medpixel] * n},
                              columns=EXTRACT_COLUMNS)
        return self.RemoveRedundant(result, 'size')

    # ------------------------------------------------------------------ redundancy filter
    def RemoveRedundant(self, df, by):
        """getStripe.py:1116-1196 (K: k_remove_redundant): pairs of rows of one chromosome whose frame
        numbers differ by at most one; the bucket table is built here, the pair tests run on the device."""
        return self._filter_redundant(df, by, same_frame_only=False)

    def _filter_redundant(self, df, by, same_frame_only):
        if by != 'size' and by != 'score' and by != 'pvalue':
            raise ValueError('"by" should be one of "size", "pvalue" and "score"')
        n = df.shape[0]
        if n == 0:
            return df
        code = pd.factorize(np.asarray(df['chr']), sort=False)[0].astype(np.int64)
        return df.iloc[self._redundant_keep(df, code, by, same_frame_only)]

    def _redundant_keep(self, df, code, by, same_frame_only):
        """Row numbers RemoveRedundant keeps; `df`: a DataFrame or a dict of equally long column arrays, `code`: one integer
        per row naming its chromosome."""
        n = len(code)
        if n == 0:
            return np.zeros(0, dtype=np.int64)
        num = np.asarray(df['num'], dtype=np.int64)
        span = int(num.max() - num.min()) + 3
        key = code * span + (num - num.min())
        order = np.argsort(key, kind='stable').astype(np.int32)
        skey = key[order]
        # bucket bounds of every row in the sorted order -- searchsorted(skey, key, 'left' / 'right') and, for the next frame number,
        # searchsorted(skey, key + 1, 'right') -- from the runs of equal keys (one pass instead of three binary searches per row)
        first = np.flatnonzero(np.concatenate(([True], skey[1:] != skey[:-1])))          # start of every run
        ends = np.append(first[1:], n)
        run = np.repeat(np.arange(len(first)), ends - first)                             # run of every sorted position
        ukey = skey[first]
        nxt = np.append(ukey[1:] == ukey[:-1] + 1, False)                                # the next run holds key + 1
        ends2 = np.where(nxt, np.append(ends[1:], n), ends)
        b0 = np.empty(n, dtype=np.int32); b1 = np.empty(n, dtype=np.int32)
        b0[order] = first[run]; b1[order] = ends[run]
        if same_frame_only:
            b2 = b1
        else:
            b2 = np.empty(n, dtype=np.int32); b2[order] = ends2[run]
        p = [np.asarray(df[c], dtype=np.int64) for c in ('pos1', 'pos2', 'pos3', 'pos4')]
        if np.any(p[1] == p[0]) or np.any(p[3] == p[2]):
            raise ZeroDivisionError('division by zero')                  # as the reference (:1142-1143)
        hh = np.asarray(df['h'], dtype=np.int32)
        ww = np.asarray(df['w'], dtype=np.int32)
        mode = {'size': 0, 'score': 1, 'pvalue': 2}[by]
        k = None
        if by == 'score':
            k = np.asarray(df['Stripiness'], dtype=np.float64)
        if by == 'pvalue':
            k = np.asarray(df['pvalue'], dtype=np.float64)
        keep = self.backend.remove_redundant(p[0], p[1], p[2], p[3], hh, ww, k, mode, order, b0, b1, b2)
        return np.where(keep)[0]
