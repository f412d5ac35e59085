"""HIP backend: the façade's view of libstripenn_hip.so (one context, one resident band per chromosome).

The façade (stripenn_amd/getStripe.py) talks to this small interface only, so the CPU test-suite
can exercise the façade's host logic with a checker backend that lives under tests/ (built on the
oracle).  The product constructs HipBackend and nothing else: no CPU fallback exists here.
"""
import ctypes as C

import numpy as np

from . import hip

NULL_SAMPLE_DTYPE = np.dtype([('row0', np.int32), ('nrow', np.int32), ('col0', np.int32), ('ncol', np.int32),
                              ('x', np.int32), ('yoff', np.int32)])
PV_STRIPE_DTYPE = np.dtype([('row0', np.int32), ('row1', np.int32), ('col0', np.int32), ('col1', np.int32),
                            ('mode', np.int32), ('upbase', np.int32), ('fixed_row', np.int32), ('fixed_tab', np.int32)])
SCORE_STRIPE_DTYPE = np.dtype([('row0', np.int32), ('row1', np.int32), ('col0', np.int32, (3,)), ('col1', np.int32, (3,)),
                               ('ex0', np.int32, (3,)), ('ey0', np.int32), ('mirror', np.int32),
                               ('mcol0', np.int32, (3,)), ('mcol1', np.int32, (3,)), ('mrow0', np.int32),
                               ('mrow1', np.int32)])
RECT_DTYPE = np.dtype([('row0', np.int32), ('row1', np.int32), ('col0', np.int32), ('col1', np.int32)])


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def score_inputs(recs, nz, starts, nbins, bs):
    """Bin rectangles of raw stripe records (hip.REC_DTYPE; `nz` / `starts`: compaction map and first bin of the
    frames they index) in the two layouts the score kernels take -- the arithmetic of getStripe.pvalue
    (getStripe.py:552-563, 583-597) and scoringstripes (:668-697) for stripes that sit on whole bins, vectorised.
    Used where candidates go straight from the search to the score kernels (bench.py, tests)."""
    f = recs['frame']
    base = np.asarray(starts)[f]
    nzr = nz.ravel(); fo = f * nz.shape[1]
    x0 = base + nzr.take(fo + recs['x']); x1 = base + nzr.take(fo + recs['x'] + recs['w'] - 1)
    y0 = base + nzr.take(fo + recs['y']); y1 = base + nzr.take(fo + recs['y'] + recs['h'] - 1)
    n = len(recs)
    pv = np.zeros(n, dtype=PV_STRIPE_DTYPE)
    pv['row0'], pv['row1'] = y0, y1 + 1
    pv['col0'], pv['col1'] = np.maximum(x0 - bs, 0), np.minimum(x1 + 1 + bs, nbins)
    pv['mode'] = np.where(x0 == y0, 0, 1)
    pv['upbase'] = y1 + 1 - y0
    sc = np.zeros(n, dtype=SCORE_STRIPE_DTYPE)
    sc['row0'], sc['row1'] = y0, y1 + 1
    lm = np.minimum(np.maximum(x0 - bs, 1), x0); rm = np.minimum(x1 + 1 + bs, nbins - 1)
    sc['col0'][:, 0], sc['col1'][:, 0] = x0, x1 + 1
    sc['col0'][:, 1], sc['col1'][:, 1] = lm, x0
    sc['col0'][:, 2], sc['col1'][:, 2] = x1 + 1, np.maximum(rm, x1 + 1)
    sc['ex0'][:, 0], sc['ex0'][:, 1], sc['ex0'][:, 2] = x0, lm, x1 + 2
    sc['ey0'] = y0
    sc['mirror'] = np.where(x0 == y0, 0, 1)
    sc['mcol0'], sc['mcol1'], sc['mrow0'], sc['mrow1'] = 1, 0, 1, 0
    return pv, sc


class HipBackend:
    name = 'hip'

    def __init__(self, device=0):
        self.ctx = hip.Context(device)
        L = self.ctx.L
        vp = C.c_void_p
        L.stp_diag_sums.argtypes = [vp, vp, vp, vp, C.c_int32]
        L.stp_null_windows.argtypes = [vp, vp, vp, vp, C.c_int32, C.c_int32, vp, vp, vp, vp]
        L.stp_background_upload.argtypes = [vp, vp, vp, vp, vp, C.c_int32, C.POINTER(vp)]
        L.stp_background_free.argtypes = [vp, vp]
        L.stp_background_free.restype = None
        L.stp_pvalue.argtypes = [vp, vp, vp, C.c_int32, vp, C.c_int64, vp]
        L.stp_stripiness.argtypes = [vp, vp, vp, vp, C.c_int64, vp, vp, vp, vp]
        L.stp_score.argtypes = [vp, vp, vp, C.c_int32, vp, vp, vp, C.c_int64, vp, vp, vp, vp, vp]
        L.stp_stripe_mean.argtypes = [vp, vp, vp, C.c_int64, vp, vp]
        L.stp_window_plane.argtypes = [vp, vp, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_double, vp]
        L.stp_remove_redundant.argtypes = [vp, C.c_int64, vp, vp, vp, vp, vp, vp, vp, C.c_int32, vp, vp, vp, vp, vp]
        L.stp_select_create.argtypes = [vp, C.POINTER(vp)]
        L.stp_select_append.argtypes = [vp, vp, vp, C.c_int64]
        L.stp_select_append_pixels_ex.argtypes = [vp, vp, vp, vp, vp, C.c_int32, C.c_int64, vp, C.c_int64]
        L.stp_select_count.argtypes = [vp, vp, C.POINTER(C.c_int64)]
        L.stp_select_ranks.argtypes = [vp, vp, vp, C.c_int32, vp]
        L.stp_select_free.argtypes = [vp, vp]
        L.stp_select_free.restype = None
        self._bg = None
        self._bg_key = None

    # ---- chromosome band
    def open_chrom(self, band_host):
        return self.ctx.band_upload(band_host)

    def pack_chrom(self, px, hw, select=None):
        """Band of one chromosome straight from its cis pixels (dict of stripenn_amd.pixels.PixelSelector.chrom_pixels);
        `select`: an order-statistic select that receives the pixel values in the same pass (one PCIe trip)."""
        return self.ctx.band_pack(px['bin1'], px['bin2'], px['count'], px['weight'], px['lo'], px['nrows'], hw, select,
                                  bin1_offset=px.get('off'))

    def close_chrom(self, band):
        band.close()

    def frames(self, band, starts, ends, keep_all=False):
        return band.frames(starts, ends, keep_all)

    def band_nearest(self, band):
        """(right, left) nearest-positive-pixel distances of a packed band as int64 arrays, None for other bands."""
        try:
            r, l = band.nearest()
        except hip.StripennHipError as e:
            if e.code == hip.STP_E_UNSUPPORTED:
                return None
            raise
        return r.astype(np.int64), l.astype(np.int64)

    def stripe_search(self, frames, M_levels, sigma, minH, maxW, bfilter):
        return frames.stripe_search(M_levels, sigma=sigma, minH=minH, maxW=maxW, bfilter=bfilter)

    def stripe_search_begin(self, frames, M_levels, sigma, minH, maxW, bfilter):
        """The same search enqueued without waiting; `.wait()` on the result gives the records."""
        return frames.stripe_search_begin(M_levels, sigma=sigma, minH=minH, maxW=maxW, bfilter=bfilter)

    # ---- score path
    def diag_sums(self, band):
        n400 = -(-band.nrows // 400)
        ps = np.zeros((n400, 400), np.float64)
        pc = np.zeros((n400, 400), np.int64)
        self.ctx._chk(self.ctx.L.stp_diag_sums(self.ctx.h, band.h, _p(ps), _p(pc), n400))
        return ps, pc

    def null_windows(self, band, samples, bs, unit_matrix=None):
        samples = np.ascontiguousarray(samples, dtype=NULL_SAMPLE_DTYPE)
        n = len(samples)
        out = [np.zeros((400, n), np.float64) for _ in range(4)]
        if n:
            um = None
            if unit_matrix is not None:
                um = np.ascontiguousarray(unit_matrix, dtype=np.float64)
            self.ctx._chk(self.ctx.L.stp_null_windows(self.ctx.h, band.h, None if um is None else _p(um), _p(samples), n,
                                                      int(bs), *[_p(o) for o in out]))
        return out

    def set_background(self, lu, ru, ld, rd):
        key = tuple(id(t) for t in (lu, ru, ld, rd))
        if self._bg is not None and key == self._bg_key:
            return
        self.clear_background()
        tabs = [np.ascontiguousarray(t, dtype=np.float64) for t in (lu, ru, ld, rd)]
        ncol = tabs[0].shape[1]
        h = C.c_void_p()
        self.ctx._chk(self.ctx.L.stp_background_upload(self.ctx.h, *[_p(t) for t in tabs], ncol, C.byref(h)))
        self._bg, self._bg_key, self._bg_keep = h, key, (lu, ru, ld, rd)

    def clear_background(self):
        if self._bg is not None and self.ctx.h:
            self.ctx.L.stp_background_free(self.ctx.h, self._bg)
        self._bg = None
        self._bg_key = None

    def pvalue(self, band, bs, stripes):
        stripes = np.ascontiguousarray(stripes, dtype=PV_STRIPE_DTYPE)
        out = np.zeros(len(stripes), np.float64)
        if len(stripes):
            self.ctx._chk(self.ctx.L.stp_pvalue(self.ctx.h, band.h, self._bg, int(bs), _p(stripes), len(stripes), _p(out)))
        return out

    def stripiness(self, band, exval, stripes):
        stripes = np.ascontiguousarray(stripes, dtype=SCORE_STRIPE_DTYPE)
        exval = np.ascontiguousarray(exval, dtype=np.float64)
        n = len(stripes)
        g, m, t = np.zeros(n), np.zeros(n), np.zeros(n)
        status = np.zeros(n, np.int32)
        if n:
            self.ctx._chk(self.ctx.L.stp_stripiness(self.ctx.h, band.h, _p(exval), _p(stripes), n, _p(g), _p(m), _p(t),
                                                     _p(status)))
        if status.any():
            # the reference's np.delete(center, rowdel, axis=0) raises here (getStripe.py:735)
            raise IndexError('index out of bounds for axis 0: an all-NaN flank column maps to a row outside stripe %d'
                             % int(np.nonzero(status)[0][0]))
        return g, m, t

    def score(self, band, bs, exval, pv_stripes, sc_stripes):
        """p-value and Stripiness of the same stripes in ONE device call (stp_score): (p, g, O/E mean, O/E total)."""
        pv = np.ascontiguousarray(pv_stripes, dtype=PV_STRIPE_DTYPE)
        sc = np.ascontiguousarray(sc_stripes, dtype=SCORE_STRIPE_DTYPE)
        if len(pv) != len(sc):
            raise ValueError('score: %d p-value stripes but %d Stripiness stripes' % (len(pv), len(sc)))
        exval = np.ascontiguousarray(exval, dtype=np.float64)
        n = len(pv)
        p, g, m, t = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
        status = np.zeros(n, np.int32)
        if n:
            self.ctx._chk(self.ctx.L.stp_score(self.ctx.h, band.h, self._bg, int(bs), _p(exval), _p(pv), _p(sc), n, _p(p), _p(g), _p(m),
                                               _p(t), _p(status)))
        if status.any():
            raise IndexError('index out of bounds for axis 0: an all-NaN flank column maps to a row outside stripe %d'
                             % int(np.nonzero(status)[0][0]))
        return p, g, m, t

    def stripe_mean(self, band, rects):
        rects = np.ascontiguousarray(rects, dtype=RECT_DTYPE)
        n = len(rects)
        m, s = np.zeros(n), np.zeros(n)
        if n:
            self.ctx._chk(self.ctx.L.stp_stripe_mean(self.ctx.h, band.h, _p(rects), n, _p(m), _p(s)))
        return m, s

    def window_plane(self, band, row0, nrows, col0, ncols, M):
        """Green = blue plane of seeimage's heat map of a window of the resident band (stp_window_plane)."""
        out = np.empty((int(nrows), int(ncols)), np.float64)
        self.ctx._chk(self.ctx.L.stp_window_plane(self.ctx.h, band.h, int(row0), int(nrows), int(col0), int(ncols), float(M), _p(out)))
        return out

    # ---- redundancy filter
    def remove_redundant(self, p1, p2, p3, p4, h, w, key, by, order, b0, b1, b2):
        c = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
        p1, p2, p3, p4 = (c(a, np.int64) for a in (p1, p2, p3, p4))
        h, w, order, b0, b1, b2 = (c(a, np.int32) for a in (h, w, order, b0, b1, b2))
        n = len(p1)
        keep = np.ones(n, np.uint8)
        if n:
            k = None if key is None else c(key, np.float64)
            self.ctx._chk(self.ctx.L.stp_remove_redundant(self.ctx.h, n, _p(p1), _p(p2), _p(p3), _p(p4), _p(h), _p(w),
                                                           None if k is None else _p(k), int(by), _p(order), _p(b0), _p(b1),
                                                           _p(b2), _p(keep)))
        return keep.astype(bool)

    # ---- order statistics of positive pixels (getQuantile_original)
    def select_open(self):
        h = C.c_void_p()
        self.ctx._chk(self.ctx.L.stp_select_create(self.ctx.h, C.byref(h)))
        return h

    def select_append(self, sel, values):
        values = np.ascontiguousarray(values, dtype=np.float64).ravel()
        if values.size:
            self.ctx._chk(self.ctx.L.stp_select_append(self.ctx.h, sel, _p(values), values.size))

    def select_append_pixels(self, sel, bin1, bin2, count, weight):
        """Balanced values of stored pixels, off-diagonal ones twice, formed on the device (stp_select_append_pixels)."""
        bin1 = np.ascontiguousarray(bin1, dtype=np.int64); bin2 = np.ascontiguousarray(bin2, dtype=np.int64)
        count, ctype = hip.count_column(count)
        w = None if weight is None else np.ascontiguousarray(weight, dtype=np.float64)
        if len(bin1):
            self.ctx._chk(self.ctx.L.stp_select_append_pixels_ex(self.ctx.h, sel, _p(bin1), _p(bin2), _p(count), ctype, len(bin1),
                                                                 None if w is None else _p(w), 0 if w is None else len(w)))

    def select_count(self, sel):
        n = C.c_int64()
        self.ctx._chk(self.ctx.L.stp_select_count(self.ctx.h, sel, C.byref(n)))
        return int(n.value)

    def select_ranks(self, sel, ranks):
        ranks = np.ascontiguousarray(ranks, dtype=np.int64)
        out = np.zeros(len(ranks), np.float64)
        if len(ranks):
            self.ctx._chk(self.ctx.L.stp_select_ranks(self.ctx.h, sel, _p(ranks), len(ranks), _p(out)))
        return out

    def select_close(self, sel):
        self.ctx.L.stp_select_free(self.ctx.h, sel)

    def close(self):
        self.clear_background()
        self.ctx.close()
