"""Checker backend for the façade (TESTS ONLY): the same interface as stripenn_amd.backend.HipBackend,
implemented with the CPU oracle.  Lets the CPU test-suite run the façade's host logic (frames, pools,
PRNG order, region arithmetic, RemoveRedundant, table assembly) against the reference's goldens."""
import numpy as np

from oracle import oracle as O
from stripenn_amd import hip


class _Band:
    def __init__(self, band):
        self.a = np.ascontiguousarray(band, dtype=np.float64)
        self.nrows, W = self.a.shape
        self.hw = W // 2

    def block(self, r0, r1, c0, c1):
        rr = np.arange(r0, r1)[:, None]
        cc = np.arange(c0, c1)[None, :]
        d = cc - rr + self.hw
        ok = (d >= 0) & (d < 2 * self.hw) & (rr >= 0) & (rr < self.nrows) & (cc >= 0) & (cc < self.nrows)
        return np.where(ok, self.a[np.clip(rr, 0, self.nrows - 1), np.clip(d, 0, 2 * self.hw - 1)], 0.0)

    def close(self):
        pass


class _Frames:
    def __init__(self, band, starts, ends, keep_all=False):
        self.band = band
        self.n = len(starts)
        self.S = np.zeros(self.n, np.int32)
        self.nz = np.zeros((self.n, 400), np.int16)
        self.medpixel = np.zeros(self.n)
        self.D = []
        for f in range(self.n):
            D, nz = O.frame_dense(band.block, int(starts[f]), int(ends[f]))
            if keep_all:
                nz = np.arange(D.shape[0])
            if keep_all or len(nz) > 10:
                self.S[f] = len(nz)
                self.nz[f, :len(nz)] = nz
                Dc = np.ascontiguousarray(D[np.ix_(nz, nz)])
                self.medpixel[f] = O.medpixel(Dc)
                self.D.append(Dc)
            else:
                self.D.append(None)

    def close(self):
        pass


class OracleBackend:
    name = 'oracle'

    def __init__(self, gauss_w=None):
        self.gauss_w = gauss_w
        self.bg = None

    def open_chrom(self, band_host):
        return _Band(band_host)

    def pack_chrom(self, px, hw, select=None):
        if select is not None:
            self.select_append_pixels(select, px['bin1'], px['bin2'], px['count'], px['weight'])
        b = _Band(O.band_from_pixels(px['bin1'], px['bin2'], px['count'], px['weight'], px['lo'], px['nrows'], hw))
        b.near = O.nearest_from_pixels(px['bin1'], px['bin2'], px['count'], px['weight'], px['lo'], px['nrows'])
        return b

    def close_chrom(self, band):
        pass

    def frames(self, band, starts, ends, keep_all=False):
        return _Frames(band, starts, ends, keep_all)

    def band_nearest(self, band):
        return getattr(band, 'near', None)

    def stripe_search(self, frames, M_levels, sigma, minH, maxW, bfilter):
        out = []
        for f in range(frames.n):
            if frames.D[f] is None:
                continue
            for li, M in enumerate(M_levels):
                recs, tot = O.stripe_search(frames.D[f], float(M), sigma=sigma, minH=minH, maxW=maxW, bf=bfilter,
                                            gw=self.gauss_w)
                for k in range(len(recs)):
                    b, ud, x, y, w, h = (int(v) for v in recs[k])
                    out.append((f, li, b, ud, x, y, w, h, float(tot[k])))
        return np.array(out, dtype=hip.REC_DTYPE)

    def diag_sums(self, band):
        return O.diag_sums(band.block, band.nrows)

    def null_windows(self, band, samples, bs, unit_matrix=None):
        n = len(samples)
        if n == 0:
            return [np.zeros((400, 0)) for _ in range(4)]
        src = band.block if unit_matrix is None else (lambda r0, r1, c0, c1: unit_matrix)
        # one oracle call per run of samples that share a unit geometry (the facade batches several units per call)
        geo = np.stack([samples[k].astype(np.int64) for k in ('row0', 'nrow', 'col0', 'ncol', 'yoff')], axis=1)
        cuts = [0] + [i for i in range(1, n) if not np.array_equal(geo[i], geo[i - 1])] + [n]
        parts = []
        for a, b in zip(cuts, cuts[1:]):
            s0 = samples[a]
            parts.append(O.null_windows(src, int(s0['row0']), int(s0['nrow']), int(s0['col0']), int(s0['ncol']),
                                        samples['x'][a:b].astype(np.int64), int(s0['yoff']), bs))
        return [np.concatenate([p[k] for p in parts], axis=1) for k in range(4)]

    def set_background(self, lu, ru, ld, rd):
        self.bg = (np.asarray(lu), np.asarray(ru), np.asarray(ld), np.asarray(rd))

    def pvalue(self, band, bs, stripes):
        return np.array([O.pvalue_one(band.block(int(s['row0']), int(s['row1']), int(s['col0']), int(s['col1'])), bs,
                                      int(s['mode']), int(s['upbase']), int(s['fixed_row']), int(s['fixed_tab']), self.bg)
                         for s in stripes])

    def score(self, band, bs, exval, pv_stripes, sc_stripes):
        return (self.pvalue(band, bs, pv_stripes),) + tuple(self.stripiness(band, exval, sc_stripes))

    def stripiness(self, band, exval, stripes):
        g, m, t = [], [], []
        for s in stripes:
            obs = [band.block(int(s['row0']), int(s['row1']), int(s['col0'][b]), int(s['col1'][b])) for b in range(3)]
            r = O.stripiness_one(obs, exval, [int(v) for v in s['ex0']], int(s['ey0']), int(s['mirror']),
                                 [(int(s['mcol0'][b]), int(s['mcol1'][b])) for b in range(3)],
                                 (int(s['mrow0']), int(s['mrow1'])))
            g.append(r[0]); m.append(r[1]); t.append(r[2])
        return np.array(g), np.array(m), np.array(t)

    def stripe_mean(self, band, rects):
        r = [O.stripe_mean_one(band.block(int(q['row0']), int(q['row1']), int(q['col0']), int(q['col1']))) for q in rects]
        return np.array([a for a, _ in r]), np.array([b for _, b in r])

    def window_plane(self, band, row0, nrows, col0, ncols, M):
        return O.window_rgb(band.block(row0, row0 + nrows, col0, col0 + ncols), M)[..., 1]    # seeimage.py:78-85 in numpy

    def close(self):
        pass

    # order statistics of positive pixels: numpy partition as the checker
    def select_open(self):
        return []

    def select_append(self, sel, values):
        sel.append(np.asarray(values, dtype=np.float64).ravel())

    def select_append_pixels(self, sel, bin1, bin2, count, weight):
        v = np.asarray(count).astype(np.float64)
        if weight is not None:
            w = np.asarray(weight, np.float64)
            v = v * (w[np.asarray(bin1)] * w[np.asarray(bin2)])
        off = np.asarray(bin1) != np.asarray(bin2)
        v = np.concatenate([v, v[off]])
        sel.append(v[v > 0])

    def select_count(self, sel):
        return int(sum(len(v) for v in sel))

    def select_ranks(self, sel, ranks):
        a = np.sort(np.concatenate(sel)) if sel else np.zeros(0)
        return a[np.asarray(ranks, dtype=np.int64)]

    def select_close(self, sel):
        sel.clear()

    # redundancy filter: the reference's pair loops (getStripe.py:1116-1161) over the bucket table
    def remove_redundant(self, p1, p2, p3, p4, h, w, key, by, order, b0, b1, b2):
        n = len(p1)
        keep = np.ones(n, dtype=bool)
        p1, p2, p3, p4 = (np.asarray(a).tolist() for a in (p1, p2, p3, p4))
        h = np.asarray(h).tolist(); w = np.asarray(w).tolist()
        key = None if key is None else np.asarray(key).tolist()
        seen = set()
        for i in range(n):
            for q in range(int(b0[i]), int(b2[i])):
                j = int(order[q])
                if j == i:
                    continue
                a, b = (i, j) if i < j else (j, i)
                if (a, b) in seen:
                    continue
                seen.add((a, b))
                ox = max(0, min(p2[a], p2[b]) - max(p1[a], p1[b]) + 1)
                oy = max(0, min(p4[a], p4[b]) - max(p3[a], p3[b]) + 1)
                s_x = ox / min(p2[a] - p1[a], p2[b] - p1[b])
                s_y = oy / min(p4[a] - p3[a], p4[b] - p3[b])
                if s_x > 0.2 and s_y > 0.2:
                    if by == 0:
                        drop = a if h[a] / w[a] <= h[b] / w[b] else b
                    elif by == 1:
                        drop = a if key[a] <= key[b] else b
                    else:
                        drop = a if key[a] > key[b] else b
                    keep[drop] = False
        return keep
