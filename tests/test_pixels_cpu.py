"""cooler's pixel table as the input format (stripenn_amd/pixels.py): the host dense read, the .npz
round trip, the band restatement, and the whole compute pipeline fed from pixels vs from dense fetches
(oracle backend; the HIP packer is checked in test_gpu_pixels.py)."""
import os
import warnings

import numpy as np
import pytest

from oracle import oracle as O
from oracle_backend import OracleBackend
from stripenn_amd import io as sio, pixels, stripenn, synth

warnings.filterwarnings('ignore')
RESOL = 5000


class FetchOnly:
    """A selector that offers nothing but cooler's `fetch` (forces the dense route of the facade)."""

    def __init__(self, sel):
        self._sel = sel

    def fetch(self, *a):
        return self._sel.fetch(*a)


def _genome(nbins=(900, 700), seed=41, **kw):
    names = ['chrA', 'chrB'][:len(nbins)]
    chroms = {n: synth.SynthChrom(nb, seed + k, **kw) for k, (n, nb) in enumerate(zip(names, nbins))}
    return names, chroms, pixels.PixelTable.from_synth(names, chroms, RESOL)


def _dense_cooler_rule(ch):
    """Independent dense construction of cooler's balanced matrix of one synthetic chromosome."""
    n = ch.nbins
    cnt = ch.counts(0, n, 0, n)
    w = ch.w.copy(); w[ch.nan_bins] = np.nan
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing='ij')
    lo, hi = np.minimum(i, j), np.maximum(i, j)
    with np.errstate(invalid='ignore'):
        val = (cnt * w[lo]) * w[hi]
    return val          # count 0 in the row / column of a NaN-weight bin is 0 * NaN = NaN: cooler multiplies the DENSE block


def test_selector_fetch_is_coolers_dense_read():
    names, chroms, t = _genome()
    sel = pixels.PixelSelector(t, True)
    for nm in names:
        D = _dense_cooler_rule(chroms[nm])
        n = chroms[nm].nbins
        assert np.array_equal(sel.fetch(nm), D, equal_nan=True)
        got = sel.fetch('%s:%d-%d' % (nm, 100 * RESOL + 1, 300 * RESOL), '%s:%d-%d' % (nm, 40 * RESOL, n * RESOL))
        assert np.array_equal(got, D[100:300, 40:n], equal_nan=True)
    raw = pixels.PixelSelector(t, False).fetch('chrB')
    assert np.array_equal(raw, chroms['chrB'].counts(0, 700, 0, 700))
    with pytest.raises(ValueError):
        sel.fetch('chrA:0-%d' % (901 * RESOL))
    with pytest.raises(ValueError):
        pixels.PixelSelector(t, 'KR')


def test_table_validation_and_npz_round_trip(tmp_path):
    names, chroms, t = _genome()
    p = str(tmp_path / 't.npz')
    t.save(p)
    u = pixels.PixelTable.load(p)
    assert u.chromnames == t.chromnames and u.binsize == t.binsize
    for a in ('chromsizes', 'chrom_offset', 'bin1_id', 'bin2_id', 'count'):
        assert np.array_equal(getattr(u, a), getattr(t, a))
    assert np.array_equal(u.weights['weight'], t.weights['weight'], equal_nan=True)
    with pytest.raises(ValueError):
        pixels.PixelTable(['c'], [10 * RESOL], RESOL, [0, 10], [3, 1], [4, 2], [1, 1])      # unsorted
    with pytest.raises(ValueError):
        pixels.PixelTable(['c'], [10 * RESOL], RESOL, [0, 10], [5], [4], [1])                # lower triangle
    info = sio.open_matrix('pixels:' + p)
    assert list(info.chromnames) == names and info.binsize == RESOL and 'weight' in info.bins().columns


def test_band_restatement_equals_dense_route():
    names, chroms, t = _genome()
    sel = pixels.PixelSelector(t, True)
    for nm in names:
        px = sel.chrom_pixels(nm)
        band = O.band_from_pixels(px['bin1'], px['bin2'], px['count'], px['weight'], px['lo'], px['nrows'], 512)
        D = sel.fetch(nm)
        n = px['nrows']
        i = np.arange(n)[:, None]; j = i + np.arange(-512, 512)[None, :]
        exp = np.where((j >= 0) & (j < n), D[i, np.clip(j, 0, n - 1)], 0.0)
        assert np.array_equal(band, exp, equal_nan=True)


def test_compute_from_pixels_equals_compute_from_dense_fetches(tmp_path, monkeypatch):
    """The pipeline fed by the band packer + pixel-multiset quantile gives the same TSVs as the dense route."""
    names, chroms, t = _genome(nbins=(900, 700), stripe_every=90, stripe_gain=3.0)
    p = str(tmp_path / 't.npz')
    t.save(p)
    gw = O.gauss_weights(2.0)[0]
    outs = []
    for route in ('pixels', 'dense'):
        if route == 'dense':
            monkeypatch.setattr(stripenn, 'open_matrix', lambda cool: sio.MatrixInfo(
                t.chromnames, t.chromsizes, t.binsize, ['chrom', 'start', 'end', 'weight'],
                lambda balance: FetchOnly(pixels.PixelSelector(t, balance))))
        out = str(tmp_path / route)
        stripenn.compute('pixels:' + p, out, 'weight', 'all', 2.0, 10, 8, '0.97,0.99', 1, 0.5, '0', False, 3, 7,
                         force=True, backend=OracleBackend(gauss_w=gw))
        outs.append([open(os.path.join(out, f)).read() for f in ('result_unfiltered.tsv', 'result_filtered.tsv')])
    assert outs[0] == outs[1]
    assert outs[0][0].count('\n') > 5


def test_global_bin_slicing_and_slow_quantile():
    """`matrix[r0:r1, c0:c1]` with global bin ids (what the reference's `-s` quantile reads), incl. a block
    that spans two chromosomes (trans pixels do not exist in this table: count 0 there)."""
    import pandas as pd
    from stripenn_amd import getStripe as GS
    names, chroms, t = _genome()
    sel = pixels.PixelSelector(t, True)
    DA, DB = _dense_cooler_rule(chroms['chrA']), _dense_cooler_rule(chroms['chrB'])
    G = np.zeros((1600, 1600)); G[:900, :900] = DA; G[900:, 900:] = DB
    wg = t.weights['weight']
    with np.errstate(invalid='ignore'):
        G = G + 0.0 * np.outer(wg, wg)       # the trans block holds no pixels: 0 * (b1 * b2), NaN along the NaN-weight bins
    for (r0, r1, c0, c1) in ((0, 900, 0, 900), (850, 1000, 700, 1600), (1000, 1600, 900, 1600), (10, 11, 0, 1600)):
        assert np.array_equal(sel[r0:r1, c0:c1], G[r0:r1, c0:c1], equal_nan=True)

    class Info:
        chromsizes = pd.Series(t.chromsizes, index=names)
        binsize = RESOL
    obj = GS.getStripe(sel, RESOL, 10, 8, 2.0, names, names, t.chromsizes, t.chromsizes, 1, 3, 1, backend=OracleBackend())
    slow = obj.getQuantile_slow(Info, names, [0.95, 0.99])
    fast = obj.getQuantile_original(Info, names, [0.95, 0.99])
    for n in names:
        assert np.all(np.isfinite(slow[n])) and np.allclose(slow[n], fast[n], rtol=0.05)


def test_row_nonzero_equals_dense_row_sums():
    names, chroms, t = _genome(nan_frac=0.05)
    for balance in (True, False):
        sel = pixels.PixelSelector(t, balance)
        for nm, n in (('chrA', 900), ('chrB', 700)):
            for (r0, r1, c0, c1) in ((0, n, 0, n), (100, 300, 0, 700), (405, 500, 5, 900 if nm == 'chrA' else 700), (0, 50, 400, 600)):
                p1 = '%s:%d-%d' % (nm, r0 * RESOL + 1, r1 * RESOL)
                p2 = '%s:%d-%d' % (nm, c0 * RESOL + 1, min(c1, n) * RESOL)
                D = sel.fetch(p1, p2)
                D = np.where(np.isnan(D), 0.0, D)
                assert np.array_equal(sel.row_nonzero(p1, p2), D.sum(axis=1) != 0), (balance, nm, r0, r1, c0, c1)
    # a table with a negative value cannot answer without the dense sum
    neg = pixels.PixelTable(['c'], [20 * RESOL], RESOL, [0, 20], [1, 2], [3, 4], [5, -5])
    assert pixels.PixelSelector(neg, False).row_nonzero('c') is None


def test_divisive_and_multiplicative_columns_follow_coolers_rule():
    """cooler's matrix(balance=name): bias = weights[name], `1 / bias` for the divisive columns it knows by name
    (KR, VC, SQRT_VC -- what hic2cool writes; --norm KR is the reference CLI's default), then
    `arr * np.outer(bias1, bias2)`.  Hand-computed on a 5-bin table with weights that are NOT exactly
    representable products, so the multiplication order is visible in the last bit."""
    rng = np.random.default_rng(3)
    w = rng.uniform(0.3, 3.0, 5)
    w[3] = np.nan
    b1 = np.array([0, 0, 1, 1, 2, 3, 4]); b2 = np.array([0, 2, 1, 4, 2, 4, 4])
    cn = np.array([7, 3, 11, 5, 2, 9, 13], dtype=np.int32)
    t = pixels.PixelTable(['c'], [5 * RESOL], RESOL, [0, 5], b1, b2, cn, {'weight': w, 'KR': w, 'VC': w, 'SQRT_VC': w, 'VC_SQRT': w})
    cnt = np.zeros((5, 5)); cnt[b1, b2] = cn; cnt[b2, b1] = cn
    with np.errstate(invalid='ignore'):
        mult = cnt * np.outer(w, w)
        div = cnt * np.outer(1.0 / w, 1.0 / w)
    assert np.array_equal(pixels.PixelSelector(t, True).fetch('c'), mult, equal_nan=True)
    assert np.array_equal(pixels.PixelSelector(t, 'weight').fetch('c'), mult, equal_nan=True)
    assert np.array_equal(pixels.PixelSelector(t, 'VC_SQRT').fetch('c'), mult, equal_nan=True)    # not in cooler's divisive set
    for name in ('KR', 'VC', 'SQRT_VC'):
        got = pixels.PixelSelector(t, name).fetch('c')
        assert np.array_equal(got, div, equal_nan=True), name
        assert got[0, 2] == 3 * ((1.0 / w[0]) * (1.0 / w[2]))
    assert np.array_equal(pixels.PixelSelector(t, 'KR', divisive=False).fetch('c'), mult, equal_nan=True)
    # the band restatement and the selector agree on the divisive column too
    sel = pixels.PixelSelector(t, 'KR')
    px = sel.chrom_pixels('c')
    band = O.band_from_pixels(px['bin1'], px['bin2'], px['count'], px['weight'], 0, 5, 512)
    for i in range(5):
        for j in range(5):
            assert (band[i, j - i + 512] == div[i, j]) or (np.isnan(div[i, j]) and np.isnan(band[i, j - i + 512]))


def test_float_counts_are_kept_not_truncated():
    """pixels/count as float64 (coolers written with --count-as-float, merged / scaled coolers): integer-valued
    columns are narrowed to int32, anything else stays float64 and flows through the dense read, the band
    restatement and the nearest-pixel table unchanged (a value of 0.5 must not disappear)."""
    b = np.array([0, 0, 1, 2]); c = np.array([0, 1, 1, 2])
    ok = pixels.PixelTable(['c'], [3 * RESOL], RESOL, [0, 3], b, c, np.array([2.0, 5.0, 1.0, 7.0]))
    assert ok.count.dtype == np.int32 and ok.count.tolist() == [2, 5, 1, 7]
    w = np.array([0.5, 2.0, 1.5])
    t = pixels.PixelTable(['c'], [3 * RESOL], RESOL, [0, 3], b, c, np.array([2.0, 0.5, 1.25, 0.0]), {'weight': w})
    assert t.count.dtype == np.float64
    sel = pixels.PixelSelector(t, True)
    D = sel.fetch('c')
    exp = np.zeros((3, 3))
    for i, j, v in zip(b, c, [2.0, 0.5, 1.25, 0.0]):
        exp[i, j] = exp[j, i] = v * (w[i] * w[j])
    assert np.array_equal(D, exp) and D[0, 1] == 0.5 * (0.5 * 2.0)
    px = sel.chrom_pixels('c')
    band = O.band_from_pixels(px['bin1'], px['bin2'], px['count'], px['weight'], 0, 3, 512)
    assert band[0, 512 + 1] == D[0, 1] and band[1, 512 - 1] == D[1, 0] and band[2, 512] == 0.0
    right, left = O.nearest_from_pixels(px['bin1'], px['bin2'], px['count'], px['weight'], 0, 3)
    assert right.tolist()[:2] == [0, 0] and right[2] == np.iinfo(np.int32).max      # the stored 0.0 is not a positive pixel
    big = pixels.PixelTable(['c'], [3 * RESOL], RESOL, [0, 3], b, c, np.array([2, 2**40, 1, 3], dtype=np.int64))
    assert big.count.dtype == np.float64 and big.count[1] == float(2**40)
