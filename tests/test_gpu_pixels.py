"""stp_band_pack on the GPU: the band built from cooler's pixel table equals the CPU restatement bit for
bit (NaN positions included), and the compute pipeline fed from a pixel table reproduces the dense route."""
import os
import warnings

import numpy as np
import pytest

from oracle import oracle as O
from stripenn_amd import backend as BK, io as sio, pixels, stripenn, synth

pytestmark = pytest.mark.gpu
warnings.filterwarnings('ignore')
RESOL = 5000


def _same(a, b):
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


def test_packed_band_equals_restatement(hip_ctx):
    names = ['chrA', 'chrB']
    chroms = {'chrA': synth.SynthChrom(900, 41), 'chrB': synth.SynthChrom(1300, 42, nan_frac=0.02)}
    t = pixels.PixelTable.from_synth(names, chroms, RESOL)
    for balance in (True, False):
        sel = pixels.PixelSelector(t, balance)
        for nm in names:
            px = sel.chrom_pixels(nm)
            for hw in (512, 576):
                exp = O.band_from_pixels(px['bin1'], px['bin2'], px['count'], px['weight'], px['lo'], px['nrows'], hw)
                band = hip_ctx.band_pack(px['bin1'], px['bin2'], px['count'], px['weight'], px['lo'], px['nrows'], hw)
                assert _same(band.download(), exp), (balance, nm, hw)
                band.close()
    # the whole table at once: trans-free here, but bins outside [lo, lo + nrows) must be skipped
    lo, hi = t.chrom_bins('chrB')
    band = hip_ctx.band_pack(t.bin1_id, t.bin2_id, t.count, t.weights['weight'], lo, hi - lo, 512)
    exp = O.band_from_pixels(t.bin1_id, t.bin2_id, t.count, t.weights['weight'], lo, hi - lo, 512)
    assert _same(band.download(), exp) and np.isnan(exp).any()
    # and the packed band is what the search consumes: same frames as the uploaded host band
    up = hip_ctx.band_upload(exp)
    st = np.array([0, 100, 300], np.int32); en = np.array([299, 499, 699], np.int32)
    fa, fb = band.frames(st, en), up.frames(st, en)
    assert np.array_equal(fa.S, fb.S) and np.array_equal(fa.nz, fb.nz) and np.array_equal(fa.medpixel, fb.medpixel)
    fa.close(); fb.close(); up.close(); band.close()
    with pytest.raises(ValueError):
        hip_ctx.band_pack(t.bin1_id[:5], t.bin2_id[:4], t.count[:5], None, 0, 900, 512)


def test_pack_chromosome_sized_table_in_chunks(hip_ctx):
    """chr16-size chromosome: > 8 M stored pixels, i.e. more than one staging chunk."""
    ch = synth.SynthChrom(19642, 16)
    t = pixels.PixelTable.from_synth(['chr16'], {'chr16': ch}, RESOL)
    assert len(t.count) > (1 << 23)
    px = pixels.PixelSelector(t, True).chrom_pixels('chr16')
    band = hip_ctx.band_pack(px['bin1'], px['bin2'], px['count'], px['weight'], px['lo'], px['nrows'], 512)
    exp = O.band_from_pixels(px['bin1'], px['bin2'], px['count'], px['weight'], px['lo'], px['nrows'], 512)
    assert _same(band.download(), exp)
    band.close()


def test_pack_from_csr_index_and_narrow_bin2(hip_ctx):
    """stp_band_pack_csr: the bin1 column as cooler's CSR index (expanded on the device), bin2_id as int32 or int64, int32 and
    float64 counts -- band, nearest-pixel table and the quantile's select equal those of the column form (stp_band_pack_select)
    and the CPU restatement; more than one staging chunk; empty rows; argument errors."""
    names = ['chrA', 'chrB']
    chroms = {'chrA': synth.SynthChrom(900, 41), 'chrB': synth.SynthChrom(1300, 42, nan_frac=0.02)}
    t = pixels.PixelTable.from_synth(names, chroms, RESOL)
    hb = BK.HipBackend(0)
    for balance in (True, False):
        sel = pixels.PixelSelector(t, balance)
        for nm in names:
            px = sel.chrom_pixels(nm)
            assert 'off' in px and len(px['off']) == px['nrows'] + 1 and px['off'][-1] == len(px['bin2'])
            exp = O.band_from_pixels(px['bin1'], px['bin2'], px['count'], px['weight'], px['lo'], px['nrows'], 512)
            ref = hb.ctx.band_pack(px['bin1'], px['bin2'], px['count'], px['weight'], px['lo'], px['nrows'], 512)
            rn = ref.nearest()
            for b2 in (px['bin2'].astype(np.int64), px['bin2'].astype(np.int32)):
                for cnt in (px['count'], px['count'].astype(np.float64) * 0.5):
                    e = exp if cnt is px['count'] else O.band_from_pixels(px['bin1'], px['bin2'], cnt, px['weight'], px['lo'], px['nrows'], 512)
                    s1, s2 = hb.select_open(), hb.select_open()
                    a = hb.ctx.band_pack(None, b2, cnt, px['weight'], px['lo'], px['nrows'], 512, s1, bin1_offset=px['off'])
                    b = hb.ctx.band_pack(px['bin1'], px['bin2'], cnt, px['weight'], px['lo'], px['nrows'], 512, s2)
                    assert _same(a.download(), e) and _same(b.download(), e), (balance, nm, b2.dtype, cnt.dtype)
                    an = a.nearest()
                    assert np.array_equal(an[0], rn[0]) and np.array_equal(an[1], rn[1])
                    n1, n2 = hb.select_count(s1), hb.select_count(s2)
                    assert n1 == n2 and n1 > 0
                    ranks = np.array([0, n1 // 3, n1 // 2, n1 - 1], dtype=np.int64)
                    assert np.array_equal(hb.select_ranks(s1, ranks), hb.select_ranks(s2, ranks))
                    hb.select_close(s1); hb.select_close(s2)
                    a.close(); b.close()
            ref.close()
    # a chromosome-size table (more than one staging chunk), int32 bin2, through pack_chrom as the driver calls it
    ch = synth.SynthChrom(19642, 16)
    t16 = pixels.PixelTable.from_synth(['chr16'], {'chr16': ch}, RESOL)
    t16.bin2_id = t16.bin2_id.astype(np.int32)
    px = pixels.PixelSelector(t16, True).chrom_pixels('chr16')
    assert px['bin2'].dtype == np.int32 and len(px['count']) > (1 << 23)
    band = hb.pack_chrom(px, 512)
    assert _same(band.download(), O.band_from_pixels(px['bin1'], px['bin2'], px['count'], px['weight'], px['lo'], px['nrows'], 512))
    band.close()
    # rows without pixels at both ends and in the middle
    off = np.array([0, 0, 2, 2, 2, 5, 5], dtype=np.int64)
    b1 = np.array([1, 1, 4, 4, 4], dtype=np.int64); b2 = np.array([1, 3, 4, 5, 5], dtype=np.int32); cn = np.array([3, 1, 4, 1, 5], dtype=np.int32)
    a = hb.ctx.band_pack(None, b2, cn, None, 0, 6, 448, bin1_offset=off)
    assert _same(a.download(), O.band_from_pixels(b1, b2.astype(np.int64), cn, None, 0, 6, 448))
    a.close()
    from stripenn_amd import hip
    for bad in (np.array([1, 0, 2, 2, 2, 5, 5]), np.array([0, 0, 2, 2, 2, 5, 4]), np.array([0, 3, 2, 2, 2, 5, 5])):
        with pytest.raises(hip.StripennHipError):
            hb.ctx.band_pack(None, b2, cn, None, 0, 6, 448, bin1_offset=bad.astype(np.int64))
    hb.close()


def test_compute_from_pixel_table_equals_dense_route(tmp_path, monkeypatch):
    names = ['chrA', 'chrB']
    chroms = {'chrA': synth.SynthChrom(1500, 51, stripe_every=90, stripe_gain=3.0),
              'chrB': synth.SynthChrom(1100, 52, stripe_every=90, stripe_gain=3.0)}
    t = pixels.PixelTable.from_synth(names, chroms, RESOL)
    p = str(tmp_path / 't.npz')
    t.save(p)

    class FetchOnly:
        def __init__(self, sel):
            self._sel = sel

        def fetch(self, *a):
            return self._sel.fetch(*a)

    outs = []
    for route in ('pixels', 'dense'):
        if route == 'dense':
            monkeypatch.setattr(stripenn, 'open_matrix', lambda cool: sio.MatrixInfo(
                t.chromnames, t.chromsizes, t.binsize, ['chrom', 'start', 'end', 'weight'],
                lambda balance: FetchOnly(pixels.PixelSelector(t, balance))))
        out = str(tmp_path / route)
        stripenn.compute('pixels:' + p, out, 'weight', 'all', 2.0, 10, 8, '0.95,0.97,0.99', 1, 0.5, '0', False, 3, 7,
                         force=True)
        outs.append([open(os.path.join(out, f)).read() for f in ('result_unfiltered.tsv', 'result_filtered.tsv')])
    assert outs[0] == outs[1]
    assert outs[0][0].count('\n') > 10


def test_quantile_from_pixel_table_equals_dense_quantile():
    """stp_select_append_pixels: values formed on the device, off-diagonal pixels counted twice, equals
    np.quantile over the positive entries of the dense symmetric matrix (getStripe.py:160-176)."""
    import pandas as pd
    from stripenn_amd import getStripe as GS
    names = ['chrA', 'chrB']
    chroms = {'chrA': synth.SynthChrom(900, 41, nan_frac=0.03), 'chrB': synth.SynthChrom(700, 42, balanced=False)}
    t = pixels.PixelTable.from_synth(names, chroms, RESOL)
    qs = [0.5, 0.95, 0.97, 0.999]
    for balance in ('weight', False):
        sel = pixels.PixelSelector(t, balance)

        class Info:
            chromsizes = pd.Series(t.chromsizes, index=names)
            binsize = RESOL
        hb = BK.HipBackend(0)
        obj = GS.getStripe(sel, RESOL, 10, 8, 2.0, names, names, t.chromsizes, t.chromsizes, 1, 3, 1, backend=hb)
        MP = obj.getQuantile_original(Info, names, qs)
        hb.close()
        for nm in names:
            D = sel.fetch(nm)
            assert np.array_equal(MP[nm], np.quantile(D[D > 0], qs)), (balance, nm)


def test_compute_from_a_cooler_file_on_the_device(tmp_path, monkeypatch):
    """`stripenn compute tests/golden/cool_tiny.mcool::resolutions/5000` -- the committed cooler-schema file read by the
    package's own HDF5 reader (stripenn_amd/h5lite.py: this interpreter has neither cooler nor h5py), its columns packed
    into bands on the device, default (HIP) backend -- writes the TSVs that the oracle backend writes for the same genome
    held as an in-memory table (stripenn.py:80-118 + the whole driver), byte for byte."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cool_fixture as CF
    from oracle import oracle as O
    from oracle_backend import OracleBackend
    names, chroms, t = CF.table(small=True)
    fixture = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cool_tiny.mcool') + '::' + CF.GROUP
    monkeypatch.setitem(sys.modules, 'cooler', None)
    info = sio.open_matrix(fixture)
    assert list(info.chromnames) == names
    outs = []
    for src in ('file+hip', 'table+oracle'):
        out = str(tmp_path / src.replace('+', '_'))
        if src == 'table+oracle':
            monkeypatch.setattr(stripenn, 'open_matrix', lambda cool: sio.pixel_matrix(t))
            stripenn.compute(fixture, out, 'KR', 'all', 2.0, 10, 8, '0.97,0.99', 2, 0.5, '0', False, 3, 7, force=True,
                             backend=OracleBackend(gauss_w=O.gauss_weights(2.0)[0]))
        else:
            stripenn.compute(fixture, out, 'KR', 'all', 2.0, 10, 8, '0.97,0.99', 2, 0.5, '0', False, 3, 7, force=True)
        outs.append([open(os.path.join(out, f)).read() for f in ('result_unfiltered.tsv', 'result_filtered.tsv')])
    assert outs[0] == outs[1] and outs[0][0].count('\n') > 5


def test_downloads_into_recycled_host_addresses(hip_ctx):
    """Large pageable uploads followed by large downloads into freshly allocated arrays -- numpy hands the freed upload
    buffers' addresses out again.  The HIP runtime keeps the pins it makes for pageable transfers in a cache keyed by the
    host address, and a pin made for an upload is read-only to the device: before round 4 such a download could die with
    "write access to a read-only page" (three aborted runs; tools/soak_misc.py seed 489).  The library now stages or
    registers every host buffer itself (stp_xfer), the runtime never sees the caller's pointers."""
    hb = BK.HipBackend(0)
    rng = np.random.default_rng(12)
    ch = synth.SynthChrom(1500, 77)
    band_h = ch.band(512)
    for it in range(12):
        sel = hb.select_open()
        vals = rng.random(int(rng.integers(900_000, 1_600_000))) + 0.5          # 7-13 MB, a new buffer every round
        hb.select_append(sel, vals)
        n = hb.select_count(sel)
        assert n == len(vals)
        k = int(rng.integers(0, n))
        assert hb.select_ranks(sel, [k])[0] == np.partition(vals, k)[k]
        hb.select_close(sel)
        del vals
        band = hb.open_chrom(band_h)
        out = band.download()                                                   # 12 MB into a fresh array
        assert np.array_equal(out, band_h, equal_nan=True)
        del out
        r, l = hb.diag_sums(band)
        band.close()
    hb.close()
