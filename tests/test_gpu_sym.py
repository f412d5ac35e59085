"""Image symmetry in k_canny_f32 (round 6): a frame's images equal their transposes up to the order of the reference's roundings,
so the tiles strictly below the diagonal are not computed -- their class words are the transposes of the tiles above, the
undecidable pixels settled once per position.  STP_SYM=0 switches it off (every tile computed); STP_SYM=report-all marks every
image as one whose grey values differ from their mirror image (the path such an image takes: its tiles below the diagonal
through the exact kernel).  All three must give the oracle's class maps and identical records; a band that is not symmetric
must never be mirrored."""
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


class _sym:
    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.old = os.environ.get('STP_SYM')
        if self.mode is None:
            os.environ.pop('STP_SYM', None)
        else:
            os.environ['STP_SYM'] = self.mode

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop('STP_SYM', None)
        else:
            os.environ['STP_SYM'] = self.old


def _band_of(dense, hw=512):
    n = dense.shape[0]
    band = np.zeros((n, 2 * hw))
    for i in range(n):
        lo, hi = max(0, i - hw), min(n, i + hw)
        band[i, lo - i + hw:hi - i + hw] = dense[i, lo:hi]
    return band


@pytest.mark.parametrize('sigma', [2.0, 2.5, 1.5])
def test_class_maps_mirrored_computed_and_oracle(hip_ctx, sigma):
    """Frames of every size class (a first frame of 300 bins, full frames, a short last frame; NaN bins shift the tile grid
    against the diagonal blocks), three maxpixel levels, every brightness image: mirrored == every tile computed ==
    report-all == oracle (sampled)."""
    from stripenn_amd import synth, hip
    ch = synth.SynthChrom(2530, 41, stripe_every=60, stripe_gain=3.0, nan_frac=0.01)
    band = hip_ctx.band_upload(ch.band(512))
    st = np.array([0, 300, 900, 1500, 2100, 2300]); en = np.array([299, 699, 1299, 1899, 2499, 2529])
    fr = band.frames(st, en)
    blk = ch.block(0, 2530, 0, 2530)
    Ms = np.quantile(blk[blk > 0], [0.9, 0.97, 0.995])
    gw, gr = hip.gauss_weights(sigma)
    nimg = 0
    for f in range(len(st)):
        D, nz = O.frame_dense(ch.block, int(st[f]), int(en[f]))
        D = np.ascontiguousarray(D[np.ix_(nz, nz)])
        for M in Ms:
            gp = O.gplane(D, float(M))
            for bi in range(6):
                with _sym(None):
                    a = fr.dbg_stages(f, float(M), bi, sigma=sigma)
                with _sym('0'):
                    b = fr.dbg_stages(f, float(M), bi, sigma=sigma)
                assert np.array_equal(a['cls'], b['cls']), (sigma, f, M, bi)
                assert np.array_equal(a['edges'], b['edges'])
                if bi in (1, 4):
                    with _sym('report-all'):
                        c = fr.dbg_stages(f, float(M), bi, sigma=sigma)
                    assert np.array_equal(a['cls'], c['cls']), (sigma, f, M, bi, 'report-all')
                if bi in (0, 5):
                    _, dbg = O.canny(O.gray(gp, O.brightness_levels()[bi], 3), gw, gr, debug=True)
                    assert np.array_equal(a['cls'], dbg['cls']), (sigma, f, M, bi)
                nimg += 1
    assert nimg == 108
    fr.close(); band.close()


def test_sweep_records_identical_with_and_without_mirroring(hip_ctx):
    """The chr16-size five-level sweep: every record byte for byte, mirrored / every tile computed / report-all."""
    from stripenn_amd import synth
    nb = 19642
    ch = synth.SynthChrom(nb, 16)
    band_h = ch.band(512)
    band = hip_ctx.band_upload(band_h)
    nfr = -(-nb // 200)
    st = np.array([max(0, i * 200 - 100) for i in range(nfr)]); en = np.minimum((np.arange(nfr) + 1) * 200 + 99, nb - 1)
    M = np.quantile(band_h[band_h > 0], [0.95, 0.96, 0.97, 0.98, 0.99])
    fr = band.frames(st, en)
    with _sym(None):
        a = fr.stripe_search(M)
    with _sym('0'):
        b = fr.stripe_search(M)
    with _sym('report-all'):
        c = fr.stripe_search(M)
    assert len(a) == len(b) == len(c) > 10000
    assert a.tobytes() == b.tobytes() == c.tobytes()
    fr.close(); band.close()


def test_plateaus_on_and_across_the_diagonal(hip_ctx):
    """Symmetric block / stair maps: exact ties along straight edges overflow the tiles' lists, so the flagged tile-images AND the
    tiles below the diagonal that would have received their transposes go to the exact kernel."""
    n = 800
    rr, cc = np.mgrid[0:n, 0:n].astype(np.float64)
    cases = {
        'blocks': np.where(((cc // 37) + (rr // 37)) % 2 == 0, 12.0, 3.0) + np.where(np.abs(cc - rr) < 25, 20.0, 0.0),
        'bands': 2.0 + 2.0 * ((np.abs(cc - rr) // 16) % 5),
        'cross': np.where((cc % 90 < 45) ^ (rr % 90 < 45), 9.0, 2.5),
    }
    gw, gr = O.gauss_weights(2.0)
    for name, dense in cases.items():
        dense = np.where(np.abs(cc - rr) <= 500, dense, 0.0)
        assert np.array_equal(dense, dense.T)
        band = hip_ctx.band_upload(_band_of(dense))
        fr = band.frames([0, 250], [399, 649])
        for f, (s, e) in enumerate(((0, 399), (250, 649))):
            D = np.ascontiguousarray(dense[s:e + 1, s:e + 1])
            for M in (float(np.quantile(D[D > 0], 0.9)), float(D.max())):
                gp = O.gplane(D, M)
                for bi in (0, 3, 5):
                    with _sym(None):
                        a = fr.dbg_stages(f, M, bi)
                    with _sym('0'):
                        b = fr.dbg_stages(f, M, bi)
                    _, dbg = O.canny(O.gray(gp, O.brightness_levels()[bi], 3), gw, gr, debug=True)
                    assert np.array_equal(a['cls'], dbg['cls']), (name, f, M, bi)
                    assert np.array_equal(b['cls'], dbg['cls']), (name, f, M, bi, 'every tile computed')
        # the search itself (the debug entry point above keeps whole grey images; the search does not write the grey tiles no
        # computed Canny tile reads, and fills them in for the (frame, level) pairs whose tiles below the diagonal go to the
        # exact kernel: k_gray_fill): records with the measures on, off, and the oracle's
        Ms = np.array([float(np.quantile(dense[dense > 0], 0.9)), float(dense.max())])
        with _sym(None):
            ra = fr.stripe_search(Ms)
        os.environ['STP_REUSE'] = '0'
        try:
            with _sym('0'):
                rb = fr.stripe_search(Ms)
        finally:
            os.environ.pop('STP_REUSE', None)
        assert ra.tobytes() == rb.tobytes(), name
        exp = []
        for f, (s, e) in enumerate(((0, 399), (250, 649))):
            D = np.ascontiguousarray(dense[s:e + 1, s:e + 1])
            for li, M in enumerate(Ms):
                r, t = O.stripe_search(D, float(M), gw=gw)
                exp += [(f, li) + tuple(int(v) for v in q) for q in r]
        got = [tuple(int(r[k]) for k in ('frame', 'level', 'b_index', 'ud', 'x', 'y', 'w', 'h')) for r in ra]
        assert got == exp, name
        fr.close(); band.close()


def test_a_band_that_is_not_symmetric_is_never_mirrored(hip_ctx):
    """The ABI takes any band.  A contact map whose two triangles differ (here: different noise and a step that exists on one
    side only) must come out as the oracle's class maps of that very matrix -- with mirroring its lower triangle would be
    the transpose of the upper one."""
    n = 700
    rng = np.random.default_rng(12)
    rr, cc = np.mgrid[0:n, 0:n].astype(np.float64)
    base = 240.0 / (1.0 + np.abs(rr - cc)) + 1.0
    dense = np.round(base + np.sqrt(base) * rng.standard_normal((n, n))).clip(0) / 4.0
    dense += np.where((cc - rr > 40) & (cc % 120 < 4), 6.0, 0.0)                      # stripes above the diagonal only
    dense = np.where(np.abs(cc - rr) <= 500, dense, 0.0)
    assert not np.array_equal(dense, dense.T)
    band = hip_ctx.band_upload(_band_of(dense))
    fr = band.frames([0, 300], [399, 699])
    gw, gr = O.gauss_weights(2.0)
    n_asym = 0
    for f, (s, e) in enumerate(((0, 399), (300, 699))):
        D = np.ascontiguousarray(dense[s:e + 1, s:e + 1])
        for M in (float(np.quantile(D[D > 0], 0.95)), float(np.quantile(D[D > 0], 0.99))):
            gp = O.gplane(D, M)
            for bi in (0, 2, 5):
                a = fr.dbg_stages(f, M, bi)
                _, dbg = O.canny(O.gray(gp, O.brightness_levels()[bi], 3), gw, gr, debug=True)
                assert np.array_equal(a['cls'], dbg['cls']), (f, M, bi)
                n_asym += int(not np.array_equal(dbg['cls'], dbg['cls'].T))
    assert n_asym > 0                                          # the test has teeth: the oracle's class maps are not symmetric
    # ... and the records of the whole chain
    Ms = [float(np.quantile(dense[dense > 0], q)) for q in (0.95, 0.99)]
    got = fr.stripe_search(np.array(Ms))
    exp = []
    for f, (s, e) in enumerate(((0, 399), (300, 699))):
        D = np.ascontiguousarray(dense[s:e + 1, s:e + 1])
        for li, M in enumerate(Ms):
            r, t = O.stripe_search(D, M, gw=gw)
            exp += [(f, li) + tuple(int(v) for v in q) for q in r]
    gotl = [tuple(int(r[k]) for k in ('frame', 'level', 'b_index', 'ud', 'x', 'y', 'w', 'h')) for r in got]
    assert sorted(gotl) == sorted(exp) and len(exp) > 0
    fr.close(); band.close()


def test_skipped_grey_tiles_are_never_read(hip_ctx):
    """With the symmetry in use k_gray_c3 does not write the 20 grey tiles no computed Canny tile reads (the far corner below the
    diagonal).  STP_TEST_POISON_GRAY=1 fills the grey images and the cell table with NaN patterns before every launch: a read of
    an unwritten value would reach the class maps.  Ordinary data (the resolver's transposed reads), plateaus (list overflow ->
    the tiles below the diagonal through the exact kernel, their grey tiles filled in by k_gray_fill) and report-all (every pair
    filled in) give the records of the plain search."""
    from stripenn_amd import synth
    ch = synth.SynthChrom(3100, 53, stripe_every=70, stripe_gain=3.0, nan_frac=0.008)
    band_h = ch.band(512)
    n = 800
    rr, cc = np.mgrid[0:n, 0:n].astype(np.float64)
    blocks = np.where(((cc // 37) + (rr // 37)) % 2 == 0, 12.0, 3.0) + np.where(np.abs(cc - rr) < 25, 20.0, 0.0)
    blocks = np.where(np.abs(cc - rr) <= 500, blocks, 0.0)
    for name, bh in (('noisy', band_h), ('blocks', _band_of(blocks))):
        nb = bh.shape[0]
        band = hip_ctx.band_upload(bh)
        nfr = -(-nb // 200)
        st = np.array([max(0, i * 200 - 100) for i in range(nfr)]); en = np.minimum((np.arange(nfr) + 1) * 200 + 99, nb - 1)
        fr = band.frames(st, en)
        Ms = np.quantile(bh[bh > 0], [0.9, 0.99])
        os.environ['STP_REUSE'] = '0'; os.environ['STP_SYM'] = '0'
        try:
            ref = fr.stripe_search(Ms)
        finally:
            os.environ.pop('STP_REUSE', None); os.environ.pop('STP_SYM', None)
        assert len(ref) > 20
        os.environ['STP_TEST_POISON_GRAY'] = '1'
        try:
            for mode in (None, 'report-all'):
                with _sym(mode):
                    got = fr.stripe_search(Ms)
                assert got.tobytes() == ref.tobytes(), (name, mode)
        finally:
            os.environ.pop('STP_TEST_POISON_GRAY', None)
        fr.close(); band.close()
