"""k_score_wave (one wave per stripe, round 5) against the block kernels it replaces for short stripes (STP_SCORE=block),
against the fused entry point stp_score, and against the oracle's per-stripe restatements: random stripe rectangles with
masked columns / rows, NaN-ridden and unsorted background tables, inherited background rows, stripes on both sides of the
wave form's limits (192 rows, 64 columns per block)."""
import os

import numpy as np
import pytest

from oracle_backend import OracleBackend

pytestmark = pytest.mark.gpu


def _stripes(rng, nb, bs, n, hmax):
    from stripenn_amd import backend as BK
    w = rng.integers(1, 12, n); h = rng.integers(11, hmax, n)
    x0 = rng.integers(bs + 1, nb - 400 - bs - 14, n); x1 = x0 + w - 1
    down = rng.random(n) < 0.5
    y0 = np.where(down, x0, np.maximum(x1 + 1 - h, 1)); y1 = y0 + h - 1
    y1 = np.minimum(y1, nb - 2); y0 = np.minimum(y0, y1 - 10)
    pv = np.zeros(n, dtype=BK.PV_STRIPE_DTYPE)
    pv['row0'], pv['row1'] = y0, y1 + 1
    pv['col0'], pv['col1'] = np.maximum(x0 - bs, 0), np.minimum(x1 + 1 + bs, nb)
    pv['mode'] = np.where(down, 0, 1); pv['upbase'] = y1 + 1 - y0
    gen = rng.random(n) < 0.15
    pv['mode'][gen] = 2; pv['fixed_row'][gen] = rng.integers(0, 400, int(gen.sum())); pv['fixed_tab'][gen] = rng.integers(0, 2, int(gen.sum()))
    sc = np.zeros(n, dtype=BK.SCORE_STRIPE_DTYPE)
    sc['row0'], sc['row1'] = y0, y1 + 1
    lm = np.minimum(np.maximum(x0 - bs, 1), x0); rm = np.minimum(x1 + 1 + bs, nb - 1)
    sc['col0'][:, 0], sc['col1'][:, 0] = x0, x1 + 1
    sc['col0'][:, 1], sc['col1'][:, 1] = lm, x0
    sc['col0'][:, 2], sc['col1'][:, 2] = x1 + 1, np.maximum(rm, x1 + 1)
    sc['ex0'][:, 0], sc['ex0'][:, 1], sc['ex0'][:, 2] = x0, lm, x1 + 2
    sc['ey0'] = y0; sc['mirror'] = np.where(x0 == y0, 0, 1)
    sc['mcol0'], sc['mcol1'], sc['mrow0'], sc['mrow1'] = 1, 0, 1, 0
    msk = rng.random(n) < 0.15
    sc['mcol0'][msk, 0] = 0; sc['mcol1'][msk, 0] = 0; sc['mrow0'][msk] = 2; sc['mrow1'][msk] = 3
    return pv, sc


def _same(a, b, exact):
    return np.array_equal(a, b, equal_nan=True) if exact else np.allclose(a, b, rtol=1e-9, atol=0, equal_nan=True)


@pytest.mark.parametrize('seed,resol,hmax', [(1, 5000, 190), (2, 5000, 380), (3, 10000, 120), (4, 2000, 260), (5, 1000, 150)])
def test_wave_form_equals_block_form_fused_call_and_oracle(seed, resol, hmax):
    from oracle import oracle as O
    from stripenn_amd import backend as BK, synth
    O.build()
    rng = np.random.default_rng(100 + seed)
    nb = int(rng.integers(1200, 2200))
    bs = int(50000 / resol)
    ch = synth.SynthChrom(nb, 700 + seed, nan_frac=0.01 if seed % 2 else 0.0, balanced=bool(seed % 2))
    band_h = ch.band(512)
    hb = BK.HipBackend(0); ob = OracleBackend()
    gb = hb.open_chrom(band_h); cb = ob.open_chrom(band_h)
    ncol = int(rng.integers(300, 1000))
    bg = [np.sort(rng.normal(0, 3, (400, ncol)), axis=1) if k % 2 else rng.normal(0, 3, (400, ncol)) for k in range(4)]
    for t in bg:
        t[rng.random(t.shape) < 0.01] = np.nan
    hb.set_background(*bg); ob.set_background(*bg)
    EV = 240.0 / (1.0 + np.arange(400)) + 1.0 + rng.random(400)
    n = 400
    pv, sc = _stripes(rng, nb, bs, n, hmax)

    def run():
        p = hb.pvalue(gb, bs, pv)
        try:
            s = hb.stripiness(gb, EV, sc)
        except IndexError:
            s = None
        try:
            f = hb.score(gb, bs, EV, pv, sc)
        except IndexError:
            f = None
        return p, s, f
    p, s, f = run()
    os.environ['STP_SCORE'] = 'block'
    try:
        pb, sb, fb = run()
    finally:
        os.environ.pop('STP_SCORE', None)
    assert _same(p, pb, True), 'p-values: wave form vs block form'
    assert (s is None) == (sb is None) == (f is None) == (fb is None), 'IndexError behaviour differs between the forms'
    if s is not None:
        for a, b, nm in zip(s, sb, 'gmt'):
            assert _same(a, b, nm == 'g'), 'Stripiness output %s: wave form vs block form' % nm
        assert _same(f[0], p, True) and all(_same(a, b, True) for a, b in zip(f[1:], s)), 'fused call vs separate calls'
        assert _same(fb[0], pb, True) and all(_same(a, b, True) for a, b in zip(fb[1:], sb))
    # the oracle on a third of the stripes (per-stripe Python)
    idx = np.arange(seed % 3, n, 3)
    assert _same(p[idx], ob.pvalue(cb, bs, pv[idx]), True), 'p-values vs oracle'
    try:
        so = ob.stripiness(cb, EV, sc[idx])
    except IndexError:
        so = None
    if so is not None and s is not None:
        for a, b, nm in zip(s, so, 'gmt'):
            assert _same(a[idx], b, nm == 'g'), 'Stripiness output %s vs oracle' % nm
    gb.close()
    hb.close()


@pytest.mark.parametrize('kind', ['zero', 'negative', 'inf', 'nan'])
def test_expected_values_that_are_not_finite_and_positive(kind):
    """The wave kernel takes "observed / expected is NaN exactly where observed is" only when every expected value + 1e-8 is
    finite and > 0 (tested once per workgroup); a table with a zero, a negative, an infinite or a NaN entry takes the general
    path (the quotient itself is tested): wave form == block form == oracle."""
    from oracle import oracle as O
    from stripenn_amd import backend as BK, synth
    O.build()
    rng = np.random.default_rng(77)
    nb, bs = 1500, 10
    ch = synth.SynthChrom(nb, 811, nan_frac=0.01)
    band_h = ch.band(512)
    hb = BK.HipBackend(0); ob = OracleBackend()
    gb = hb.open_chrom(band_h); cb = ob.open_chrom(band_h)
    EV = 240.0 / (1.0 + np.arange(400)) + 1.0 + rng.random(400)
    bad = rng.choice(60, 12, replace=False)                      # near-diagonal entries: every stripe meets some of them
    EV[bad] = {'zero': -1e-8, 'negative': -3.0, 'inf': np.inf, 'nan': np.nan}[kind]
    pv, sc = _stripes(rng, nb, bs, 240, 150)
    with np.errstate(all='ignore'):
        try:
            s = hb.stripiness(gb, EV, sc)
        except IndexError:
            s = None
        os.environ['STP_SCORE'] = 'block'
        try:
            try:
                sb = hb.stripiness(gb, EV, sc)
            except IndexError:
                sb = None
        finally:
            os.environ.pop('STP_SCORE', None)
        assert (s is None) == (sb is None)
        idx = np.arange(0, 240, 2)
        try:
            so = ob.stripiness(cb, EV, sc[idx])
        except IndexError:
            so = None
    if s is not None:
        for a, b, nm in zip(s, sb, 'gmt'):
            assert _same(a, b, nm == 'g'), 'Stripiness output %s: wave form vs block form (%s)' % (nm, kind)
        if so is not None:
            for a, b, nm in zip(s, so, 'gmt'):
                assert _same(a[idx], b, nm == 'g'), 'Stripiness output %s vs oracle (%s)' % (nm, kind)
    gb.close()
    hb.close()


def test_symmetric_reads_are_used_only_for_symmetric_bands():
    """The score kernels read a stripe's pixels as M[c][r] (coalesced) only after k_band_symcheck has found the band bit for bit
    symmetric.  (i) A symmetric band: the symmetric reads and the row reads (STP_SCORE_NOSYM=1) give identical outputs, and the
    oracle's.  (ii) A band with ONE asymmetric cell pair far from every stripe, and one whose upper triangle is perturbed
    everywhere: the library must fall back to the row reads -- outputs equal the oracle's, which reads M[r][c] from row r."""
    from oracle import oracle as O
    from stripenn_amd import backend as BK, synth
    O.build()
    rng = np.random.default_rng(77)
    nb, bs = 1500, 10
    ch = synth.SynthChrom(nb, 777, nan_frac=0.01)
    sym = ch.band(512)
    one = sym.copy(); one[1400, 512 + 7] += 1.0                    # M[1400][1407] != M[1407][1400]
    allp = sym.copy(); allp[:, 513:] *= 1.0 + 1e-3 * rng.random((nb, 511))
    ncol = 500
    bg = [np.sort(rng.normal(0, 3, (400, ncol)), axis=1) for _ in range(4)]
    EV = 240.0 / (1.0 + np.arange(400)) + 1.0 + rng.random(400)
    pv, sc = _stripes(rng, nb, bs, 300, 190)
    keep = pv['row1'] < 1300                                       # (the single asymmetric pair stays outside every patch)
    pv, sc = pv[keep], sc[keep]
    hb = BK.HipBackend(0); ob = OracleBackend()
    hb.set_background(*bg); ob.set_background(*bg)
    idx = np.arange(0, len(pv), 2)
    for name, band_h in (('symmetric', sym), ('one cell pair', one), ('upper triangle perturbed', allp)):
        gb = hb.open_chrom(band_h); cb = ob.open_chrom(band_h)
        p, g, m, t = hb.score(gb, bs, EV, pv, sc)
        os.environ['STP_SCORE_NOSYM'] = '1'
        try:
            p2, g2, m2, t2 = hb.score(gb, bs, EV, pv, sc)
        finally:
            os.environ.pop('STP_SCORE_NOSYM', None)
        assert _same(p, p2, True) and _same(g, g2, True) and _same(m, m2, True) and _same(t, t2, True), name
        assert _same(p[idx], ob.pvalue(cb, bs, pv[idx]), True), name + ': p-values vs oracle'
        eg, em, et = ob.stripiness(cb, EV, sc[idx])
        assert _same(g[idx], eg, True) and _same(m[idx], em, False) and _same(t[idx], et, False), name + ': Stripiness vs oracle'
        gb.close()
    hb.close()


def test_identical_stripes_are_scored_once_with_the_same_results():
    """stp_score scores byte-identical rows once and copies the results (round 6: 59 % of a search's candidates repeat another row).
    A list in which every stripe occurs 1-5 times in random order gives, row by row, what the separate calls give -- which never
    merge rows -- and the repeats of a row are equal among themselves; a list without repeats takes the plain path."""
    from stripenn_amd import backend as BK, synth
    rng = np.random.default_rng(77)
    nb, bs = 1800, 10
    ch = synth.SynthChrom(nb, 901, nan_frac=0.01)
    band_h = ch.band(512)
    hb = BK.HipBackend(0)
    gb = hb.open_chrom(band_h)
    bg = [np.sort(rng.normal(0, 3, (400, 700)), axis=1) for _ in range(4)]
    hb.set_background(*bg)
    EV = 240.0 / (1.0 + np.arange(400)) + 1.0 + rng.random(400)
    pv0, sc0 = _stripes(rng, nb, bs, 300, 190)
    ok = np.ones(len(pv0), bool)
    try:
        hb.stripiness(gb, EV, sc0)
    except IndexError:                      # (keep only stripes the reference would not raise on: one by one)
        for i in range(len(pv0)):
            try:
                hb.stripiness(gb, EV, sc0[i:i + 1])
            except IndexError:
                ok[i] = False
    pv0, sc0 = pv0[ok], sc0[ok]
    reps = rng.integers(1, 6, len(pv0))
    idx = np.repeat(np.arange(len(pv0)), reps)
    rng.shuffle(idx)
    pv, sc = pv0[idx], sc0[idx]
    assert len(pv) > 600
    p, g, m, t = hb.score(gb, bs, EV, pv, sc)
    p1 = hb.pvalue(gb, bs, pv)
    g1, m1, t1 = hb.stripiness(gb, EV, sc)
    assert _same(p, p1, True) and _same(g, g1, True) and _same(m, m1, True) and _same(t, t1, True)
    pu, gu, mu, tu = hb.score(gb, bs, EV, pv0, sc0)              # (no repeats)
    assert _same(p, pu[idx], True) and _same(g, gu[idx], True) and _same(m, mu[idx], True) and _same(t, tu[idx], True)
    hb.close()
