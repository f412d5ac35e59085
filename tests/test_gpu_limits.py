"""The hard limits round 1 answered with an error, now handled like the reference handles them:
more record slots than the sweep grants an image (chunk re-run), score stripes longer than 1 024 bins / wider
than 256, background windows of more than 8 192 pixels (bins below 556 bp), band halfwidth derived from the
resolution / the longest stripe.  Every case against the oracle."""
import numpy as np
import pandas as pd
import pytest

from oracle import oracle as O
from oracle_backend import OracleBackend

pytestmark = pytest.mark.gpu
HW = 512


def _dense_lines(S=400, period=3, seed=0):
    """Dense frame crowded with thin lines through the diagonal: ~100 candidate stripes per image (the most a 400-bin
    image could be made to produce; the sweep's 128 slots were never exceeded on any data)."""
    rng = np.random.default_rng(seed)
    D = np.full((S, S), 2.0) + rng.random((S, S)) * 0.2
    for c in range(5, S - 5, period):
        L = 60 + (c * 7) % 100
        D[c:min(S, c + L), c] += 6; D[c, c:min(S, c + L)] += 6
        D[max(0, c - L):c, c] += 6; D[c, max(0, c - L):c] += 6
    return (D + D.T) / 2


def _band_of(D, hw=HW):
    S = D.shape[0]
    band = np.zeros((S, 2 * hw))
    rr = np.arange(S)[:, None]
    cc = rr + np.arange(-hw, hw)[None, :]
    ok = (cc >= 0) & (cc < S)
    band[ok] = D[np.broadcast_to(rr, cc.shape)[ok], cc[ok]]
    return band


def test_record_slot_overflow_reruns_the_chunk(hip_ctx):
    """An image that needs more record slots than the first pass grants: the chunk is searched again with one slot
    per possible column pair and the records equal the oracle's (here the first pass is shrunk to 8 slots through
    the debugging hook, so ordinary crowded images take the re-run path; sigma 1.0 = generic-radius Canny kernel)."""
    D = _dense_lines()
    M = float(np.quantile(D[D > 0], 0.95))
    band = hip_ctx.band_upload(_band_of(D))
    fr = band.frames([0], [D.shape[0] - 1])
    exp_r, exp_t = O.stripe_search(D, M, sigma=1.0, bf=1)
    per_image = np.bincount(exp_r[:, 0], minlength=6)
    assert per_image.max() > 64
    want = [tuple(int(v) for v in q) + (float(t),) for q, t in zip(exp_r, exp_t)]
    for slots in (128, 8):
        hip_ctx.dbg_set_sweep_slots(slots)
        try:
            recs = fr.stripe_search([M], sigma=1.0, bfilter=1)
        finally:
            hip_ctx.dbg_set_sweep_slots(128)
        got = [(int(r['b_index']), int(r['ud']), int(r['x']), int(r['y']), int(r['w']), int(r['h']), float(r['total']))
               for r in recs]
        assert got == want, 'slots %d' % slots
    fr.close(); band.close()


def test_long_and_wide_score_stripes():
    """Stripes of 2 000 rows and 600 columns (round 1 refused > 1 024 x 256): p-value, Stripiness, means vs oracle.py;
    the facade widens the band's halfwidth for them (score._halfwidth_for)."""
    from stripenn_amd import synth, getStripe as GS, score
    from stripenn_amd.backend import HipBackend
    resol = 5000
    nb = 3000
    ch = synth.SynthChrom(nb, 9, nan_frac=0.0)     # (a NaN bin inside a 600-column block would raise the reference's own IndexError)
    names, sizes = ['chr1'], np.array([nb * resol])
    sel = synth.SynthSelector({'chr1': ch}, resol)
    table = pd.DataFrame({'chr': ['chr1'] * 4,
                          'pos1': [500 * resol + 1, 1400 * resol + 1, 100 * resol + 1, 2000 * resol + 1],
                          'pos2': [520 * resol, 2000 * resol, 130 * resol, 2010 * resol],
                          'chr2': ['chr1'] * 4,
                          'pos3': [500 * resol + 1, 1400 * resol + 1, 100 * resol + 1, 800 * resol + 1],
                          'pos4': [2500 * resol, 1500 * resol, 1300 * resol, 2010 * resol]})
    hw = score._halfwidth_for(table, resol)
    assert hw is not None and hw > 2000 and hw % 64 == 0
    out = {}
    for tag, be in (('hip', HipBackend(0)), ('ora', OracleBackend())):
        obj = GS.getStripe(sel, resol, 10, 8, 2.5, names, names, sizes, sizes, 2, 1, 7, backend=be, halfwidth=hw)
        EV = obj.mpmean()
        bg = obj.nulldist()
        out[tag] = (obj.pvalue(*bg, table), obj.getMean(table), obj.scoringstripes(table, EV, '0'))
        be.close()
    assert out['hip'][0] == out['ora'][0]                                             # p-values: exact
    assert np.array_equal(out['hip'][2][0], out['ora'][2][0], equal_nan=True)         # Stripiness: exact
    for a, b in ((out['hip'][1][0], out['ora'][1][0]), (out['hip'][1][1], out['ora'][1][1]),
                 (out['hip'][2][1], out['ora'][2][1]), (out['hip'][2][2], out['ora'][2][2])):
        assert np.allclose(np.asarray(a, float), np.asarray(b, float), rtol=1e-9, atol=0, equal_nan=True)


def test_background_windows_beyond_8192_pixels():
    """500 bp bins: bs = 100, every window mean covers 10 000 pixels -- numpy reduces them buffer by buffer (8 192
    elements); the band's halfwidth follows the resolution (448 + 2 bs -> 704).  All four background tables of a
    small chromosome, HIP vs oracle backend, bit for bit (band-served and unit-matrix batches)."""
    from stripenn_amd import synth, getStripe as GS
    from stripenn_amd.backend import HipBackend
    resol = 500
    nb = 2600
    ch = synth.SynthChrom(nb, 21, nan_frac=0.01)
    names, sizes = ['chr1'], np.array([nb * resol])
    sel = synth.SynthSelector({'chr1': ch}, resol)
    tabs = {}
    for tag, be in (('hip', HipBackend(0)), ('ora', OracleBackend())):
        obj = GS.getStripe(sel, resol, 10, 8, 2.0, names, names, sizes, sizes, 1, 3, 11, backend=be)
        assert obj.halfwidth == 704
        tabs[tag] = obj.nulldist()
        be.close()
    for a, b in zip(tabs['hip'], tabs['ora']):
        assert a.shape == b.shape and a.shape[0] == 400 and a.shape[1] > 100
        assert np.array_equal(a, b, equal_nan=True)
