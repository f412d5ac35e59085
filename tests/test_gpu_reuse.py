"""Frame overlap (round 6): the trailing block of frame f is the leading block of frame f + 1 (getStripe.py:794-799).  Where both
compacted frames keep the same bins of it, k_canny_f32 does not compute the tiles inside the block's interior (Gaussian
radius + 3 from its border) for frame f: k_lines takes those class words from frame f + 1's planes.  STP_REUSE=0 computes
everything.  Records must be identical either way, with and without the image symmetry, and equal to the oracle's; a pair of
frames that do not keep the same bins must fall back."""
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


class _env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _frame_table(nb):
    nfr = -(-nb // 200)
    st = np.array([max(0, i * 200 - 100) for i in range(nfr)])
    en = np.minimum((np.arange(nfr) + 1) * 200 + 99, nb - 1)
    return st, en


def _band_of(dense, hw=512):
    n = dense.shape[0]
    band = np.zeros((n, 2 * hw))
    for i in range(n):
        lo, hi = max(0, i - hw), min(n, i + hw)
        band[i, lo - i + hw:hi - i + hw] = dense[i, lo:hi]
    return band


def _oracle_records(block, st, en, Ms, gw):
    exp = []
    for f in range(len(st)):
        D, nz = O.frame_dense(block, int(st[f]), int(en[f]))
        if len(nz) <= 10:
            continue
        D = np.ascontiguousarray(D[np.ix_(nz, nz)])
        for li, M in enumerate(Ms):
            r, t = O.stripe_search(D, float(M), gw=gw)
            exp += [(f, li) + tuple(int(v) for v in q) + (float(tt),) for q, tt in zip(r, t)]
    return exp


def _as_tuples(recs):
    return [tuple(int(r[k]) for k in ('frame', 'level', 'b_index', 'ud', 'x', 'y', 'w', 'h')) + (float(r['total']),) for r in recs]


@pytest.mark.parametrize('sigma', [2.0, 2.5])
def test_sweep_records_with_and_without_the_overlap(hip_ctx, sigma):
    """A chromosome of 17 frames with NaN bins (the compacted frames differ in size, the shared blocks start at different
    indices): all four kernel selections give the same bytes, and the records are the oracle's."""
    from stripenn_amd import synth, hip
    nb = 3333
    ch = synth.SynthChrom(nb, 23, stripe_every=90, stripe_gain=3.0, nan_frac=0.008)
    band_h = ch.band(512)
    band = hip_ctx.band_upload(band_h)
    st, en = _frame_table(nb)
    fr = band.frames(st, en)
    sh = fr.overlap()
    assert (sh[:-1] >= 0).all() and sh[-1] == -1          # NaN bins are dropped by both frames of a pair: every pair shares its block
    assert (np.abs(sh[1:-1] - 200) <= 8).all()
    Ms = np.quantile(band_h[band_h > 0], [0.95, 0.99])
    gw, gr = hip.gauss_weights(sigma)
    runs = {}
    for name, kv in (('default', dict(STP_REUSE=None, STP_SYM=None)), ('no_overlap', dict(STP_REUSE='0', STP_SYM=None)),
                     ('no_symmetry', dict(STP_REUSE=None, STP_SYM='0')), ('neither', dict(STP_REUSE='0', STP_SYM='0'))):
        with _env(**kv):
            runs[name] = fr.stripe_search(Ms, sigma=sigma)
    ref = runs['neither']
    assert len(ref) > 300
    for name, r in runs.items():
        assert r.tobytes() == ref.tobytes(), name
    with _env(STP_CANNY='exact', STP_GRAY='exact'):
        assert fr.stripe_search(Ms, sigma=sigma).tobytes() == ref.tobytes()
    exp = _oracle_records(ch.block, st, en, Ms, gw)
    assert _as_tuples(runs['default']) == exp
    fr.close(); band.close()


def test_chr16_sweep_identical(hip_ctx):
    from stripenn_amd import synth
    nb = 19642
    ch = synth.SynthChrom(nb, 16)
    band_h = ch.band(512)
    band = hip_ctx.band_upload(band_h)
    st, en = _frame_table(nb)
    M = np.quantile(band_h[band_h > 0], [0.95, 0.96, 0.97, 0.98, 0.99])
    fr = band.frames(st, en)
    a = fr.stripe_search(M)
    with _env(STP_REUSE='0'):
        b = fr.stripe_search(M)
    with _env(STP_REUSE='0', STP_SYM='0'):
        c = fr.stripe_search(M)
    assert len(a) > 10000 and a.tobytes() == b.tobytes() == c.tobytes()
    fr.close(); band.close()


def test_pairs_that_do_not_keep_the_same_bins_fall_back(hip_ctx):
    """Bin 330 has contacts only with bins beyond 420: its column is empty in frame 1 (rows 100-499 hold a few of them -> kept)
    ... precisely: frame 1 covers bins 100-499, frame 2 bins 300-699.  Bin 350 touches only bins 520-560: frame 1 drops it (no
    contact inside 100-499), frame 2 keeps it -> the pair (1, 2) does not share its block; every other pair does.  Records
    equal the oracle's either way."""
    n = 1100
    rng = np.random.default_rng(5)
    rr, cc = np.mgrid[0:n, 0:n].astype(np.float64)
    base = 240.0 / (1.0 + np.abs(rr - cc)) + 1.0
    up = np.triu(np.round(base + np.sqrt(base) * rng.standard_normal((n, n))).clip(0) / 4.0)
    dense = up + np.triu(up, 1).T
    dense += np.where((np.abs(cc - rr) > 30) & ((np.minimum(cc, rr) % 150) < 3), 5.0, 0.0)
    dense = np.where(np.abs(cc - rr) <= 500, dense, 0.0)
    b = 350
    keep = np.zeros(n, bool); keep[520:561] = True
    dense[b, ~keep] = 0.0; dense[~keep, b] = 0.0
    assert np.array_equal(dense, dense.T)
    band = hip_ctx.band_upload(_band_of(dense))
    st, en = _frame_table(n)
    fr = band.frames(st, en)
    sh = fr.overlap()
    assert sh[1] == -1 and (sh[[0, 2, 3, 4]] >= 0).all() and sh[-1] == -1, sh
    assert fr.S[1] == 399 and fr.S[2] == 400
    Ms = np.quantile(dense[dense > 0], [0.95, 0.99])
    a = fr.stripe_search(Ms)
    with _env(STP_REUSE='0', STP_SYM='0'):
        c = fr.stripe_search(Ms)
    assert a.tobytes() == c.tobytes() and len(a) > 50

    def block(r0, r1, c0, c1):
        return dense[r0:r1, c0:c1]
    gw, gr = O.gauss_weights(2.0)
    assert _as_tuples(a) == _oracle_records(block, st, en, Ms, gw)
    fr.close(); band.close()
