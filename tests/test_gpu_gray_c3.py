"""k_gray_c3 (grey images from min / shared row sums / one product, certified against float rounding boundaries, the
flagged lanes redone in the reference's operations) against k_gray<1> (every operation of the reference:
getStripe.py:889-913, ImageProcessing.py:15-31) and against the oracle: the grey images must be bit for bit the same, and so
must everything downstream.  STP_GRAY=exact selects the exact kernel per call."""
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


class _exact_gray:
    def __enter__(self):
        os.environ['STP_GRAY'] = 'exact'

    def __exit__(self, *a):
        os.environ.pop('STP_GRAY', None)


def _band_of(dense, hw=512):
    n = dense.shape[0]
    band = np.zeros((n, 2 * hw))
    for i in range(n):
        lo, hi = max(0, i - hw), min(n, i + hw)
        band[i, lo - i + hw:hi - i + hw] = dense[i, lo:hi]
    return band


def test_grey_images_of_both_kernels_and_the_oracle(hip_ctx):
    """Synthetic frames (noisy, planted stripes, NaN bins: ragged frame sizes) at maxpixel levels from the median to
    beyond the maximum (everything saturated / nothing saturated): every brightness image, certified == exact == oracle."""
    from stripenn_amd import synth
    ch = synth.SynthChrom(2600, 78, stripe_every=60, stripe_gain=3.0, nan_frac=0.02)
    band = hip_ctx.band_upload(ch.band(512))
    st = np.array([0, 300, 900, 1500, 2200]); en = st + 399
    fr = band.frames(st, en)
    blk = ch.block(0, 2600, 0, 2600)
    pos = blk[blk > 0]
    Ms = list(np.quantile(pos, [0.5, 0.9, 0.97, 0.995])) + [float(pos.max()) * 3.0, float(pos.min()) * 0.5]
    nimg = 0
    for f in range(len(st)):
        D, nz = O.frame_dense(ch.block, int(st[f]), int(en[f]))
        D = np.ascontiguousarray(D[np.ix_(nz, nz)])
        for M in Ms:
            gp = O.gplane(D, float(M))
            for bi in range(6):
                a = fr.dbg_stages(f, float(M), bi)
                with _exact_gray():
                    b = fr.dbg_stages(f, float(M), bi)
                assert np.array_equal(a['gray'].view(np.uint32), b['gray'].view(np.uint32)), (f, M, bi)
                assert np.array_equal(a['cls'], b['cls'])
                if bi in (0, 3, 5):
                    og = O.gray(gp, O.brightness_levels()[bi], 3)
                    assert np.array_equal(a['gray'].view(np.uint32), og.view(np.uint32)), (f, M, bi)
                nimg += 1
    assert nimg == 180
    fr.close(); band.close()


def test_few_bit_contact_values(hip_ctx):
    """Contact maps made of values with few significant bits (integer counts, halves, powers of two relative to M): their
    blurred sums sit on float half-way patterns far more often than noisy data, so the flagged path does real work."""
    n = 800
    rr, cc = np.mgrid[0:n, 0:n].astype(np.float64)
    rng = np.random.default_rng(5)
    ints = rng.integers(0, 9, (n, n)).astype(np.float64)
    cases = {
        'ints': ints,
        'halves': ints / 2.0 + np.where((rr + cc) % 7 == 0, 0.25, 0.0),
        'pow2': np.ldexp(1.0, -rng.integers(0, 30, (n, n))) * 8.0,
        'blocks': np.where((cc // 37 + rr // 53) % 2 == 0, 12.0, 3.0),
    }
    for name, dense in cases.items():
        dense = np.where(np.abs(cc - rr) <= 500, dense, 0.0)
        dense = np.triu(dense) + np.triu(dense, 1).T
        band = hip_ctx.band_upload(_band_of(dense))
        st = np.array([0, 200, 400]); en = st + 399
        fr = band.frames(st, en)
        for M in (8.0, 6.0, 1.0 + 2.0 ** -20):
            for f in range(3):
                D = dense[st[f]:en[f] + 1, st[f]:en[f] + 1]
                nz = np.where(np.nan_to_num(D).sum(axis=0) != 0)[0]
                Dc = np.ascontiguousarray(D[np.ix_(nz, nz)])
                gp = O.gplane(Dc, M)
                for bi in (0, 2, 5):
                    a = fr.dbg_stages(f, M, bi)
                    with _exact_gray():
                        b = fr.dbg_stages(f, M, bi)
                    assert np.array_equal(a['gray'].view(np.uint32), b['gray'].view(np.uint32)), (name, M, f, bi)
                    og = O.gray(gp, O.brightness_levels()[bi], 3)
                    assert np.array_equal(a['gray'].view(np.uint32), og.view(np.uint32)), (name, M, f, bi)
        fr.close(); band.close()


def test_search_records_identical_under_both_grey_kernels(hip_ctx):
    """A chromosome-size sweep: stripe records byte for byte the same under the certified and the exact grey kernel."""
    from stripenn_amd import synth
    nb = 6000
    ch = synth.SynthChrom(nb, 41)
    band_h = ch.band(512)
    band = hip_ctx.band_upload(band_h)
    nfr = -(-nb // 200)
    st = np.array([max(0, i * 200 - 100) for i in range(nfr)]); en = np.minimum((np.arange(nfr) + 1) * 200 + 99, nb - 1)
    M = np.quantile(band_h[band_h > 0], [0.95, 0.97, 0.99])
    fr = band.frames(st, en)
    a = fr.stripe_search(M)
    with _exact_gray():
        b = fr.stripe_search(M)
    assert len(a) > 200 and a.tobytes() == b.tobytes()
    fr.close(); band.close()
