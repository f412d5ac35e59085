"""Randomised parity sweep: StripeSearch records of the HIP chain vs the CPU oracle on frames the fixed
fixtures do not cover -- tiny / ragged frames, empty and NaN-ridden bins, fully saturated images (every
tile takes the flat-window path), faint structure around the skip threshold, dense stripes, sigma 2.5
(radius 10) and other mean-filter sizes.  Bit-exact, like every other stage test."""
import numpy as np
import pytest

from oracle import oracle as O
from stripenn_amd import synth

pytestmark = pytest.mark.gpu
HW = 512


def _band_of(A):
    n = A.shape[0]
    i = np.arange(n)[:, None]; j = i + np.arange(-HW, HW)[None, :]
    return np.ascontiguousarray(np.where((j >= 0) & (j < n), A[i, np.clip(j, 0, n - 1)], 0.0))


def _check(hip_ctx, A, frames, Ms, sigma=2.0, bfilter=3, minH=10, maxW=8, need=0):
    band = hip_ctx.band_upload(_band_of(A))
    st = np.array([f[0] for f in frames], np.int32); en = np.array([f[1] for f in frames], np.int32)
    fr = band.frames(st, en)
    recs = fr.stripe_search(Ms, sigma=sigma, bfilter=bfilter, minH=minH, maxW=maxW)
    gw, gr = O.gauss_weights(sigma)
    exp = []
    for fi, (s, e) in enumerate(frames):
        D = A[s:e + 1, s:e + 1].copy(); D[np.isnan(D)] = 0
        nz = np.where(D.sum(axis=0) != 0)[0]
        assert fr.S[fi] == (len(nz) if len(nz) > 10 else 0)
        if len(nz) <= 10:
            continue
        D = np.ascontiguousarray(D[np.ix_(nz, nz)])
        for li, M in enumerate(Ms):
            r, tot = O.stripe_search(D, float(M), sigma=sigma, minH=minH, maxW=maxW, bf=bfilter, gw=gw)
            exp += [(fi, li) + tuple(int(v) for v in r[k]) + (float(tot[k]),) for k in range(len(r))]
    got = [(int(r['frame']), int(r['level']), int(r['b_index']), int(r['ud']), int(r['x']), int(r['y']), int(r['w']),
            int(r['h']), float(r['total'])) for r in recs]
    assert got == exp
    assert len(got) >= need
    fr.close(); band.close()
    return len(got)


def _sym(rng, n, f):
    A = f(rng, n)
    A = np.triu(A) + np.triu(A, 1).T
    return A


@pytest.mark.parametrize('seed', range(6))
def test_random_synthetic_chromosomes(hip_ctx, seed):
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(320, 1000))
    ch = synth.SynthChrom(n, 1000 + seed, stripe_every=int(rng.integers(20, 200)), stripe_gain=float(rng.uniform(1.5, 5.0)),
                          nan_frac=float(rng.choice([0.0, 0.005, 0.05, 0.2])), balanced=bool(rng.integers(0, 2)))
    A = ch.block(0, n, 0, n)
    pos = A[A > 0]
    Ms = np.quantile(pos, np.sort(rng.uniform(0.5, 0.999, 3)))
    frames = []
    for _ in range(4):
        s = int(rng.integers(0, n - 12)); e = min(n - 1, s + int(rng.integers(11, 400)))
        frames.append((s, e))
    frames.append((0, min(399, n - 1)))
    _check(hip_ctx, A, frames, Ms, sigma=float(rng.choice([2.0, 2.5])), bfilter=int(rng.choice([3, 3, 5])))


def test_saturated_and_nearly_flat_images(hip_ctx):
    """M far below / above the data (all-black / all-white images: every tile is skipped by the flat-window
    rule), and smooth ramps whose windows straddle the rule's threshold."""
    rng = np.random.default_rng(7)
    n = 800
    A = _sym(rng, n, lambda r, m: r.gamma(2.0, 2.0, (m, m)))
    _check(hip_ctx, A, [(0, 399), (350, 749)], [1e-9, 1e9, float(np.quantile(A, 0.9))])
    # one sharp block on a constant background: only the tiles around the block have work to do
    B = np.full((n, n), 3.0)
    B[200:260, 205:212] = 0.2; B[205:212, 200:260] = 0.2
    B[500:640, 520:524] = 0.5; B[520:524, 500:640] = 0.5
    got = _check(hip_ctx, B, [(100, 499), (400, 799)], [4.0, 6.0, 12.0], need=1)
    assert got > 0
    # gentle ramps: grey ranges per window from far below to just above STP_FLAT_RANGE
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing='ij')
    for slope in (1e-5, 6e-5, 1.2e-4, 4e-4, 2e-3):
        Cm = 5.0 + slope * np.abs(i - j) * 40 + 1e-3 * np.sin(0.07 * (i + j))
        _check(hip_ctx, Cm, [(0, 399), (250, 649)], [10.0, 20.0])


def test_tiny_sparse_and_ragged_frames(hip_ctx):
    rng = np.random.default_rng(11)
    n = 700
    A = _sym(rng, n, lambda r, m: r.poisson(0.4, (m, m)).astype(np.float64) * (r.random((m, m)) < 0.5))
    A[300:330, :] = 0; A[:, 300:330] = 0                 # a block of empty bins
    A[100:103, :] = np.nan; A[:, 100:103] = np.nan       # NaN bins
    frames = [(0, 10), (0, 11), (5, 30), (90, 140), (280, 360), (0, 399), (301, 700 - 1), (650, 699), (688, 699)]
    _check(hip_ctx, A, frames, [1.0, 2.0, 5.0], minH=3, maxW=12)
    _check(hip_ctx, A, frames[:6], [1.0, 3.0], minH=0, maxW=2)
