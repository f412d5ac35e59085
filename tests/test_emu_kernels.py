"""CPU replay (tests/emu) of the HIP kernels' phase code vs the oracle: the kernel logic is checked
bit for bit before anything runs on a GPU.  The emulator is test-only code."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


class ER(C.Structure):
    _fields_ = [('ud', C.c_int32), ('x', C.c_int32), ('y', C.c_int32), ('w', C.c_int32), ('h', C.c_int32),
                ('total', C.c_double)]


@pytest.fixture(scope='module')
def emu():
    subprocess.check_call(['make', '-s', '-C', os.path.join(HERE, 'emu')])
    E = C.CDLL(os.path.join(HERE, 'emu', 'libstp_emu.so'))
    E.emu_lines.restype = C.c_int
    E.emu_compact.restype = C.c_int
    return E


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _unpack(words, S):
    b = np.unpackbits(words.view(np.uint8).reshape(400, 56), axis=1, bitorder='little')
    return b[:S, :S]


def _unpack_cls(words, S):
    """a class bit-plane as the Canny kernels hand it to k_lines: word-column-major (STP_CLS: word w of row y at w * 400 + y)"""
    return _unpack(np.ascontiguousarray(words.reshape(7, 400).T), S)


def _canny_f32(emu, full, S, R, gw):
    """Replay of k_canny_f32 (stp_canny32.h): f32 phases, certified class test, exact per-pixel resolver.
    Returns the class map and (candidates, pixels sent to the resolver)."""
    low = np.zeros(2800, np.uint64); high = np.zeros(2800, np.uint64); cnt = np.zeros(2, np.int64)
    emu.emu_canny_f32(_p(np.ascontiguousarray(full)), S, R, _p(gw), _p(low), _p(high), _p(cnt))
    return _unpack_cls(low, S).astype(np.uint8) + _unpack_cls(high, S), cnt


def _canny_f32_sym(emu, full, S, R, gw):
    """... with `mirror` (round 6): the tiles strictly below the diagonal are not computed; the tiles whose transpose lies there
    hand their class words over transposed -- the f32 verdicts as they are, the undecidable pixels settled once per position."""
    low = np.zeros(2800, np.uint64); high = np.zeros(2800, np.uint64); cnt = np.zeros(2, np.int64)
    emu.emu_canny_f32_sym(_p(np.ascontiguousarray(full)), S, R, _p(gw), _p(low), _p(high), _p(cnt))
    return _unpack_cls(low, S).astype(np.uint8) + _unpack_cls(high, S), cnt


@pytest.mark.parametrize('ci', [0, 2, 4, 5])
def test_emulated_kernels_match_oracle(emu, golden_stages, chr7, ci):
    g = golden_stages
    hw, W = 512, 1024
    band = chr7.band(hw)
    gw = np.ascontiguousarray(g['gw_2p0'])
    bvals = np.ascontiguousarray(g['bvals'])
    p = 'c%d_' % ci
    start, end, S, M = int(g[p + 'start']), int(g[p + 'end']), int(g[p + 'S']), float(g[p + 'M'])
    nz = np.zeros(400, np.int16)
    assert emu.emu_compact(_p(band), W, hw, C.c_int64(start), end - start + 1, _p(nz)) == S
    assert np.array_equal(nz[:S], g[p + 'nz'])
    gray = np.zeros((6, 400, 400), np.float32)
    emu.emu_gray(_p(band), W, hw, C.c_int64(start), _p(nz), S, C.c_double(M), _p(bvals), 6, 1, _p(gray))
    gray_c3 = np.zeros((6, 400, 400), np.float32); c3n = np.zeros(2, np.int64)     # k_gray_c3: the certified evaluation
    emu.emu_gray_c3(_p(band), W, hw, C.c_int64(start), _p(nz), S, C.c_double(M), _p(bvals), 6, _p(gray_c3), _p(c3n))
    assert np.array_equal(gray_c3, gray) and c3n[0] == 6 * S * S and c3n[1] < 20
    D, nzo = O.frame_dense(chr7.block, start, end)
    D = np.ascontiguousarray(D[np.ix_(nzo, nzo)])
    gp = O.gplane(D, M)
    nrec = 0
    exp_recs, exp_tot = O.stripe_search(D, M, gw=gw)
    got = []
    for bi in range(6):
        og = O.gray(gp, bvals[bi])
        assert np.array_equal(og, gray[bi, :S, :S])
        low = np.zeros(2800, np.uint64); high = np.zeros(2800, np.uint64)
        emu.emu_canny(_p(gray[bi]), S, 8, _p(gw), _p(low), _p(high))
        oe, dbg = O.canny(og, gw, 8, debug=True)
        assert np.array_equal(_unpack_cls(low, S).astype(np.uint8) + _unpack_cls(high, S), dbg['cls'])
        cls32, cnt = _canny_f32(emu, gray[bi], S, 8, gw)       # k_canny_f32's phases: same classes, few exact resolutions
        assert np.array_equal(cls32, dbg['cls']) and cnt[1] < 0.01 * cnt[0] + 50
        assert np.array_equal(gray[bi, :S, :S], gray[bi, :S, :S].T)          # a frame's image is its own transpose ...
        cls32s, cnts = _canny_f32_sym(emu, gray[bi], S, 8, gw)               # ... so the tiles below the diagonal need not be computed
        assert np.array_equal(cls32s, dbg['cls']) and cnts[0] <= cnt[0]
        dbgw = np.zeros(4 * 2800, np.uint64); cols = np.zeros(1200, np.int16); recs = (ER * 128)(); sw = C.c_int(0)
        n = emu.emu_lines(_p(low), _p(high), _p(band), W, hw, C.c_int64(start), _p(nz), S, 10, 8, _p(dbgw), _p(cols), recs,
                          128, C.byref(sw))
        assert np.array_equal(_unpack(dbgw[:2800], S), oe)
        ov = O.vertical_line(oe)
        assert np.array_equal(_unpack(dbgw[2800:5600], S), ov)
        t_, e_, u_ = O.columns(ov, 10)
        assert np.array_equal(cols[:S], t_) and np.array_equal(cols[400:400 + S], e_) and np.array_equal(cols[800:800 + S], u_)
        tm1, _ = O.join_dbg(oe, ov, 1, 10, 8)
        tm2, _ = O.join_dbg(oe, ov, 2, 10, 8)
        assert np.array_equal(_unpack(dbgw[5600:8400], S), tm1) and np.array_equal(_unpack(dbgw[8400:], S), tm2)
        got += [(bi, recs[k].ud, recs[k].x, recs[k].y, recs[k].w, recs[k].h, recs[k].total) for k in range(n)]
    exp = [tuple(int(v) for v in exp_recs[k]) + (float(exp_tot[k]),) for k in range(len(exp_recs))]
    assert got == exp


def test_vertical_line_bitsliced_random(emu):
    """The bit-sliced verticalLine / hysteresis phases on random masks vs the oracle's per-pixel code."""
    rng = np.random.default_rng(5)
    for S in (400, 129, 64, 37):
        low = (rng.random((S, S)) < 0.25)
        high = low & (rng.random((S, S)) < 0.2)
        def pack(m):
            full = np.zeros((400, 448), np.uint8); full[:S, :S] = m
            rowmajor = np.packbits(full, axis=1, bitorder='little').view(np.uint64).reshape(400, 7)
            return np.ascontiguousarray(rowmajor.T).reshape(-1)          # class planes are word-column-major (STP_CLS)
        lw, hg = pack(low), pack(high)
        band = np.zeros((S + 8, 1024)); nz = np.arange(400, dtype=np.int16)
        dbgw = np.zeros(4 * 2800, np.uint64); cols = np.zeros(1200, np.int16); recs = (ER * 128)(); sw = C.c_int(0)
        emu.emu_lines(_p(lw), _p(hg), _p(band), 1024, 512, C.c_int64(0), _p(nz), S, 10, 8, _p(dbgw), _p(cols), recs, 128,
                      C.byref(sw))
        # oracle hysteresis: components of low containing a high pixel
        from scipy import ndimage as ndi
        lab, n = ndi.label(low, np.ones((3, 3)))
        good = np.zeros(n + 1, bool); good[np.unique(lab[high])] = True; good[0] = False
        edges = good[lab]
        assert np.array_equal(_unpack(dbgw[:2800], S).astype(bool), edges)
        assert np.array_equal(_unpack(dbgw[2800:5600], S), O.vertical_line(edges.astype(np.uint8)))


def _plateau_images(S):
    """Grey images full of exact ties: straight step edges (equal magnitudes along the edge), a symmetric
    blob, a staircase -- the certified-approximation NMS must fall back to the exact hypot there."""
    yy, xx = np.mgrid[0:S, 0:S]
    imgs = []
    imgs.append(np.where(xx < S // 2, 0.25, 0.9))                                   # vertical step
    imgs.append(np.where(yy < S // 3, 0.8, 0.3))                                    # horizontal step
    imgs.append(np.where(xx + yy < S, 0.2, 0.95))                                   # anti-diagonal step
    imgs.append(np.where(xx - yy > 7, 0.35, 0.85))                                  # diagonal step
    imgs.append(0.5 + 0.4 * ((np.abs(xx - S // 2) < 20) & (np.abs(yy - S // 2) < 20)))   # symmetric square
    imgs.append(0.3 + 0.1 * ((xx // 16) % 5))                                       # staircase
    return [np.ascontiguousarray(i, dtype=np.float32) for i in imgs]


@pytest.mark.parametrize('S', [400, 143])
def test_canny_ties_and_plateaus(emu, S, golden_stages):
    gw = np.ascontiguousarray(golden_stages['gw_2p0'])
    for k, img in enumerate(_plateau_images(S)):
        full = np.zeros((400, 400), np.float32)
        full[:S, :S] = img
        low = np.zeros(2800, np.uint64); high = np.zeros(2800, np.uint64)
        emu.emu_canny(_p(full), S, 8, _p(gw), _p(low), _p(high))
        oe, dbg = O.canny(img, gw, 8, debug=True)
        got = _unpack_cls(low, S).astype(np.uint8) + _unpack_cls(high, S)
        assert np.array_equal(got, dbg['cls']), 'plateau image %d (S=%d)' % (k, S)
        assert dbg['cls'].any()
        cls32, cnt = _canny_f32(emu, full, S, 8, gw)            # exact ties: (nearly) every candidate goes to the resolver
        assert np.array_equal(cls32, dbg['cls']), 'f32 path, plateau image %d (S=%d)' % (k, S)
        assert cnt[1] > 0


def test_generic_radius_path(emu, golden_stages):
    """k_canny (generic radius, run-time loops) replayed with blocked = 0 on one golden image."""
    g = golden_stages
    gw = np.ascontiguousarray(g['gw_2p0'])
    img = _plateau_images(200)[2] * 0.5 + 0.25 * np.random.default_rng(3).random((200, 200)).astype(np.float32)
    img = np.ascontiguousarray(img, dtype=np.float32)
    full = np.zeros((400, 400), np.float32); full[:200, :200] = img
    low = np.zeros(2800, np.uint64); high = np.zeros(2800, np.uint64)
    emu.emu_canny2(_p(full), 200, 8, _p(gw), _p(low), _p(high), 0)
    oe, dbg = O.canny(img, gw, 8, debug=True)
    assert np.array_equal(_unpack_cls(low, 200).astype(np.uint8) + _unpack_cls(high, 200), dbg['cls'])
    # sigma 3.0 -> radius 12 exists only in the generic form
    from oracle.oracle import gauss_weights
    w3, r3 = gauss_weights(3.0)
    emu.emu_canny2(_p(full), 200, r3, _p(w3), _p(low), _p(high), 0)
    oe, dbg = O.canny(img, w3, r3, debug=True)
    assert np.array_equal(_unpack_cls(low, 200).astype(np.uint8) + _unpack_cls(high, 200), dbg['cls'])


# ----------------------------------------------------------------------------- certified FMA (stp_gauss_fma.h)
@pytest.fixture(scope='module')
def emu_allnear(emu):
    """The same replay built with every Gaussian output flagged "near a rounding boundary": only the exact-order
    fallback of the certified-FMA passes runs."""
    return C.CDLL(os.path.join(HERE, 'emu', 'libstp_emu_allnear.so'))


def _noisy(S, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:S, 0:S]
    img = 0.299 + 0.701 * np.clip(0.5 + 0.35 * np.sin(xx / 7.0) * np.cos(yy / 11.0) + 0.25 * rng.standard_normal((S, S)), 0, 1)
    return np.ascontiguousarray(img, dtype=np.float32)


@pytest.mark.parametrize('sigma,key', [(2.0, 'gw_2p0'), (2.5, 'gw_2p5')])
@pytest.mark.parametrize('S', [400, 399, 301, 143, 64, 41, 12])
def test_canny_sizes_radii_and_exact_fallback(emu, emu_allnear, golden_stages, S, sigma, key):
    """Image sizes that end inside a tile, both default radii, noisy and plateau images: classes bit-exact vs the
    oracle -- with the certified-FMA Gaussian passes and with their exact-order fallback alone."""
    gw = np.ascontiguousarray(golden_stages[key])
    R = (len(gw) - 1) // 2
    for k, img in enumerate([_noisy(S, 11 + S)] + (_plateau_images(S)[:3] if S >= 41 else [])):
        full = np.zeros((400, 400), np.float32)
        full[:S, :S] = img
        oe, dbg = O.canny(img, gw, R, debug=True)
        for lib in (emu, emu_allnear):
            low = np.zeros(2800, np.uint64); high = np.zeros(2800, np.uint64)
            lib.emu_canny(_p(full), S, R, _p(gw), _p(low), _p(high))
            got = _unpack_cls(low, S).astype(np.uint8) + _unpack_cls(high, S)
            assert np.array_equal(got, dbg['cls']), 'image %d (S=%d, sigma %.1f)' % (k, S, sigma)
        cls32, cnt = _canny_f32(emu, full, S, R, gw)
        assert np.array_equal(cls32, dbg['cls']), 'f32 path, image %d (S=%d, sigma %.1f)' % (k, S, sigma)


def test_fma_certification_random_windows(emu, golden_stages):
    """The claim behind the fused Gaussian sum: whenever the near-boundary test does not flag an output,
    float(fused sum) == float(sum in the reference's order).  8 M random windows per case (grey-like values, some
    zero taps); the two f64 sums never differ by more than the R + 1 ulp of the derivation."""
    emu.emu_certify_fma.restype = C.c_longlong
    for key, lo in (('gw_2p0', 0.299), ('gw_2p5', 0.299), ('gw_2p0', 1e-3)):
        gw = np.ascontiguousarray(golden_stages[key])
        R = (len(gw) - 1) // 2
        flagged = C.c_longlong(0); worst = C.c_double(0)
        bad = emu.emu_certify_fma(_p(gw), R, C.c_longlong(8000000), C.c_ulonglong(R * 7 + 1), C.c_double(lo),
                                  C.byref(flagged), C.byref(worst))
        assert bad == 0
        assert worst.value <= R + 1
        assert flagged.value < 100          # ~64 / 2^29 of the outputs


@pytest.mark.parametrize('mode', [0, 1, 2])
def test_gray_certification_random_windows(emu, mode):
    """The claim behind k_gray_c3 (stp_phases.h, "certified grey"): whenever the near-boundary test does not flag an
    output, the grey value from min / shared row sums / one product equals the one from the reference's two divisions,
    two products and nine sequential additions.  1.4e7 random 3 x 3 windows per mode (spread values with zeros and
    saturated pixels; flat windows; values at M (1 - 2^-k) and at the brightness cut +- ulps): no unflagged mismatch,
    the two f64 sums within the 22 ulp of the derivation, ~2 * 64 / 2^29 of the outputs flagged."""
    emu.emu_certify_gray.restype = C.c_longlong
    flagged = C.c_longlong(0); worst = C.c_double(0)
    n = 14000000
    bad = emu.emu_certify_gray(C.c_longlong(n), C.c_ulonglong(1234 + mode), mode, C.byref(flagged), C.byref(worst))
    assert bad == 0
    assert worst.value <= 22.0
    # expected n * 2 * 64 / 2^29 = 3.3 on spread data; sums of few-bit values (mode 2) sit on half-way patterns more often
    assert flagged.value < (40 if mode < 2 else n // 1000)


@pytest.mark.parametrize('sigma,key', [(2.0, 'gw_2p0'), (2.5, 'gw_2p5')])
def test_f32_path_on_threshold_and_octant_boundaries(emu, golden_stages, sigma, key):
    """Adversarial inputs for k_canny_f32's error budget: ramps whose magnitude sits on the 0.1 / 0.2 thresholds
    (8 * slope), diagonal ramps (|isobel| == |jsobel|: octant boundary), each with noise far below and around the f32
    rounding error, and a bright image (grey near 1: the largest absolute errors).  Classes must equal the oracle's
    whether the certified test or the exact resolver decides."""
    gw = np.ascontiguousarray(golden_stages[key])
    R = (len(gw) - 1) // 2
    S = 200
    yy, xx = np.mgrid[0:S, 0:S].astype(np.float64)
    rng = np.random.default_rng(5)
    imgs = []
    for slope, amp in [(0.0125, 0.0), (0.0125, 1e-7), (0.0125, 3e-6), (0.025 / 2, 1e-5), (0.025, 1e-6), (0.0124, 2e-5)]:
        base = 0.05 + slope * (xx % 60)                                   # saw-tooth: ramps and falling steps
        imgs.append(base + amp * rng.standard_normal((S, S)))
    for amp in (0.0, 1e-6, 1e-4):
        imgs.append(0.1 + 0.00884 * ((xx + yy) % 80) / 1.0 + amp * rng.standard_normal((S, S)))       # diagonal ramp, m ~ 0.1
        imgs.append(0.9 - 0.004 * ((xx - yy) % 50) + amp * rng.standard_normal((S, S)))
    imgs.append(np.clip(0.97 + 0.03 * rng.standard_normal((S, S)), 0, 1))     # bright, saturating
    imgs.append(np.clip(0.5 + 0.5 * np.sin(xx / 3.0) * np.sin(yy / 5.0), 0, 1))
    total = np.zeros(2, np.int64)
    for k, img in enumerate(imgs):
        img = np.ascontiguousarray(np.clip(img, 0, 1), dtype=np.float32)
        full = np.zeros((400, 400), np.float32)
        full[:S, :S] = img
        oe, dbg = O.canny(img, gw, R, debug=True)
        cls32, cnt = _canny_f32(emu, full, S, R, gw)
        total += cnt
        assert np.array_equal(cls32, dbg['cls']), 'image %d (sigma %.1f): %d pixels differ' % (k, sigma, int((cls32 != dbg['cls']).sum()))
    assert total[1] > 1000        # the inputs do reach the resolver


@pytest.mark.parametrize('sigma,key', [(2.0, 'gw_2p0'), (2.5, 'gw_2p5')])
def test_f32_error_budget_holds_for_every_intermediate(emu, golden_stages, chr7, sigma, key):
    """The bounds k_canny_f32's tile-wide certificates rest on (stp_canny32.h, c32_budget), checked quantity by quantity
    against the reference's f64 intermediates (oracle debug arrays) on every pixel of noisy, bright, dark and
    synthetic-frame images: with E_G the tile's Sobel bound in units of u g (interior tiles ~131, border tiles ~165
    at sigma 2), |S_f32 - S| <= (E_G - 16.1) / 8, |Sobel_f32 - Sobel| <= E_G, |m_f32 - m| <= sqrt(2) E_G + 17.7,
    g the scale the tile used (flat tiles are skipped by the kernel and carry no values).  Also reports how far below
    the bounds the observed errors stay."""
    gw = np.ascontiguousarray(golden_stages[key])
    R = (len(gw) - 1) // 2
    u = 2.0 ** -24
    rng = np.random.default_rng(17)
    S = 300
    yy, xx = np.mgrid[0:S, 0:S].astype(np.float64)
    imgs = [_noisy(S, 5),
            np.clip(0.9 + 0.1 * rng.standard_normal((S, S)), 0, 1),                      # bright: the largest absolute errors
            0.299 + 0.05 * rng.random((S, S)),                                            # dark, low contrast
            np.clip(0.5 + 0.5 * np.sin(xx / 2.5) * np.sin(yy / 3.5), 0, 1),               # strong gradients everywhere
            np.clip(rng.random((S, S)) > 0.5, 0.299, 1.0).astype(np.float64)]             # salt and pepper between the extremes
    g = golden_stages
    p = 'c2_'
    band = chr7.band(512)
    start, end, Sg, M = int(g[p + 'start']), int(g[p + 'end']), int(g[p + 'S']), float(g[p + 'M'])
    nz = np.zeros(400, np.int16)
    emu.emu_compact(_p(band), 1024, 512, C.c_int64(start), end - start + 1, _p(nz))
    gray6 = np.zeros((6, 400, 400), np.float32)
    emu.emu_gray(_p(band), 1024, 512, C.c_int64(start), _p(nz), Sg, C.c_double(M), _p(np.ascontiguousarray(g['bvals'])), 6, 1, _p(gray6))
    imgs += [gray6[0, :Sg, :Sg], gray6[5, :Sg, :Sg]]
    worst = np.zeros(3)            # observed error / bound
    budgets = set()
    for k, img in enumerate(imgs):
        img = np.ascontiguousarray(img, dtype=np.float32)
        Si = img.shape[0]
        full = np.zeros((400, 400), np.float32); full[:Si, :Si] = img
        d = [np.full((400, 400), np.nan, np.float32) for _ in range(6)]
        emu.emu_canny_f32_dump(_p(full), Si, R, _p(gw), *[_p(a) for a in d])
        _, dbg = O.canny(img, gw, R, debug=True)
        dS, dI, dJ, dM, dG, dE = [a[:Si, :Si].astype(np.float64) for a in d]
        live = ~np.isnan(dS)                                     # tiles the kernel did not skip as flat
        assert live.any()
        budgets |= set(np.round(dE[live], 1).tolist())
        ug = u * dG
        inner = live.copy(); inner[0, :] = inner[-1, :] = False; inner[:, 0] = inner[:, -1] = False
        rS = (np.abs(dS - dbg['smoothed']) / ug / ((dE - 16.1) / 8.0))[live]
        rG = (np.maximum(np.abs(dI - dbg['isobel']), np.abs(dJ - dbg['jsobel'])) / ug / dE)[inner]
        rM = (np.abs(dM - dbg['mag']) / ug / (1.41422 * dE + 17.7))[inner]
        assert rS.max() <= 1.0 and rG.max() <= 1.0 and rM.max() <= 1.0, (k, rS.max(), rG.max(), rM.max())
        worst = np.maximum(worst, [rS.max(), rG.max(), rM.max()])
    print('E_G budgets in use (u g):', sorted(budgets), '; largest observed error / bound: smoothed %.2f, Sobel %.2f, magnitude %.2f' % tuple(worst))
    assert worst[0] > 0.05           # the comparison is not vacuous


@pytest.mark.parametrize('sigma', [1.0, 1.5, 3.0])
@pytest.mark.parametrize('S', [400, 301, 64, 41])
def test_other_tiled_radii(emu, emu_allnear, S, sigma):
    """The tiled kernels are also instantiated for the radii of sigma 1.0, 1.5 and 3.0 (4, 6, 12): k_canny_pipe's phases
    (certified FMA and its exact-order fallback alone) and k_canny_f32's against the oracle, noisy and plateau images."""
    gw, R = O.gauss_weights(sigma)
    gw = np.ascontiguousarray(gw)
    assert R in (4, 6, 12)
    for k, img in enumerate([_noisy(S, 3 + S)] + (_plateau_images(S)[:4] if S >= 41 else [])):
        full = np.zeros((400, 400), np.float32)
        full[:S, :S] = img
        oe, dbg = O.canny(img, gw, R, debug=True)
        for lib in (emu, emu_allnear):
            low = np.zeros(2800, np.uint64); high = np.zeros(2800, np.uint64)
            lib.emu_canny(_p(full), S, R, _p(gw), _p(low), _p(high))
            got = _unpack_cls(low, S).astype(np.uint8) + _unpack_cls(high, S)
            assert np.array_equal(got, dbg['cls']), 'image %d (S=%d, sigma %.1f)' % (k, S, sigma)
        cls32, cnt = _canny_f32(emu, full, S, R, gw)
        assert np.array_equal(cls32, dbg['cls']), 'f32 path, image %d (S=%d, sigma %.1f)' % (k, S, sigma)


@pytest.mark.parametrize('sigma', [1.0, 1.5, 2.0, 2.5, 3.0])
def test_tile_wide_budget_constants(emu, sigma):
    """c32_budget's constants from first principles: rho = 3 + sum_k P_k / W recomputed here in numpy for the full window,
    for every one-sided cut and for every cut; the budgets follow from them (E_S = rho_y + rho_x + scaling roundings,
    E_G = 8 E_S + 16, E_M = sqrt(2) E_G + 17.6, T0 = 2 E_M + 17) and are ordered interior < one-sided < any cut < the
    relative budget the second look uses."""
    gw, R = O.gauss_weights(sigma)
    gw = np.ascontiguousarray(gw)

    def rho(lo, hi):
        W = P = gw[R]
        tot = 0.0
        for k in range(R, 0, -1):
            add = (gw[R - k] if -k >= lo else 0.0) + (gw[R - k] if k <= hi else 0.0)
            P += add; W += add; tot += P
        return 3.0 + tot / W
    cuts = [(lo, hi) for lo in range(-R, 1) for hi in range(0, R + 1)]
    r_int = rho(-R, R)
    r_one = max(rho(lo, hi) for lo, hi in cuts if lo == -R or hi == R)
    r_any = max(rho(lo, hi) for lo, hi in cuts)
    out = np.zeros(9, np.float32); rr = np.zeros(3)
    emu.emu_c32_budget(_p(gw), R, _p(out), _p(rr))
    assert np.allclose(rr, [r_int, r_one, r_any], rtol=1e-12)
    assert 5.0 < r_int < r_one <= r_any <= 3.0 + R + 1e-9          # a window of the centre tap alone: P_k = W at every step
    for t, (r, sc) in enumerate(((r_int, 2.1), (r_one, 4.2), (r_any, 4.2))):
        es = 2 * r + sc
        eg = 8 * es + 16.1
        assert eg <= out[3 * t] <= eg * 1.001
        assert out[3 * t + 1] >= 1.41421 * out[3 * t] + 17.6 and out[3 * t + 2] >= 2 * out[3 * t + 1] + 17.0
    assert out[0] < out[3] <= out[6] and out[3] < 234.0


@pytest.mark.parametrize('sigma,key', [(2.0, 'gw_2p0'), (2.5, 'gw_2p5')])
def test_interior_flat_window_threshold(emu, golden_stages, sigma, key):
    """c32_flat_interior (stp_canny32.h): in tiles whose every window lies inside the image a grey range below
    0.0849 / (5.657 (v_0 + v_1)) keeps every magnitude below 0.085.  Worst cases for the bound -- two-level images whose
    amplitude is just under the threshold: steps, corners, checkers of every period, random two-level noise -- must give
    NO class in the oracle (the magnitudes themselves are checked against 0.085), and the replay must equal the oracle on
    them and on the same patterns just above the threshold (where the tile is not skipped)."""
    gw = np.ascontiguousarray(golden_stages[key], dtype=np.float64)
    R = (len(gw) - 1) // 2
    v01 = (gw[R] + gw[R - 1]) / gw.sum()
    thr = 0.0849 / (5.65686 * v01) - 1e-5
    assert thr > 0.039                      # sigma 2.0: 0.0400, sigma 2.5: 0.0489 (STP_FLAT_RANGE is 0.015)
    S = 400
    yy, xx = np.mgrid[0:S, 0:S]
    rng = np.random.default_rng(17)
    pats = [(xx > 200), (xx > 200) ^ (yy > 200), (xx + yy > 400), ((xx // 3) + (yy // 3)) % 2 == 0, (xx // R) % 2 == 0,
            ((xx // (R + 1)) + (yy // (R + 1))) % 2 == 0, rng.random((S, S)) < 0.5, (xx - 200) ** 2 + (yy - 200) ** 2 < 90 ** 2]
    for pi, pat in enumerate(pats):
        for base in (0.0, 0.5, 1.0 - thr):
            for amp, flat in ((thr * 0.9995, True), (thr * 1.3, False)):
                img = np.ascontiguousarray(np.clip(base + amp * pat, 0, 1), dtype=np.float32)
                oe, dbg = O.canny(img, gw, R, debug=True)
                if flat:
                    inner = dbg['mag'][R + 2:S - R - 2, R + 2:S - R - 2]        # pixels whose windows lie inside the image
                    assert float(inner.max()) < 0.085, (pi, base, float(inner.max()))
                cls32, _ = _canny_f32(emu, img, S, R, gw)
                assert np.array_equal(cls32, dbg['cls']), 'pattern %d base %.2f amp %.4f: %d pixels differ' % (
                    pi, base, amp, int((cls32 != dbg['cls']).sum()))


# ----------------------------------------------------------------------------- image symmetry (k_canny_f32's `mirror`)
def _symmetric_images(S, seed):
    """Grey images that equal their transpose bit for bit: noise, steps along and across the diagonal, a square on the diagonal,
    a staircase of diagonal bands (plateaus and exact ties: the resolver decides, at both positions of a pair)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:S, 0:S]
    n = 0.25 * rng.standard_normal((S, S))
    base = 0.5 + 0.35 * np.sin((xx + yy) / 9.0) * np.cos((xx - yy) / 13.0)
    imgs = [0.299 + 0.701 * np.clip(base + (n + n.T) / 2, 0, 1),
            np.where(xx + yy < S, 0.2, 0.95),
            np.where(np.abs(xx - yy) > 7, 0.35, 0.85),
            0.5 + 0.4 * ((np.abs(xx - S // 2) < S // 5) & (np.abs(yy - S // 2) < S // 5)),
            0.3 + 0.1 * ((np.abs(xx - yy) // 16) % 5),
            np.maximum(np.where(xx < S // 2, 0.25, 0.9), np.where(yy < S // 2, 0.25, 0.9))]
    out = [np.ascontiguousarray(i, dtype=np.float32) for i in imgs]
    for i in out:
        assert np.array_equal(i, i.T)
    return out


@pytest.mark.parametrize('sigma,key', [(2.0, 'gw_2p0'), (2.5, 'gw_2p5')])
@pytest.mark.parametrize('S', [400, 399, 301, 225, 143, 97, 64, 41, 12])
def test_canny_mirrored_tiles(emu, golden_stages, S, sigma, key):
    """Class maps with the tiles below the diagonal taken from the transposes of the tiles above it == the oracle's, on symmetric
    images of sizes that end inside a tile row / column / word half, both default radii."""
    gw = np.ascontiguousarray(golden_stages[key])
    R = (len(gw) - 1) // 2
    for k, img in enumerate(_symmetric_images(S, 5 + S)):
        full = np.zeros((400, 400), np.float32)
        full[:S, :S] = img
        oe, dbg = O.canny(img, gw, R, debug=True)
        got, cnt = _canny_f32_sym(emu, full, S, R, gw)
        assert np.array_equal(got, dbg['cls']), 'mirrored tiles, image %d (S=%d, sigma %.1f)' % (k, S, sigma)
        ref, cnt0 = _canny_f32(emu, full, S, R, gw)
        assert np.array_equal(ref, got)
        if S > 128:
            assert cnt[0] < cnt0[0] or cnt0[0] == 0          # fewer tiles ran


@pytest.mark.parametrize('sigma', [1.0, 1.5, 3.0])
def test_canny_mirrored_tiles_other_radii(emu, sigma):
    from oracle.oracle import gauss_weights
    gw, R = gauss_weights(sigma)
    for S in (400, 210):
        for k, img in enumerate(_symmetric_images(S, 77)[:3]):
            full = np.zeros((400, 400), np.float32)
            full[:S, :S] = img
            oe, dbg = O.canny(img, gw, R, debug=True)
            got, _ = _canny_f32_sym(emu, full, S, R, gw)
            assert np.array_equal(got, dbg['cls']), 'mirrored tiles, image %d (S=%d, sigma %.1f)' % (k, S, sigma)


def test_every_class_word_has_exactly_one_writer(emu):
    """Round 6 invariant (image symmetry + frame overlap): for every image size, shared-block position, radius and both settings
    of `mirror`, every (row, half word) of the class planes k_lines reads is written by exactly one computed tile of
    k_canny_f32 -- directly or as a transpose -- or patched from the next frame by the loader (stp_reuse_mask); never by nobody,
    never twice; patched rows lie inside the shared block's interior."""
    emu.emu_class_word_cover.restype = C.c_longlong
    n = 0
    for S in list(range(11, 80, 7)) + list(range(80, 401, 3)):
        for R in (4, 8, 12):
            for mirror in (0, 1):
                assert emu.emu_class_word_cover(S, -1, R, mirror, 1) == 0, (S, R, mirror, 'no shared block')
                for shift in {0, 1, S // 3, S // 2 - 1, S // 2, S // 2 + 3, S - 150, S - 64, S - 40} - {s for s in range(-400, 0)}:
                    if shift >= S:
                        continue
                    assert emu.emu_class_word_cover(S, shift, R, mirror, 1) == 0, (S, shift, R, mirror)
                    assert emu.emu_class_word_cover(S, shift, R, mirror, 0) == 0, (S, shift, R, mirror, 'next frame in another launch')
                    n += 1
    assert n > 4000


def test_every_grey_value_a_computed_canny_tile_reads_is_written(emu):
    """Round 6 invariant ("grey tiles nobody reads"): for every image size, shared-block position and radius, the grey tiles
    k_gray_c3 skips (stp_gray_tile_unread) are disjoint from the windows and cells of the Canny tiles k_canny_f32 computes; a full
    frame without a shared block skips the 20 tiles of the symmetry rule, an ordinary frame six more."""
    emu.emu_gray_reader_cover.restype = C.c_longlong
    sk = C.c_int(0)
    for S in list(range(11, 80, 7)) + list(range(80, 401, 3)):
        for R in (4, 8, 12):
            assert emu.emu_gray_reader_cover(S, -1, R, 1, C.byref(sk)) == 0, (S, R)
            for shift in (0, 1, S // 3, S // 2 - 1, S // 2, S // 2 + 3, max(S - 150, 0), max(S - 64, 0)):
                assert emu.emu_gray_reader_cover(S, shift, R, 1, C.byref(sk)) == 0, (S, shift, R)
                assert emu.emu_gray_reader_cover(S, shift, R, 0, C.byref(sk)) == 0, (S, shift, R)
    assert emu.emu_gray_reader_cover(400, -1, 8, 1, C.byref(sk)) == 0 and sk.value == 20
    assert emu.emu_gray_reader_cover(398, 198, 8, 1, C.byref(sk)) == 0 and sk.value == 26
