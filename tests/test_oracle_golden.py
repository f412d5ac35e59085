"""The CPU oracle against vectors produced by the unmodified reference (tests/golden/stages_chr7.npz,
made by oracle/refharness/gen_golden.py): every stage of StripeSearch, bit for bit."""
import hashlib

import numpy as np
import pytest

from oracle import oracle as O


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _frame(chr7, g, p):
    D, nz = O.frame_dense(chr7.block, int(g[p + 'start']), int(g[p + 'end']))
    assert np.array_equal(nz, g[p + 'nz'])
    D = np.ascontiguousarray(D[np.ix_(nz, nz)])
    assert sha(D) == str(g[p + 'D_sha'])      # the regenerated synthetic input is the one the reference saw
    return D


@pytest.mark.parametrize('ci', range(6))
def test_stages_match_reference(golden_stages, chr7, ci):
    g = golden_stages
    p = 'c%d_' % ci
    D = _frame(chr7, g, p)
    M = float(g[p + 'M'])
    gw = np.ascontiguousarray(g['gw_2p0'])
    assert np.array_equal(O.brightness_levels(), g['bvals'])
    gp = O.gplane(D, M)
    for bi, b in enumerate(g['bvals']):
        gr = O.gray(gp, b)
        assert sha(gr) == str(g[p + 'gray_sha'][bi])
        assert np.array_equal(gr[D.shape[0] // 2], g[p + 'gray_row'][bi])
        e, dbg = O.canny(gr, gw, 8, debug=True)
        assert sha(dbg['smoothed']) == str(g[p + 'smoothed_sha'][bi])
        assert sha(dbg['isobel']) == str(g[p + 'isobel_sha'][bi])
        assert sha(dbg['jsobel']) == str(g[p + 'jsobel_sha'][bi])
        assert sha(dbg['mag']) == str(g[p + 'mag_sha'][bi])
        assert np.array_equal(np.packbits(e.astype(bool), axis=1), g[p + 'edges'][bi])
        v = O.vertical_line(e)
        assert np.array_equal(np.packbits(v.astype(bool), axis=1), g[p + 'vert'][bi])
        t, en, ud = O.columns(v, 10)
        assert np.array_equal(np.stack([t, en], 1), g[p + 'block'][bi])
    recs, tot = O.stripe_search(D, M, gw=gw)
    got = recs[:, [2, 3, 5, 4]] if len(recs) else np.zeros((0, 4), np.int64)
    assert np.array_equal(got, g[p + 'rec_xywh'])
    assert np.array_equal(tot, g[p + 'rec_total'])
    assert O.medpixel(D) == float(g[p + 'medpixel'])


def test_gauss_weights_close_to_pinned(golden_stages):
    """The weights come from numpy's exp; numpy 2.2 and the pinned 1.26.4 differ by <= 1 ulp on two taps,
    which is why the weights are an explicit input of the C ABI (and the goldens carry theirs)."""
    for sg, key in ((2.0, 'gw_2p0'), (2.5, 'gw_2p5')):
        w, r = O.gauss_weights(sg)
        assert r == int(4 * sg + 0.5) and len(w) == 2 * r + 1
        assert np.allclose(w, golden_stages[key], rtol=0, atol=1e-17)


def test_block_mean_is_numpy_order():
    """np.mean of a small strided 2-D slice = one pairwise sum over the flattened block."""
    import ctypes as C
    L = O._lib_score()
    rng = np.random.default_rng(0)
    mat = rng.random((120, 300)) * rng.integers(1, 1000, (120, 300))
    for bs in (10, 50, 7):
        for _ in range(100):
            r0 = int(rng.integers(0, 120 - bs)); c0 = int(rng.integers(0, 300 - bs))
            assert float(np.mean(mat[r0:r0 + bs, c0:c0 + bs])) == L.so_block_mean(mat.ctypes.data, 120, 300, r0, r0 + bs, c0, c0 + bs)


def test_block_mean_is_numpys_mean():
    """so_block_mean (the window mean of the background step) against np.mean of the same 2-D slices with THIS
    numpy -- also above 8192 elements, where numpy's buffered reduction adds up one pairwise sum per 8192-element
    buffer (the 1.26.4 build of the harness interpreter gives the same bits: docs/history/DESIGN_r01-r05.md, exactness rules)."""
    import ctypes as C
    L = O._lib_score()
    rng = np.random.default_rng(5)
    M = np.ascontiguousarray(rng.random((900, 1100)) * 1e3)
    for bs in (1, 7, 10, 50, 90, 91, 100, 128, 200):
        for _ in range(12):
            r0 = int(rng.integers(0, 900 - bs)); c0 = int(rng.integers(0, 1100 - bs))
            got = L.so_block_mean(M.ctypes.data_as(C.c_void_p), 900, 1100, r0, r0 + bs, c0, c0 + bs)
            assert got == np.mean(M[r0:r0 + bs, c0:c0 + bs]), bs
