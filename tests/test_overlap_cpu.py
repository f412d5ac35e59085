"""Frame overlap, the claim behind it, on the ORACLE (CPU): grey values and Canny class maps of two neighbouring frames are the
same numbers inside the block they share, 1 / Gaussian radius + 3 pixels from the block's border (stp_phases.h, "frame
overlap"; the reference's windows: getStripe.py:794-799).  At margin 0 they are not (border handling), so the test has teeth."""
import numpy as np
import pytest

from oracle import oracle as O
from stripenn_amd import synth


@pytest.mark.parametrize('sigma', [2.0, 2.5])
def test_class_maps_of_neighbouring_frames_agree_inside_their_shared_block(sigma):
    O.build()
    nb = 1500
    ch = synth.SynthChrom(nb, 9, stripe_every=70, stripe_gain=3.0, nan_frac=0.01)
    gw, R = O.gauss_weights(sigma)
    m = R + 3
    blk = ch.block(0, nb, 0, nb)
    Ms = np.quantile(blk[blk > 0], [0.95, 0.99])
    nfr = -(-nb // 200)
    st = [max(0, i * 200 - 100) for i in range(nfr)]
    en = [min((i + 1) * 200 + 99, nb - 1) for i in range(nfr)]
    frames = []
    for f in range(nfr):
        D, nz = O.frame_dense(ch.block, st[f], en[f])
        frames.append((np.ascontiguousarray(D[np.ix_(nz, nz)]), nz + st[f]))
    n_pairs = n_diff0 = 0
    for f in range(nfr - 1):
        (D0, b0), (D1, b1) = frames[f], frames[f + 1]
        p = int(np.searchsorted(b0, st[f + 1]))
        q = len(b0) - p
        assert q > 2 * m + 32 and np.array_equal(b0[p:], b1[:q])          # NaN bins are dropped by both frames
        for M in Ms:
            g0, g1 = O.gplane(D0, float(M)), O.gplane(D1, float(M))
            for bi in (0, 3, 5):
                b = O.brightness_levels()[bi]
                y0, y1 = O.gray(g0, b, 3), O.gray(g1, b, 3)
                assert np.array_equal(y0[p + 1:len(b0) - 1, p + 1:len(b0) - 1], y1[1:q - 1, 1:q - 1])
                _, d0 = O.canny(y0, gw, R, debug=True)
                _, d1 = O.canny(y1, gw, R, debug=True)
                c0, c1 = d0['cls'], d1['cls']
                S0 = len(b0)
                assert np.array_equal(c0[p + m:S0 - m, p + m:S0 - m], c1[m:q - m, m:q - m]), (f, M, bi)
                n_diff0 += int(not np.array_equal(c0[p:S0, p:S0], c1[:q, :q]))
                n_pairs += 1
    assert n_pairs == (nfr - 1) * 6 and n_diff0 > n_pairs // 2
