"""Other data regimes on the device: raw integer counts (`--norm None`), shallow libraries (counts // 8, // 32:
mostly 0 / 1 / 2 away from the diagonal, few-level images full of exact ties), a 10 x deeper matrix -- every stripe record of
every frame against the oracle, shipped kernels (f32 Canny classes, image symmetry, frame overlap) and all switches off."""
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

REGIMES = {
    'raw_counts': dict(balanced=False),
    'div8': dict(balanced=False, count_div=8),
    'div32': dict(balanced=False, count_div=32),
    'div8_balanced': dict(count_div=8),
    'depth_x10': dict(depth=10.0),
}


@pytest.mark.parametrize('name', sorted(REGIMES))
def test_records_equal_the_oracle(hip_ctx, name):
    from stripenn_amd import synth
    nb = 5000
    ch = synth.SynthChrom(nb, 31, stripe_every=110, stripe_gain=3.0, nan_frac=0.006, **REGIMES[name])
    band_h = ch.band(512)
    pos = band_h[band_h > 0]
    assert len(pos) > 1000
    Ms = np.quantile(pos, [0.95, 0.99])
    band = hip_ctx.band_upload(band_h)
    nfr = -(-nb // 200)
    st = np.array([max(0, i * 200 - 100) for i in range(nfr)]); en = np.minimum((np.arange(nfr) + 1) * 200 + 99, nb - 1)
    fr = band.frames(st, en)
    got = fr.stripe_search(Ms)
    old = {k: os.environ.get(k) for k in ('STP_SYM', 'STP_REUSE')}
    os.environ['STP_SYM'] = '0'; os.environ['STP_REUSE'] = '0'
    try:
        plain = fr.stripe_search(Ms)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert got.tobytes() == plain.tobytes()
    gw, gr = O.gauss_weights(2.0)
    exp = []
    for f in range(nfr):
        D, nz = O.frame_dense(ch.block, int(st[f]), int(en[f]))
        assert len(nz) == fr.S[f] or (len(nz) <= 10 and fr.S[f] == 0)
        if len(nz) <= 10:
            continue
        D = np.ascontiguousarray(D[np.ix_(nz, nz)])
        for li, M in enumerate(Ms):
            r, t = O.stripe_search(D, float(M), gw=gw)
            exp += [(f, li) + tuple(int(v) for v in q) + (float(tt),) for q, tt in zip(r, t)]
    gotl = [tuple(int(r[k]) for k in ('frame', 'level', 'b_index', 'ud', 'x', 'y', 'w', 'h')) + (float(r['total']),) for r in got]
    assert gotl == exp, name
    # the f32 path's own counters on two frames: candidates, pixels sent to the resolver, tile-images handed to the exact kernel
    for f in (3, nfr // 2):
        if fr.S[f] == 0:
            continue
        for bi in (0, 5):
            c = fr.dbg_canny_f32(f, float(Ms[0]), bi)
            assert c['resolved'] <= c['candidates']
    fr.close(); band.close()
