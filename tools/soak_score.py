"""One-off randomised soak of the score path: stp_pvalue / stp_stripiness on random stripe rectangles against the
oracle's per-stripe restatements (bit-exact comparison, NaN == NaN).
    python tools/soak_score.py [first_seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import oracle as O
from oracle_backend import OracleBackend
from stripenn_amd import backend as BK, synth

O.build()
hb = BK.HipBackend(0); ob = OracleBackend()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
t0 = time.time(); nst = 0; bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    nb = int(rng.integers(900, 2500))
    resol = int(rng.choice([5000, 5000, 10000, 2000]))
    bs = int(50000 / resol)
    ch = synth.SynthChrom(nb, 300 + seed, nan_frac=float(rng.choice([0.0, 0.0, 0.01])), balanced=bool(rng.integers(0, 2)))
    band_h = ch.band(512)
    gb = hb.open_chrom(band_h); cb = ob.open_chrom(band_h)
    ncol = int(rng.integers(200, 1000))
    bg = [np.sort(rng.normal(0, 3, (400, ncol)), axis=1) if rng.random() < 0.3 else rng.normal(0, 3, (400, ncol)) for _ in range(4)]
    for t in bg:
        t[rng.random(t.shape) < 0.01] = np.nan
    hb.set_background(*bg); ob.set_background(*bg)
    EV = 240.0 / (1.0 + np.arange(400)) + 1.0 + rng.random(400)
    n = 150
    w = rng.integers(1, 12, n); h = rng.integers(11, 380, n)
    x0 = rng.integers(bs + 1, nb - 400 - bs - 14, n); x1 = x0 + w - 1
    down = rng.random(n) < 0.5
    y0 = np.where(down, x0, np.maximum(x1 + 1 - h, 1)); y1 = y0 + h - 1
    y1 = np.minimum(y1, nb - 2); y0 = np.minimum(y0, y1 - 10)
    pv = np.zeros(n, dtype=BK.PV_STRIPE_DTYPE)
    pv['row0'], pv['row1'] = y0, y1 + 1
    pv['col0'], pv['col1'] = np.maximum(x0 - bs, 0), np.minimum(x1 + 1 + bs, nb)
    pv['mode'] = np.where(down, 0, 1); pv['upbase'] = y1 + 1 - y0
    gen = rng.random(n) < 0.15                       # rows that inherit a fixed background row
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
    msk = rng.random(n) < 0.1                        # some masked columns / rows
    sc['mcol0'][msk, 0] = 0; sc['mcol1'][msk, 0] = 0; sc['mrow0'][msk] = 2; sc['mrow1'][msk] = 3
    pg, po = hb.pvalue(gb, bs, pv), ob.pvalue(cb, bs, pv)
    if not np.array_equal(pg, po, equal_nan=True):
        bad += 1; print('PVALUE MISMATCH seed', seed, int(np.sum(~((pg == po) | (np.isnan(pg) & np.isnan(po))))), flush=True)
    try:
        rg = hb.stripiness(gb, EV, sc); eg = None
    except IndexError as e:
        rg, eg = None, e
    try:
        ro = ob.stripiness(cb, EV, sc); eo = None
    except IndexError as e:
        ro, eo = None, e
    if (eg is None) != (eo is None):
        bad += 1; print('STRIPINESS ERROR BEHAVIOUR differs, seed', seed, eg, eo, flush=True)
    elif rg is not None:
        for a, b, nm in zip(rg, ro, ('g', 'mean', 'total')):
            ok = np.array_equal(a, b, equal_nan=True) if nm == 'g' else np.allclose(a, b, rtol=1e-9, atol=0, equal_nan=True)
            if not ok:
                bad += 1
                nn = np.isnan(a) != np.isnan(b)
                if nn.any():
                    j = int(np.nonzero(nn)[0][0])
                    print('   NaN pattern differs at', int(nn.sum()), 'stripes; first', j, 'gpu', a[j], 'oracle', b[j], 'masked', bool(msk[j]),
                          'w', int(w[j]), 'h', int(h[j]), 'down', bool(down[j]), 'g gpu/oracle', rg[0][j], ro[0][j], 'mean', rg[1][j], ro[1][j], flush=True)
                d = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
                k = int(np.nanargmax(d))
                print('STRIPINESS', nm, 'MISMATCH seed', seed, 'stripe', k, 'gpu', a[k], 'oracle', b[k], 'rel', d[k],
                      'masked', bool(msk[k]), 'w', int(w[k]), 'h', int(h[k]), 'down', bool(down[k]), 'nworse', int(np.sum(d > 1e-9)), flush=True)
    nst += n
    gb.close()
print('%d chromosomes, %d stripes, %d mismatching arrays, %.0f s' % (count, nst, bad, time.time() - t0))
hb.close()
