"""One-off randomised soak of the remaining device entry points against the oracle / numpy: frame compaction +
medpixel, diagonal sums, null windows (band-served and unit-matrix), order-statistic select, RemoveRedundant, band
packer.      python tools/soak_misc.py [first_seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import pandas as pd
from oracle import oracle as O
from oracle_backend import OracleBackend
from stripenn_amd import backend as BK, getStripe as GS, pixels, synth

O.build()
hb = BK.HipBackend(0); ob = OracleBackend()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = {}
def check(name, ok):
    if not ok:
        bad[name] = bad.get(name, 0) + 1
        print('MISMATCH', name, 'seed', seed, flush=True)
def same(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
t0 = time.time()
for seed in range(first, first + count):
    if os.environ.get('STP_SOAK_VERBOSE'):
        print('seed', seed, flush=True)
    rng = np.random.default_rng(seed)
    nb = int(rng.integers(700, 3000))
    kind = int(rng.integers(0, 3))
    ch = synth.SynthChrom(nb, 900 + seed, nan_frac=float(rng.choice([0.0, 0.01, 0.1])), balanced=(kind != 1))
    band_h = ch.band(512)
    if kind == 2:                                       # heavy duplicates: rounded values
        band_h = np.where(np.isnan(band_h), np.nan, np.round(band_h))
    gb, cb = hb.open_chrom(band_h), ob.open_chrom(band_h)
    # frames: compaction + medpixel
    nf = 12
    st = rng.integers(0, nb - 12, nf); en = np.minimum(st + rng.integers(10, 400, nf), nb - 1)
    fg, fo = hb.frames(gb, st.astype(np.int32), en.astype(np.int32)), ob.frames(cb, st, en)
    check('frames.S', same(fg.S, fo.S))
    check('frames.nz', all(same(fg.nz[f, :fo.S[f]], fo.nz[f, :fo.S[f]]) for f in range(nf)))   # unspecified beyond S
    live = fo.S > 0
    check('medpixel', same(fg.medpixel[live], fo.medpixel[live]))
    fg.close()
    # diagonal sums
    a, b = hb.diag_sums(gb), ob.diag_sums(cb)
    check('diag_sums', same(a[0], b[0]) and same(a[1], b[1]))
    # null windows: band-served and unit-matrix
    bs = int(rng.choice([10, 10, 25, 50]))
    r0 = int(rng.integers(0, max(1, nb - 600))); nrow = int(rng.integers(100, min(500, nb - r0)))
    c0 = max(r0 - 400, 0); c1 = min(r0 + nrow + 400, nb)
    yoff = 400 if c0 > 0 and r0 - c0 == 400 else r0 - c0
    xs = rng.integers(21, max(22, nrow - 21), 24)
    smp = np.zeros(len(xs), dtype=BK.NULL_SAMPLE_DTYPE)
    smp['row0'], smp['nrow'], smp['col0'], smp['ncol'], smp['x'], smp['yoff'] = r0, nrow, c0, c1 - c0, xs, yoff
    reach = 399 + bs // 2 + (bs - bs // 2) + bs + abs((c0 + yoff) - r0)
    wraps = (xs.min() - bs // 2 - bs < 0) or (xs.min() + yoff - 399 - bs // 2 < 0)
    unit = cb.block(r0, r0 + nrow, c0, c1) if (wraps or reach >= 512 or rng.random() < 0.3) else None
    a, b = hb.null_windows(gb, smp, bs, unit), ob.null_windows(cb, smp, bs, unit)
    check('null_windows' + ('_unit' if unit is not None else ''), all(same(x, y) for x, y in zip(a, b)))
    # order statistics
    vals = band_h[np.arange(nb)[:, None], :][:, 0, :]
    pos = vals[vals > 0]
    sg, so = hb.select_open(), ob.select_open()
    for part in np.array_split(vals.ravel(), 3):
        hb.select_append(sg, part[part > 0]); ob.select_append(so, part[part > 0])
    n = hb.select_count(sg)
    check('select_count', n == len(pos) == ob.select_count(so))
    ranks = np.unique(np.clip(rng.integers(0, max(n, 1), 8), 0, max(n - 1, 0)))
    if n:
        check('select_ranks', same(hb.select_ranks(sg, ranks), np.sort(pos)[ranks]))
    hb.select_close(sg); ob.select_close(so)
    # RemoveRedundant (facade bucket table + device pair tests) vs the reference-order loops of the oracle backend
    m = int(rng.integers(5, 400))
    num = np.sort(rng.integers(0, 12, m))
    p1 = num * 1000000 + rng.integers(1, 800000, m); p2 = p1 + rng.integers(5000, 60000, m)
    p3 = num * 1000000 + rng.integers(1, 800000, m); p4 = p3 + rng.integers(50000, 900000, m)
    df = pd.DataFrame({'chr': ['c%d' % (v % 2) for v in rng.integers(0, 2, m)], 'pos1': p1, 'pos2': p2, 'pos3': p3, 'pos4': p4,
                       'h': rng.integers(10, 200, m), 'w': rng.integers(1, 9, m), 'num': num,
                       'pvalue': np.round(rng.random(m), 2), 'Stripiness': np.round(rng.normal(0, 2, m), 1)})
    for by in ('size', 'pvalue', 'score'):
        og = GS.getStripe.__new__(GS.getStripe); og.backend = hb
        oo = GS.getStripe.__new__(GS.getStripe); oo.backend = ob
        for sf in (True, False):
            check('remove_redundant_' + by, og._filter_redundant(df, by, sf).index.tolist() == oo._filter_redundant(df, by, sf).index.tolist())
    # band packer incl. pixels of another chromosome, pixels beyond the band, raw and balanced
    chs = {'a': synth.SynthChrom(int(rng.integers(300, 700)), 50 + seed), 'b': ch}
    t = pixels.PixelTable.from_synth(['a', 'b'], chs, 5000, hw_limit=int(rng.choice([600, 300, 520])))
    for bal in (True, False):
        w = t.weight(bal)
        lo, hi = t.chrom_bins('b')
        pk = hb.ctx.band_pack(t.bin1_id, t.bin2_id, t.count, w, lo, hi - lo, 512)
        check('band_pack', same(pk.download(), O.band_from_pixels(t.bin1_id, t.bin2_id, t.count, w, lo, hi - lo, 512)))
        pk.close()
    gb.close()
print('%d cases in %.0f s; mismatches: %s' % (count, time.time() - t0, bad if bad else 'none'))
hb.close()
