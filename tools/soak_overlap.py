"""Randomised soak of the two round-6 measures (image symmetry, frame overlap): whole frame tables (getStripe.py:794-799) of random
chromosomes -- NaN bins, bins whose contacts lie on one side only (a frame pair that does not keep the same bins), short last
frames, shallow / raw / deep count regimes, the five tiled sigmas -- searched with the shipped selection and with both measures
switched off: the record buffers must be identical; every 4th configuration is also compared with the oracle record by record.
    python tools/soak_overlap.py [first_seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import oracle as O
from stripenn_amd import hip, synth

O.build()
ctx = hip.Context(0)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
t0 = time.time(); nrec = 0; nfr_tot = 0; nshared = 0; nora = 0; bad = []


def band_of(dense, hw=512):
    n = dense.shape[0]
    band = np.zeros((n, 2 * hw))
    for i in range(n):
        lo, hi = max(0, i - hw), min(n, i + hw)
        band[i, lo - i + hw:hi - i + hw] = dense[i, lo:hi]
    return band


for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(450, 2600))
    kw = dict(stripe_every=int(rng.integers(20, 250)), stripe_gain=float(rng.uniform(1.5, 5.0)),
              nan_frac=float(rng.choice([0.0, 0.005, 0.02, 0.1])), balanced=bool(rng.integers(0, 2)),
              depth=float(rng.choice([1.0, 1.0, 10.0, 0.3])), count_div=int(rng.choice([1, 1, 1, 8, 32])))
    ch = synth.SynthChrom(n, seed, **kw)
    dense = ch.block(0, n, 0, n)
    for b in rng.integers(0, n, size=int(rng.integers(0, 4))):          # bins that touch only bins well ahead of them
        keep = np.zeros(n, bool); keep[min(n - 1, b + 150):min(n, b + 200)] = True
        dense[b, ~keep] = 0.0; dense[~keep, b] = 0.0
    pos = dense[np.nan_to_num(dense) > 0]
    if len(pos) < 100:
        continue
    Ms = np.quantile(pos, np.sort(rng.uniform(0.8, 0.999, int(rng.integers(1, 4)))))
    sigma = float(rng.choice([2.0, 2.0, 2.0, 2.5, 1.0, 1.5, 3.0]))
    nfr = -(-n // 200)
    st = np.array([max(0, i * 200 - 100) for i in range(nfr)]); en = np.minimum((np.arange(nfr) + 1) * 200 + 99, n - 1)
    band = ctx.band_upload(band_of(dense))
    fr = band.frames(st, en)
    nfr_tot += nfr; nshared += int((fr.overlap() >= 0).sum())
    os.environ.pop('STP_SYM', None); os.environ.pop('STP_REUSE', None)
    a = fr.stripe_search(Ms, sigma=sigma)
    os.environ['STP_SYM'] = '0'; os.environ['STP_REUSE'] = '0'
    b = fr.stripe_search(Ms, sigma=sigma)
    os.environ.pop('STP_SYM', None); os.environ.pop('STP_REUSE', None)
    nrec += len(a)
    ok = a.tobytes() == b.tobytes()
    if ok and seed % 4 == 0:
        gw, gr = hip.gauss_weights(sigma)
        exp = []
        for f in range(nfr):
            D = np.array(dense[st[f]:en[f] + 1, st[f]:en[f] + 1]); D[np.isnan(D)] = 0
            nz = np.where(D.sum(axis=0) != 0)[0]
            if len(nz) <= 10:
                continue
            Dc = np.ascontiguousarray(D[np.ix_(nz, nz)])
            for li, M in enumerate(Ms):
                r, t = O.stripe_search(Dc, float(M), gw=gw)
                exp += [(f, li) + tuple(int(v) for v in q) + (float(tt),) for q, tt in zip(r, t)]
        got = [tuple(int(r[k]) for k in ('frame', 'level', 'b_index', 'ud', 'x', 'y', 'w', 'h')) + (float(r['total']),) for r in a]
        ok = got == exp
        nora += 1
    if not ok:
        bad.append((seed, n, kw, sigma))
        print('MISMATCH seed', seed, n, kw, sigma, flush=True)
    fr.close(); band.close()
print('%d configurations (%d frames, %d frame pairs sharing their block), %d records compared, %d also against the oracle, %d mismatches, %.0f s'
      % (count, nfr_tot, nshared, nrec, nora, len(bad), time.time() - t0))
for b in bad:
    print(b)
ctx.close()
